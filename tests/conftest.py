import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def bits(a):
    """uint32 view of float32 data, for bit-for-bit comparisons."""
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="session")
def oracle():
    from oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref (compiled reference libcpu) not present")
    return Ref()


@pytest.fixture(scope="session")
def clouds():
    class _Clouds(dict):
        def __missing__(self, name):
            self[name] = np.load(os.path.join(GOLDEN, "data", name + ".npy"))
            return self[name]
    return _Clouds()


def flat_preorder(ft):
    """Pre-order dump of a flat tree (capi.FlatTree) in the layout of oracle.tree_dump / the
    reference's own tree dump (tests/golden/make_golden.py)."""
    kind, node_f, node_row, leaf_count, leaf_xyz, leaf_row = [], [], [], [], [], []
    nc = np.stack([ft.node_cx, ft.node_cy, ft.node_cz, ft.node_radius], 1) if ft.n_nodes else None
    st = [ft.root]
    while st:
        c = st.pop()
        if c < 0:
            b, n = int(ft.leaf_begin[~c]), int(ft.leaf_count[~c])
            kind.append(0)
            leaf_count.append(n)
            leaf_xyz.append(np.stack([ft.slot_x[b:b + n], ft.slot_y[b:b + n], ft.slot_z[b:b + n]], 1))
            leaf_row.append(ft.slot_row[b:b + n])
        else:
            kind.append(1)
            node_f.append(nc[c])
            node_row.append(ft.node_row[c])
            st.append(int(ft.node_outside[c]))
            st.append(int(ft.node_inside[c]))
    return dict(kind=np.array(kind, np.uint8), node_f=np.array(node_f, np.float32).reshape(-1, 4),
                node_row=np.array(node_row, np.uint32), leaf_count=np.array(leaf_count, np.uint32),
                leaf_xyz=np.concatenate(leaf_xyz).astype(np.float32),
                leaf_row=np.concatenate(leaf_row).astype(np.uint32))


def tree_sha256(dump):
    """Fingerprint of a pre-order tree dump, as tests/golden/make_golden.py takes it of the
    reference's compiled libcpu tree (golden field tree_sha256)."""
    import hashlib
    h = hashlib.sha256()
    for k in ("kind", "node_f", "leaf_count", "leaf_xyz"):
        h.update(np.ascontiguousarray(dump[k]).tobytes())
    return h.hexdigest()


def golden_nn(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


NN_GOLDEN = [
    ("line_ref", "line_translated_2_3_4", 8), ("line_ref", "line_translated_2_3_4", 32),
    ("cow_ref", "cow_tr1", 8), ("cow_ref", "cow_tr1", 32), ("cow_ref", "cow_tr1", 256),
    ("cow_ref", "cow_tr2", 8), ("cow_ref", "cow_tr2", 32), ("cow_ref", "cow_tr2", 256),
    ("horse_ref", "horse_tr1", 32), ("horse_ref", "horse_tr1", 256), ("horse_ref", "horse_tr2", 256),
]


def has_gpu():
    try:
        from icp_amd import capi
        ctx = capi.Context()
        ctx.close()
        return True
    except Exception:
        return False
