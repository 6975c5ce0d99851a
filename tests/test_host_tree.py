"""CPU suite, part 2: the product's host code (tree builder, validation, mean/centre, SVD solve)
against the oracle, through the C ABI.  No GPU call is made here."""
import ctypes as C

import numpy as np
import pytest

from conftest import NN_GOLDEN, bits, flat_preorder as preorder, golden_nn, tree_sha256
from icp_amd import capi, synth


def same_tree(ft, oracle, ot):
    od, pd = oracle.tree_dump(ot), preorder(ft)
    for k in od:
        assert od[k].shape == pd[k].shape, k
        assert (od[k].view(np.uint8) == pd[k].view(np.uint8)).all(), k
    st = oracle.tree_stats(ot)
    assert (st["height"], st["max_leaf"], st["internal"], st["leaves"]) == \
        (ft.height, ft.max_leaf_count, ft.n_nodes, ft.n_leaves)


@pytest.mark.parametrize("mname,pname,cap", NN_GOLDEN)
def test_builder_and_flat_traversal_on_reference_data(oracle, clouds, mname, pname, cap):
    g = golden_nn(f"nn_{pname}_cap{cap}")
    m, p = clouds[mname], clouds[pname]
    mu_m = capi.mean_host(m)
    assert (bits(mu_m) == bits(g["mu_m"])).all()
    mc = capi.center_host(m, mu_m)
    ft = capi.FlatTree(mc, cap)
    ft.validate()
    same_tree(ft, oracle, oracle.tree_build(cap, mc))
    assert tree_sha256(preorder(ft)) == str(g["tree_sha256"])  # the reference's own tree
    # the kernel's state machine, emulated on the host over the flat layout, gives the
    # reference's answers (golden)
    q = (p - g["mu_p"]).astype(np.float32)
    fw = oracle.flat_walk_closest(ft._ptr, q)
    assert (fw["row"] == g["row"]).all() and (bits(fw["dist"]) == bits(g["dist"])).all()
    ft.close()


@pytest.mark.parametrize("n,cap", [(1, 8), (7, 8), (8, 8), (9, 8), (1000, 3), (5000, 4), (20000, 17),
                                   (100_000, 64), (300_000, 256), (50_000, 1024)])
def test_builder_matches_oracle_on_synthetic(oracle, n, cap):
    m = synth.cloud(n, seed=n + cap)
    mc = capi.center_host(m, capi.mean_host(m))
    ft = capi.FlatTree(mc, cap)
    ft.validate()
    ot = oracle.tree_build(cap, mc)
    same_tree(ft, oracle, ot)
    q = synth.apply_fp32(mc, synth.ground_truth(3.0))[: min(n, 20000)]
    fw = oracle.flat_walk_closest(ft._ptr, q)
    want = oracle.tree_closest(ot, q)
    assert (fw["row"] == want["row"]).all() and (bits(fw["dist"]) == bits(want["dist"])).all()
    assert (bits(fw["xyz"]) == bits(want["xyz"])).all()
    # every row of the cloud owns exactly one place
    rows = np.concatenate([ft.node_row, ft.slot_row[ft.slot_row != 0xFFFFFFFF]])
    assert len(rows) == n and (np.sort(rows) == np.arange(n)).all()
    ft.close()


def test_duplicate_points_and_ties(oracle):
    """Ragged input with repeated points: same tree and same tie decisions as the oracle."""
    rng = np.random.default_rng(5)
    base = rng.integers(-3, 4, size=(400, 3)).astype(np.float32)  # many equal distances
    pts = np.concatenate([base, base[:50]])
    mc = capi.center_host(pts, capi.mean_host(pts))
    ot = oracle.tree_build(16, mc)
    if oracle.tree_degenerate(ot):
        with pytest.raises(capi.IcpHipError) as e:
            capi.FlatTree(mc, 16)
        assert e.value.status == 2
        return
    ft = capi.FlatTree(mc, 16)
    same_tree(ft, oracle, ot)
    q = (mc + np.float32(0.5)).astype(np.float32)
    fw = oracle.flat_walk_closest(ft._ptr, q)
    want = oracle.tree_closest(ot, q)
    assert (fw["row"] == want["row"]).all() and (bits(fw["dist"]) == bits(want["dist"])).all()


def test_degenerate_and_bad_arguments_fail_loudly(oracle):
    same = np.ones((64, 3), np.float32)
    with pytest.raises(capi.IcpHipError) as e:
        capi.FlatTree(same, 8)
    assert e.value.status == 2 and "degenerate" in str(e.value)
    assert oracle.tree_degenerate(oracle.tree_build(8, same))
    for cap in (0, 1, 2):
        with pytest.raises(capi.IcpHipError) as e:
            capi.FlatTree(synth.cloud(100), cap)
        assert e.value.status == 2
    with pytest.raises(capi.IcpHipError) as e:
        capi.FlatTree(np.zeros((0, 3), np.float32), 8)
    assert e.value.status == 1


def test_validate_rejects_malformed_trees():
    m = synth.cloud(5000, seed=1)
    ft = capi.FlatTree(capi.center_host(m, capi.mean_host(m)), 32)
    L = capi.load()

    def corrupted(**kw):
        s = capi.FlatTreeStruct()
        C.memmove(C.byref(s), C.byref(ft.struct), C.sizeof(s))
        keep = []
        for name, (idx, val) in kw.items():
            arr = np.array(getattr(ft, name), copy=True)
            arr[idx] = val
            keep.append(arr)
            setattr(s, name, arr.ctypes.data_as(type(getattr(s, name))))
        return s, keep

    assert L.icp_hip_tree_validate(C.byref(ft.struct)) == 0
    for kw in (dict(node_inside=(0, 10 ** 6)), dict(node_parent=(3, 0 if ft.node_parent[3] != 0 else 1)),
               dict(node_outside=(1, int(ft.node_inside[1]))), dict(leaf_count=(0, 0)),
               dict(leaf_begin=(2, int(ft.n_slots))), dict(slot_row=(int(ft.leaf_begin[1]), 10 ** 7)),
               dict(node_inside=(5, 0))):
        s, keep = corrupted(**kw)
        assert L.icp_hip_tree_validate(C.byref(s)) == 1, kw
    s, _ = corrupted()
    s.height += 1
    assert L.icp_hip_tree_validate(C.byref(s)) == 1
    # what the header promises about buckets and the kernels rely on: 16-slot alignment, +inf /
    # 0xFFFFFFFF padding, disjoint buckets, every row owned exactly once
    b1, c1 = int(ft.leaf_begin[1]), int(ft.leaf_count[1])
    pad = b1 + c1  # first padding slot behind bucket 1 (buckets of this tree are not full)
    assert pad % 16 != 0 and ft.slot_row[pad] == 0xFFFFFFFF and np.isinf(ft.slot_x[pad])
    for kw in (dict(leaf_begin=(1, b1 + 4)),                       # not a multiple of 16
               dict(slot_x=(pad, 0.0)), dict(slot_z=(pad, np.nan)),  # padding must be +inf
               dict(slot_row=(pad, 7)),                            # padding row
               dict(leaf_begin=(2, b1)),                           # overlaps bucket 1
               dict(slot_row=(b1, int(ft.slot_row[b1 + 1]))),      # a row owned twice
               dict(node_row=(0, int(ft.slot_row[b1])))):          # ... by a node and a slot
        s, keep = corrupted(**kw)
        assert L.icp_hip_tree_validate(C.byref(s)) == 1, kw
        assert b"icp_hip_tree_validate" in L.icp_hip_last_global_error()
    ft.close()


def test_alignment_solve_matches_oracle_bitwise(oracle):
    rng = np.random.default_rng(0)
    for _ in range(500):
        H = (rng.standard_normal((3, 3)) * 10.0 ** rng.uniform(-2, 3)).astype(np.float32)
        a, b = rng.standard_normal(3).astype(np.float32), rng.standard_normal(3).astype(np.float32)
        assert (bits(capi.alignment_host(H, a, b)) == bits(oracle.alignment_from_cov(H, a, b))).all()
    for H in (np.zeros((3, 3)), np.eye(3), np.outer([1, 1, 1], [1, 1, 1]), np.diag([1, 0, 0])):
        assert (bits(capi.alignment_host(H, a, b)) == bits(oracle.alignment_from_cov(H, a, b))).all()


def test_mean_and_center_match_oracle(oracle):
    for n in (1, 2, 10, 2903, 100_000):
        a = synth.cloud(n, seed=n) * np.float32(3) + np.float32(0.25)
        mu = capi.mean_host(a)
        assert (bits(mu) == bits(oracle.mean(a))).all()
        assert (bits(capi.center_host(a, mu)) == bits(oracle.subtract(a, mu))).all()
