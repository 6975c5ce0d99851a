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
