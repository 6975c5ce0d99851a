"""GPU parity of the NN kernel, through the C ABI (icp_hip_nn), bit for bit.

Oracles: (1) the committed golden fixtures, which are outputs of the reference's own compiled
libcpu (tests/golden/make_golden.py); (2) the CPU restatement (oracle/) on seeded synthetic
inputs at sizes it finishes in seconds.  Bar: exact equality of row indices, fp32 distance bits
and returned coordinates.
"""
import numpy as np
import pytest

from conftest import NN_GOLDEN, bits, golden_nn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from icp_amd import capi
    return capi


@pytest.mark.parametrize("mname,pname,cap", NN_GOLDEN)
def test_nn_matches_reference_golden(capi, clouds, mname, pname, cap):
    g = golden_nn(f"nn_{pname}_cap{cap}")
    m, p = clouds[mname], clouds[pname]
    mu_m = capi.mean_host(m)
    assert (bits(mu_m) == bits(g["mu_m"])).all()
    mc = capi.center_host(m, mu_m)
    tree = capi.FlatTree(mc, cap)
    assert tree.height == int(g["height"]) and tree.n_leaves == int(g["leaves"])
    with capi.Context() as ctx:
        ctx.set_tree(tree)
        ctx.set_queries(p)
        out = ctx.nn(g["mu_p"])  # queries are p - mu_p with the reference's mu_p bits
        tm = ctx.timing()
    assert (out["row"] == g["row"]).all()
    assert (bits(out["dist"]) == bits(g["dist"])).all()
    assert (bits(out["xyz"]) == bits(mc[g["row"]])).all()
    # the kernel's work counters are the reference traversal's (they price the algorithmic bytes
    # of bench.py's roofline block): internal nodes visited, bucket points scanned
    assert tm["queries"] == len(p)
    assert tm["nodes_visited"] == float(g["n_int"].astype(np.int64).sum())
    assert tm["points_scanned"] == float(g["n_pts"].astype(np.int64).sum())
    tree.close()


@pytest.mark.parametrize("n,cap,angle", [
    (1000, 3, 5.0), (5000, 8, 2.0), (20000, 32, 5.0), (100_000, 64, 1.0), (100_000, 65, 5.0),
    (100_000, 100, 1.0), (200_000, 129, 3.0), (200_000, 256, 5.0), (200_000, 257, 1.0),
    (100_000, 700, 5.0), (100_000, 1024, 1.0), (1_000_000, 256, 1.0), (1_000_000, 32, 5.0),
])
def test_nn_matches_oracle_synthetic(capi, oracle, n, cap, angle):
    from icp_amd import synth
    m = synth.cloud(n, seed=1000 + n + cap)
    p = synth.apply_fp32(m, synth.ground_truth(angle))
    mu_m = capi.mean_host(m)
    mc = capi.center_host(m, mu_m)
    mu_p = oracle.mean(p)
    q = oracle.subtract(p, mu_p)
    ot = oracle.tree_build(cap, mc)
    assert not oracle.tree_degenerate(ot)
    want = oracle.tree_closest(ot, q, counters=True)
    tree = capi.FlatTree(mc, cap)
    with capi.Context() as ctx:
        ctx.set_tree(tree)
        ctx.set_queries(p)
        out = ctx.nn(mu_p)
        tm = ctx.timing()
    assert tm["nodes_visited"] == float(want["n_int"].astype(np.int64).sum())
    assert tm["points_scanned"] == float(want["n_pts"].astype(np.int64).sum())
    assert (out["row"] == want["row"]).all()
    assert (bits(out["dist"]) == bits(want["dist"])).all()
    assert (bits(out["xyz"]) == bits(want["xyz"])).all()
    tree.close()
