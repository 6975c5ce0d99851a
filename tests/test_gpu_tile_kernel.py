"""GPU tests of the production NN kernel's own machinery (nn_tile.hip), through the C ABI.

The parity tests (test_gpu_nn_parity.py) already run on this kernel; here the paths that ordinary
inputs touch rarely are forced and checked against the oracle / the level-stack kernel:

* the overflow of the 2-entry LDS traversal stack into HBM (deeply nested far-side searches);
* buckets larger than one staged window (> 256 slots) and buckets of a single point;
* every ordering of the shard (as given, Morton, Morton + leaf order) gives the same answers in
  the caller's order;
* the per-chunk sums make a run bitwise reproducible although chunks are pulled from a queue.
"""
import os

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def _nn(capi, tree, p, mu_p, env=None):
    """One NN pass in a fresh context created under `env` (the library reads it at create time)."""
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        with capi.Context() as ctx:
            ctx.set_tree(tree)
            ctx.set_queries(p)
            out = ctx.nn(mu_p)
            out["timing"] = ctx.timing()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return out


def _same(a, b):
    assert (a["row"] == b["row"]).all()
    assert (bits(a["dist"]) == bits(b["dist"])).all()
    assert (bits(a["xyz"]) == bits(b["xyz"])).all()


def test_stack_spill_matches_oracle(oracle):
    """Small buckets + a grossly misaligned cloud: far sides nest deeper than the LDS entries."""
    from icp_amd import capi, synth
    n, cap = 60_000, 4
    m = synth.cloud(n, seed=5)
    p = synth.apply_fp32(m, synth.ground_truth(25.0))
    mu_m = capi.mean_host(m)
    mc = capi.center_host(m, mu_m)
    mu_p = oracle.mean(p)
    want = oracle.tree_closest(oracle.tree_build(cap, mc), oracle.subtract(p, mu_p))
    tree = capi.FlatTree(mc, cap)
    assert tree.height > 7
    got = _nn(capi, tree, p, mu_p)
    assert got["timing"]["stack_spills"] > 0, "input did not reach the spill path"
    _same(got, want)
    deep = _nn(capi, tree, p, mu_p, {"ICP_HIP_NN_KERNEL": "deep"})
    assert deep["timing"]["stack_spills"] == 0
    _same(deep, want)
    # both kernels count the reference traversal's work identically
    assert deep["timing"]["nodes_visited"] == got["timing"]["nodes_visited"]
    assert deep["timing"]["points_scanned"] == got["timing"]["points_scanned"]
    tree.close()


@pytest.mark.parametrize("n,cap", [(40_000, 300), (40_000, 2000), (3_000, 3)])
def test_window_and_tiny_buckets(oracle, n, cap):
    from icp_amd import capi, synth
    m = synth.cloud(n, seed=n + cap)
    p = synth.apply_fp32(m, synth.ground_truth(3.0))
    mu_m = capi.mean_host(m)
    mc = capi.center_host(m, mu_m)
    mu_p = oracle.mean(p)
    want = oracle.tree_closest(oracle.tree_build(cap, mc), oracle.subtract(p, mu_p))
    tree = capi.FlatTree(mc, cap)
    _same(_nn(capi, tree, p, mu_p), want)
    tree.close()


def test_single_leaf_tree(oracle):
    """capacity > n: the root is a bucket (no internal node at all)."""
    from icp_amd import capi, synth
    m = synth.cloud(700, seed=3)
    p = synth.apply_fp32(m, synth.ground_truth(10.0))
    mc = capi.center_host(m, capi.mean_host(m))
    mu_p = oracle.mean(p)
    want = oracle.tree_closest(oracle.tree_build(1024, mc), oracle.subtract(p, mu_p))
    tree = capi.FlatTree(mc, 1024)
    assert tree.n_nodes == 0
    _same(_nn(capi, tree, p, mu_p), want)
    tree.close()


def test_orderings_agree(oracle):
    from icp_amd import capi, synth
    n, cap = 150_000, 128
    m = synth.cloud(n, seed=77)
    p = synth.apply_fp32(m, synth.ground_truth(2.0))
    mc = capi.center_host(m, capi.mean_host(m))
    mu_p = oracle.mean(p)
    tree = capi.FlatTree(mc, cap)
    base = _nn(capi, tree, p, mu_p)
    for env in ({"ICP_HIP_SORT_QUERIES": "0"}, {"ICP_HIP_LEAF_ORDER": "0"},
                {"ICP_HIP_TILE_CUT": "0"}, {"ICP_HIP_TILE_CUT": "4"},
                {"ICP_HIP_NN_KERNEL": "deep"},
                # how the work is cut and how many workgroups share a CU never shows either
                {"ICP_HIP_CHUNK_BATCHES": "1"}, {"ICP_HIP_CHUNK_BATCHES": "8"},
                {"ICP_HIP_TILE_WGS_PER_CU": "1"}, {"ICP_HIP_TILE_WGS_PER_CU": "5"}):
        got = _nn(capi, tree, p, mu_p, env)
        _same(got, base)
        assert got["timing"]["nodes_visited"] == base["timing"]["nodes_visited"]
        assert got["timing"]["points_scanned"] == base["timing"]["points_scanned"]
    tree.close()


def test_duplicate_points_and_exact_ties(oracle):
    """Every point twice + queries ON points: zero distances and exact d2 ties inside buckets
    (the kernel's 'minimum not isolated' path); the reference keeps the first slot."""
    from icp_amd import capi, synth
    base = synth.cloud(20_000, seed=9)
    m = np.concatenate([base, base[::-1]]).astype(np.float32)
    mu_m = capi.mean_host(m)
    mc = capi.center_host(m, mu_m)
    ot = oracle.tree_build(64, mc)
    if oracle.tree_degenerate(ot):
        pytest.skip("duplicate cloud hits the reference's undefined empty-side case")
    tree = capi.FlatTree(mc, 64)
    want = oracle.tree_closest(ot, oracle.subtract(m, mu_m))
    _same(_nn(capi, tree, m, mu_m), want)
    tree.close()


def test_run_is_bitwise_reproducible():
    import icp_amd
    from icp_amd import synth
    m = synth.cloud(300_000, seed=11)
    p = synth.apply_fp32(m, synth.ground_truth(1.5))
    a = icp_amd.icp(m, p, 12, 0.0, 64, return_info=True)
    b = icp_amd.icp(m, p, 12, 0.0, 64, return_info=True)
    assert (bits(a[0]) == bits(b[0])).all()
    assert (bits(a[1]) == bits(b[1])).all()
    assert a[2]["error"] == b[2]["error"]


@pytest.mark.parametrize("n,cap,angle", [(48_000, 32, 6.0), (48_000, 256, 2.0), (700, 16, 10.0)])
def test_batch_widths_agree(oracle, n, cap, angle):
    """A shard too small to fill the chip runs 4/8/16/32 queries per wave instead of 64 (idle lanes
    join the bucket scans): every width gives the reference's answers, and the work counters of
    the reference traversal do not depend on it."""
    from icp_amd import capi, synth
    m = synth.cloud(n, seed=n + cap)
    p = synth.apply_fp32(m, synth.ground_truth(angle))
    mc = capi.center_host(m, capi.mean_host(m))
    mu_p = oracle.mean(p)
    want = oracle.tree_closest(oracle.tree_build(cap, mc), oracle.subtract(p, mu_p))
    tree = capi.FlatTree(mc, cap)
    runs = [_nn(capi, tree, p, mu_p, {"ICP_HIP_BATCH_QUERIES": str(w)}) for w in (64, 32, 16, 8, 4, 2)]
    for r in runs:
        _same(r, want)
        assert r["timing"]["nodes_visited"] == runs[0]["timing"]["nodes_visited"]
        assert r["timing"]["points_scanned"] == runs[0]["timing"]["points_scanned"]
    _same(_nn(capi, tree, p, mu_p), want)  # the width the library picks by itself
    tree.close()


@pytest.mark.parametrize("ulps", [1, 8, 200])
def test_queries_a_few_ulps_from_tree_points(oracle, ulps):
    """The kernel skips the way back up when the bucket's answer is closer than every
    |radius - d| of the descent (nn_tile.hip, chain_lb).  Every internal node has a tree point
    exactly ON its sphere (the median element, vp-tree.cc:48-52,77), and every centre is a tree
    point: queries a few ulps away from the tree's own points put the answer's distance, the
    distances to the spheres and the centre distances all at the same rounding-level scale."""
    from icp_amd import capi, synth
    n, cap = 120_000, 16
    m = synth.cloud(n, seed=4242)
    mu_m = capi.mean_host(m)
    mc = capi.center_host(m, mu_m)
    rng = np.random.default_rng(ulps)
    k = rng.integers(-ulps, ulps + 1, size=mc.shape).astype(np.int32)
    q = (mc.view(np.int32) + k).view(np.float32)  # neighbouring floats of the tree's coordinates
    q = np.ascontiguousarray(q)
    assert np.isfinite(q).all()
    want = oracle.tree_closest(oracle.tree_build(cap, mc), q)
    tree = capi.FlatTree(mc, cap)
    zero = np.zeros(3, np.float32)
    _same(_nn(capi, tree, q, zero), want)
    _same(_nn(capi, tree, q, zero, {"ICP_HIP_NN_KERNEL": "deep"}), want)
    tree.close()


@pytest.mark.parametrize("noise", [2e-6, 1e-4, 0.0])
def test_ring_walk_with_many_queries_per_bucket(oracle, noise):
    """The ring walk (nn_tile.hip, ring_walk) needs 16 lanes on one bucket: several queries per
    tree point, at noise level (the walk finishes after a point or two), a ring spacing away (it
    often does not, and the tile scan takes over) and exactly ON the tree points (distance 0 and
    duplicates of the query: the isolation test sends them to the exact scan).  The tree is small
    enough to go up as one host-packed block, whose ring words the host ranks (icp_hip.cc)."""
    from icp_amd import capi
    from icp_amd import synth
    n, reps = 100_000, 8
    rng = np.random.default_rng(5)
    m = synth.cloud(n, seed=n)
    mc = capi.center_host(m, capi.mean_host(m))
    p = np.tile(mc, (reps, 1)) + rng.normal(0, noise, size=(n * reps, 3)).astype(np.float32)
    p = np.ascontiguousarray(p[rng.permutation(len(p))].astype(np.float32))
    zero = np.zeros(3, np.float32)
    tree = capi.FlatTree(mc, 256)
    got = _nn(capi, tree, p, zero)
    sub = rng.permutation(len(p))[:60_000]
    want = oracle.tree_closest(oracle.tree_build(256, mc), p[sub])
    assert (got["row"][sub] == want["row"]).all()
    assert (bits(got["dist"][sub]) == bits(want["dist"])).all()
    assert (bits(got["xyz"][sub]) == bits(want["xyz"])).all()
    tree.close()


def test_outputs_not_asked_for_are_not_needed():
    """icp_hip_nn computes rows and distances on the device only when the caller passes a buffer
    for them (they go back to the caller's order with scattered stores): every subset of the
    outputs gives the same values as the full call."""
    from icp_amd import capi, synth
    n = 300_000
    m = synth.cloud(n, seed=41)
    mu_m = capi.mean_host(m)
    p = synth.apply_fp32(m, synth.ground_truth(0.5))[np.random.default_rng(2).permutation(n)]
    mu_p = capi.mean_host(p)
    with capi.Context() as ctx:
        ctx.build_tree(capi.center_host(m, mu_m), 256)
        ctx.set_queries(p)
        full = ctx.nn(mu_p)
        for want in (("xyz",), ("row",), ("dist",), ("row", "dist"), ()):
            part = ctx.nn(mu_p, want=want)
            for k in ("row", "xyz", "dist"):
                if k in want:
                    assert (bits(part[k]) == bits(full[k])).all() if k != "row" else (part[k] == full[k]).all()
                else:
                    assert part[k] is None
        again = ctx.nn(mu_p)
        assert (again["row"] == full["row"]).all() and (bits(again["dist"]) == bits(full["dist"])).all()
