"""BASELINE.json config 5 on the GPU: the 64,000,000-point synthetic cloud, capacity 256.

What is checked at full size (reference rules: libcpu/libcpu/vp-tree.cc:54-123 for the tree,
:125-187 for search / closest):
  (a) the tree built on the device is the host builder's tree, array by array (the host builder
      is pinned to the reference's compiled libcpu in tests/test_host_tree.py);
  (b) one NN pass over ALL 64M points to move (0.25 degree motion, the bench's workload): a
      1,000,000-query subsample (every 64th point) is bit-exact -- row, fp32 distance,
      coordinates -- against the oracle walking the same tree on the host;
  (c) the kernel's work counters (internal nodes visited, bucket points scanned) over that
      subsample equal the oracle's, query for query in the sum;
  (d) size-independent properties over all 64M answers: every returned distance is
      sqrt(|q - y|^2) of the returned row evaluated in the reference's operation order, and no
      sampled brute-force neighbour is closer than the answer;
  (e) the registration converges to the known motion (|T - T_true| <= 1e-5).
"""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu

N = 64_000_000
CAP = 256
ANGLE = 0.25


@pytest.fixture(scope="module")
def world():
    from icp_amd import capi, synth
    m = synth.cloud(N)
    mu_m = capi.mean_host(m)
    mc = capi.center_host(m, mu_m)
    p = synth.apply_fp32(m, synth.ground_truth(ANGLE))
    del m
    ctx = capi.Context()
    ctx.build_tree(mc, CAP)
    yield dict(capi=capi, synth=synth, mu_m=mu_m, mc=mc, p=p, ctx=ctx)
    ctx.close()


def ref_dist(q, y):
    """sqrt(((dx*dx)+(dy*dy))+(dz*dz)) in fp32, query first (point-3d.cc:136-143, vp-tree.cc:132)."""
    d = q - y
    s = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    return np.sqrt(s, dtype=np.float32)


def test_64m_tree_nn_counters_and_convergence(world, oracle):
    capi, ctx, mc, p, mu_m = (world[k] for k in ("capi", "ctx", "mc", "p", "mu_m"))
    info = ctx.tree_info()
    assert info["n_points"] == N and info["height"] == 18 and info["n_leaves"] == 262144

    # (a) device-built tree == host-built tree
    dev = ctx.get_tree()
    host = capi.FlatTree(mc, CAP)
    for k in ("node_cx", "node_cy", "node_cz", "node_radius", "slot_x", "slot_y", "slot_z"):
        assert (bits(getattr(dev, k)) == bits(getattr(host, k))).all(), k
    for k in ("node_inside", "node_outside", "node_parent", "node_row", "leaf_begin", "leaf_count",
              "slot_row"):
        assert (np.asarray(getattr(dev, k)) == np.asarray(getattr(host, k))).all(), k
    host.close()

    # (b) the full pass
    mu_p = p.mean(axis=0, dtype=np.float64).astype(np.float32)
    ctx.set_queries(p)
    out = ctx.nn(mu_p)
    tm_full = ctx.timing()
    assert tm_full["queries"] == N
    sub = np.arange(0, N, 64)
    q_sub = (p[sub] - mu_p).astype(np.float32)  # fp32 subtraction, as subtract() does
    otree = oracle.tree_from_flat(dev)
    want = oracle.tree_closest(otree, q_sub, counters=True)
    assert (out["row"][sub] == want["row"]).all()
    assert (bits(out["dist"][sub]) == bits(want["dist"])).all()
    assert (bits(out["xyz"][sub]) == bits(want["xyz"])).all()

    # (d) properties over every answer
    q = p - mu_p
    assert (bits(out["xyz"]) == bits(mc[out["row"]])).all()
    assert (bits(ref_dist(q, out["xyz"])) == bits(out["dist"])).all()
    rng = np.random.default_rng(0)
    probe = rng.integers(0, N, 200)
    cand = mc[rng.integers(0, N, 200_000)]
    # (the prune test compares rounded fp32 distances, vp-tree.cc:152: the answer is the nearest
    # point up to that rounding, hence the 1e-6 of slack)
    for i in probe[:200]:
        assert ref_dist(q[i][None, :], cand).min() * (1 + 1e-6) >= out["dist"][i]
    # a point's own pre-image is a candidate: the answer is never farther than it
    assert (out["dist"] <= ref_dist(q, mc) * np.float32(1 + 1e-6)).all()
    del q, out

    # (c) counters: the subsample as a shard of its own
    ctx.set_queries(p[sub])
    o2 = ctx.nn(mu_p)
    tm = ctx.timing()
    assert (o2["row"] == want["row"]).all()
    assert tm["nodes_visited"] == float(want["n_int"].astype(np.int64).sum())
    assert tm["points_scanned"] == float(want["n_pts"].astype(np.int64).sum())
    del otree
    dev.close()

    # (e) the registration itself
    ctx.set_queries(p)
    T, iters, err = ctx.run(mu_m, 30, 1e-5)
    want_T = world["synth"].expected_transform(ANGLE)
    assert np.abs(T.astype(np.float64) - want_T).max() <= 1e-5
    moved = ctx.get_points()
    assert np.abs(moved - (mc + mu_m)).max() < 5e-3  # test_functionnal.cc's tolerance
