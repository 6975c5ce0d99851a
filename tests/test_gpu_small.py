"""Shards below 8192 points take a path of their own on the host side (icp_hip.cc, kSmallBelow):
no sort, loop state + work queue + coordinates in one device block that goes up with the first
user and comes back with the loop state, contexts built from a per-device pool.  The kernels and
the arithmetic are the ones every other size uses; what is checked here is that the plumbing
does not change a bit: every consumer of the lazily uploaded block, both sides of the size
boundary against the oracle (libcpu/libcpu/vp-tree.cc:125-187), contexts recycled through the
pool in any order of sizes.
"""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def _register(capi, m, p, cap, iters=200, thr=1e-5):
    mu_m = capi.mean_host(m)
    with capi.Context() as ctx:
        ctx.build_tree(capi.center_host(m, mu_m), cap)
        ctx.set_queries(p)
        T, it, err = ctx.run(mu_m, iters, thr)
        return T, it, err, ctx.get_points()


def test_every_consumer_uploads_the_block(clouds):
    from icp_amd import capi
    m, p = clouds["cow_ref"], clouds["cow_tr2"]
    mu_m = capi.mean_host(m)
    mc = capi.center_host(m, mu_m)
    mu_p = p.mean(axis=0, dtype=np.float64).astype(np.float32)
    want = _register(capi, m, p, 32)
    with capi.Context() as ctx:
        ctx.build_tree(mc, 32)
        # get_points is the first user
        ctx.set_queries(p)
        assert (bits(ctx.get_points()) == bits(p)).all()
        # nn is the first user; then the loop (state uploaded alone), then the points
        ctx.set_queries(p)
        a = ctx.nn(mu_p)
        T, it, err = ctx.run(mu_m, 200, 1e-5)
        assert it == want[1] and (bits(T) == bits(want[0])).all()
        assert (bits(ctx.get_points()) == bits(want[3])).all()
        # the loop is the first user, twice over: the second run starts from the moved cloud
        ctx.set_queries(p)
        T, it, err = ctx.run(mu_m, 200, 1e-5)
        assert it == want[1] and (bits(T) == bits(want[0])).all()
        T2, it2, err2 = ctx.run(mu_m, 200, 1e-5)
        assert it2 >= 1 and np.abs(T2 - np.eye(4, dtype=np.float32)).max() < 1e-3
        assert np.abs(ctx.get_points() - m).max() < 5e-3
        # a pass after the loop sees the moved cloud
        b = ctx.nn(capi.mean_host(ctx.get_points()))
        assert (b["dist"] <= a["dist"].max()).all() and b["dist"].max() < 1e-2
        # zero iterations: nothing came back with the state, the cloud is read from the device
        ctx.set_queries(p)
        T, it, err = ctx.run(mu_m, 0, 1e-5)
        assert it == 0 and (bits(ctx.get_points()) == bits(p)).all()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 8191, 8192, 8193])
def test_both_sides_of_the_size_boundary_match_the_oracle(oracle, n):
    from icp_amd import capi, synth
    m = synth.cloud(20_000, seed=5)
    mu_m = capi.mean_host(m)
    mc = capi.center_host(m, mu_m)
    p = synth.apply_fp32(synth.cloud(n, seed=6), synth.ground_truth(3.0))
    mu_p = p.mean(axis=0, dtype=np.float64).astype(np.float32)
    with capi.Context() as ctx:
        ctx.build_tree(mc, 32)
        ctx.set_queries(p)
        out = ctx.nn(mu_p)
        tm = ctx.timing()
        dev = ctx.get_tree()
    want = oracle.tree_closest(oracle.tree_from_flat(dev), (p - mu_p).astype(np.float32),
                               counters=True)
    dev.close()
    assert (out["row"] == want["row"]).all()
    assert (bits(out["dist"]) == bits(want["dist"])).all()
    assert (bits(out["xyz"]) == bits(want["xyz"])).all()
    assert tm["nodes_visited"] == float(want["n_int"].astype(np.int64).sum())
    assert tm["points_scanned"] == float(want["n_pts"].astype(np.int64).sum())


def test_one_context_small_large_small(clouds):
    """The views into the small block are dropped and taken again as the shard size changes."""
    from icp_amd import capi
    cases = [("cow", "cow_tr2", 32), ("horse", "horse_tr2", 256), ("line", "line_translated_2_3_4", 8),
             ("horse", "horse_tr1", 32), ("cow", "cow_tr1", 256)]
    ref = {c: _register(capi, clouds[f"{c[0]}_ref"], clouds[c[1]], c[2]) for c in cases}
    with capi.Context() as ctx:
        for c in cases + cases[::-1]:
            T, new_p, it, err = ctx.icp(clouds[f"{c[0]}_ref"], clouds[c[1]], 200, 1e-5, c[2])
            assert it == ref[c][1] and err == ref[c][2], c
            assert (bits(T) == bits(ref[c][0])).all(), c
            assert (bits(new_p) == bits(ref[c][3])).all(), c


def test_contexts_from_the_pool(clouds):
    """Streams, events, pinned and device blocks are handed from a destroyed context to the next
    one: results do not depend on what the previous owner left in them, and contexts that are
    alive together never share any."""
    from icp_amd import capi
    small = ("cow", "cow_tr2", 32)
    big = ("horse", "horse_tr2", 256)
    ref = {c: _register(capi, clouds[f"{c[0]}_ref"], clouds[c[1]], c[2]) for c in (small, big)}
    for rep in range(12):
        c = small if rep % 3 else big
        got = _register(capi, clouds[f"{c[0]}_ref"], clouds[c[1]], c[2])
        assert got[1] == ref[c][1] and (bits(got[0]) == bits(ref[c][0])).all()
        assert (bits(got[3]) == bits(ref[c][3])).all()
    # three at once, steps interleaved
    ctxs = [capi.Context() for _ in range(3)]
    try:
        m, p = clouds["cow_ref"], clouds["cow_tr2"]
        mu_m = capi.mean_host(m)
        mc = capi.center_host(m, mu_m)
        for x in ctxs:
            x.build_tree(mc, 32)
        for x in ctxs:
            x.set_queries(p)
        res = [x.run(mu_m, 200, 1e-5) for x in ctxs]
        pts = [x.get_points() for x in ctxs]
        for r, q in zip(res, pts):
            assert r[1] == ref[small][1] and (bits(r[0]) == bits(ref[small][0])).all()
            assert (bits(q) == bits(ref[small][3])).all()
    finally:
        for x in ctxs:
            x.close()


def test_small_shard_rejects_non_finite_input(clouds):
    from icp_amd import capi
    m = clouds["cow_ref"]
    mu_m = capi.mean_host(m)
    with capi.Context() as ctx:
        ctx.build_tree(capi.center_host(m, mu_m), 32)
        for bad in (np.nan, np.inf, -np.inf):
            p = clouds["cow_tr1"].copy()
            p[17, 1] = bad
            with pytest.raises(capi.IcpHipError) as e:
                ctx.set_queries(p)
            assert e.value.status == 1 and "finite" in str(e.value)
        # +inf and -inf in one column cancel to NaN in the sum, not to a finite value
        p = clouds["cow_tr1"].copy()
        p[3, 0], p[4, 0] = np.inf, -np.inf
        with pytest.raises(capi.IcpHipError):
            ctx.set_queries(p)
        # finite values whose fp32 sum would overflow are legal
        p = clouds["cow_tr1"].copy()
        p[:, 2] = np.float32(3e38)
        ctx.set_queries(p)
        assert (bits(ctx.get_points()) == bits(p)).all()
