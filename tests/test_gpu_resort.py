"""The second ordering pass of a long registration (icp_hip.cc, maybe_resort): once the clouds are
aligned the shard is put in leaf order again.  The order a shard is stored in is invisible to the
caller: rows, distances and the moved cloud come back in the caller's order (the reference
evaluates queries in input order, libcpu/libcpu/vp-tree.cc:182-184), NN answers are the oracle's
bit for bit before and after the pass, and the registration is the one without the pass up to the
order of the fp64 sums."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu

N = 700_000


@pytest.fixture(scope="module")
def world():
    from icp_amd import capi, synth
    m = synth.cloud(N, seed=11)
    mu_m = capi.mean_host(m)
    mc = capi.center_host(m, mu_m)
    rng = np.random.default_rng(3)
    order = rng.permutation(N)            # the caller's order is not a spatial one
    p = synth.apply_fp32(m, synth.ground_truth(1.0))[order]
    return dict(m=m, mu_m=mu_m, mc=mc, p=p, order=order)


def _run(monkeypatch, resort, world, iters):
    from icp_amd import capi
    if resort:
        monkeypatch.delenv("ICP_HIP_RESORT", raising=False)
    else:
        monkeypatch.setenv("ICP_HIP_RESORT", "0")
    with capi.Context() as ctx:
        ctx.build_tree(world["mc"], 256)
        ctx.set_queries(world["p"])
        T, it, err = ctx.run(world["mu_m"], iters, 0.0)
        moved = ctx.get_points()
        trace = ctx.nn_trace()
        mu = capi.mean_host(moved)
        out = ctx.nn(mu)
        dev = ctx.get_tree()
    return dict(T=T, it=it, err=err, moved=moved, trace=trace, mu=mu, out=out, tree=dev)


def test_resort_is_invisible_to_the_caller(monkeypatch, world, oracle):
    a = _run(monkeypatch, True, world, 16)
    b = _run(monkeypatch, False, world, 16)
    assert a["it"] == b["it"] == 16
    # the same registration up to the summation order of the fp64 sums
    assert np.abs(a["T"] - b["T"]).max() <= 2e-7
    assert np.abs(a["moved"] - b["moved"]).max() <= 1e-6
    # the moved cloud is in the caller's order: row i is the image of input row i
    m = world["m"][world["order"]]
    assert np.abs(a["moved"] - m).max() < 1e-4
    # a pass after the re-ordering: rows / distances / coordinates in the caller's order, and the
    # oracle's bit for bit on the same queries
    q = (a["moved"] - a["mu"]).astype(np.float32)
    sub = np.arange(0, N, 7)
    want = oracle.tree_closest(oracle.tree_from_flat(a["tree"]), q[sub])
    assert (a["out"]["row"][sub] == want["row"]).all()
    assert (bits(a["out"]["dist"][sub]) == bits(want["dist"])).all()
    assert (bits(a["out"]["xyz"][sub]) == bits(want["xyz"])).all()
    assert (a["out"]["row"] == world["order"]).mean() > 0.999  # each point finds its own pre-image
    # the pass did happen, and it paid: later NN passes are faster than without it
    assert np.median(a["trace"][-4:]) < np.median(b["trace"][-4:])
    a["tree"].close()
    b["tree"].close()


def test_short_or_unfinished_registrations_are_left_alone(monkeypatch, world):
    """Fewer than four iterations to go, or still far from aligned: no second pass (the run is
    then bit-identical to the run with the pass switched off)."""
    a = _run(monkeypatch, True, world, 3)
    b = _run(monkeypatch, False, world, 3)
    assert (bits(a["T"]) == bits(b["T"])).all() and (bits(a["moved"]) == bits(b["moved"])).all()
    a["tree"].close()
    b["tree"].close()


def test_resort_under_a_communicator_equals_the_plain_run(monkeypatch, world):
    """With an RCCL communicator (one rank here) the loop condition is known one slot late, and
    the second ordering pass is decided from `started`, not `it`: the same iteration triggers it
    as in the host-driven run of the same sharded protocol (an all-reduce callback with nobody
    to add), so the two runs are bit-identical; the one-GPU loop (queries centred on mean(new_p)
    instead of mu_m) agrees within the centroid's rounding."""
    from icp_amd import capi
    monkeypatch.delenv("ICP_HIP_RESORT", raising=False)
    res = []
    for mode in ("callback", "rccl", "one GPU"):
        with capi.Context() as ctx:
            ctx.build_tree(world["mc"], 256)
            if mode == "rccl":
                ctx.comm_init_rank(1, 0, capi.Context.comm_unique_id())
            elif mode == "callback":
                ctx.set_allreduce(lambda a: None)
            ctx.set_queries(world["p"])
            errs = []
            T, it, err = ctx.run(world["mu_m"], 16, 0.0, progress=lambda i, mx, e: errs.append(e))
            res.append((T.copy(), it, err, errs, ctx.get_points(), ctx.nn_trace()))
    assert res[0][1] == res[1][1] == 16 and res[0][3] == res[1][3]
    assert (bits(res[0][0]) == bits(res[1][0])).all()
    assert (bits(res[0][4]) == bits(res[1][4])).all()
    assert res[2][1] == 16 and np.abs(res[2][0] - res[1][0]).max() <= 1e-6
    assert np.abs(res[2][4] - res[1][4]).max() <= 1e-6


def test_many_rank_rule_decides_from_nothing_a_single_rank_sees(monkeypatch, world):
    """With more than one rank every rank must enqueue the same slots, so the decision about the
    pass (and the enqueue blocks that depend on it) is taken from the all-reduced loop state and
    the tree alone, not from the shard (icp_hip.cc, resort_rule_global).  ICP_HIP_RESORT=g applies
    that rule to a communicator of one rank: the pass still happens, later passes are faster than
    without it, and the registration is the plain one up to the order of the fp64 sums."""
    from icp_amd import capi
    res = {}
    for mode in ("0", "g"):
        monkeypatch.setenv("ICP_HIP_RESORT", mode)
        with capi.Context() as ctx:
            ctx.build_tree(world["mc"], 256)
            ctx.comm_init_rank(1, 0, capi.Context.comm_unique_id())
            ctx.set_queries(world["p"])
            T, it, err = ctx.run(world["mu_m"], 16, 0.0)
            res[mode] = (T.copy(), it, ctx.get_points(), ctx.nn_trace())
    assert res["0"][1] == res["g"][1] == 16
    assert np.abs(res["0"][0] - res["g"][0]).max() <= 2e-7
    assert np.abs(res["0"][2] - res["g"][2]).max() <= 1e-6
    assert np.median(res["g"][3][-4:]) < np.median(res["0"][3][-4:])


def test_host_and_device_loops_agree_across_the_pass_on_a_large_shard(monkeypatch):
    """8M points: the shard is large enough for chunks of two batches, which become chunks of
    eight after the second ordering pass (icp_hip.cc, maybe_resort).  The host-driven and the
    device-resident loop take the pass at the same iteration and change the chunks alike: same
    bits in the transform, the moved cloud and the error."""
    from icp_amd import capi, synth
    n = 8_000_000
    m = synth.cloud(n, seed=21)
    mu_m = capi.mean_host(m)
    mc = capi.center_host(m, mu_m)
    p = synth.apply_fp32(m, synth.ground_truth(0.5))
    monkeypatch.delenv("ICP_HIP_RESORT", raising=False)
    res = []
    for mode in ("device", "host"):
        monkeypatch.setenv("ICP_HIP_LOOP", mode)
        with capi.Context() as ctx:
            ctx.build_tree(mc, 256)
            ctx.set_queries(p)
            T, it, err = ctx.run(mu_m, 14, 0.0)
            res.append((T.copy(), it, err, ctx.get_points()))
    assert res[0][1] == res[1][1] == 14 and res[0][2] == res[1][2]
    assert (bits(res[0][0]) == bits(res[1][0])).all()
    assert (bits(res[0][3]) == bits(res[1][3])).all()
    assert np.abs(res[0][0].astype(np.float64) - synth.expected_transform(0.5)).max() < 1e-5


def _replay(ctx, capi, shard, n_total, mu_m, iters, rec):
    """Register `shard` alone with the recorded all-reduced sums of the whole cloud's run in place
    of its own: what it would see as one rank of many."""
    k = [0]

    def cb(a):
        a[:] = rec[k[0]]
        k[0] += 1
    ctx.set_allreduce(cb)
    ctx.set_queries(shard, n_total=n_total)
    T, it, err = ctx.run(mu_m, iters, 0.0)
    return T, it, ctx.get_points(), np.array(ctx.nn_trace())


def test_compact_shard_is_ordered_again_around_the_clouds_mean(monkeypatch, capfd):
    """A shard's set-up order is the leaf order of p - mean(shard); the loop searches
    p - mean(whole cloud), known after the first all-reduce (icp_hip.cc, recentre_order).  A
    spatially compact shard is ordered again around that mean, a random one is not; either way the
    rank sees the sums of the whole run, ends with its transform bit for bit and returns its part
    of the moved cloud in its own row order -- across the second ordering pass as well."""
    from icp_amd import capi, synth
    from icp_amd.dist import spatial_shard
    n, iters = 1_200_000, 12
    monkeypatch.setenv("ICP_HIP_RESORT", "v")
    m = synth.cloud(n, seed=31)
    mu_m = capi.mean_host(m)
    mc = capi.center_host(m, mu_m)
    p = synth.apply_fp32(m, synth.ground_truth(0.5))
    rec = []
    with capi.Context() as ctx:
        ctx.build_tree(mc, 256)
        ctx.set_allreduce(lambda a: rec.append(a.copy()))
        ctx.set_queries(p)
        T0, it0, _ = ctx.run(mu_m, iters, 0.0)
        whole = ctx.get_points()
        capfd.readouterr()
        for rows, expect in ((spatial_shard(m, 2, 1), "shard ordered again"),
                             (np.arange(n // 2), "order kept")):
            res = {}
            for mode in ("0", "v"):
                monkeypatch.setenv("ICP_HIP_RECENTRE", mode)
                res[mode] = _replay(ctx, capi, np.ascontiguousarray(p[rows]), n, mu_m, iters, rec)
                said = capfd.readouterr().err
                assert ("[recentre]" in said) == (mode == "v")
                if mode == "v":
                    assert expect in said
                # long enough, and the shard large enough, for the second ordering pass: a
                # callback-driven run takes it too, each rank for its own shard
                assert "second ordering pass" in said
            for T, it, moved, _ in res.values():
                assert it == it0 and (bits(T) == bits(T0)).all()
                assert (bits(moved) == bits(whole[rows])).all()
        # the context remembers the whole cloud's mean: the next shard of a cloud of this size is
        # ordered around it at set-up, and the run finds nothing to order again
        rows = spatial_shard(m, 2, 1)
        T, it, moved, _ = _replay(ctx, capi, np.ascontiguousarray(p[rows]), n, mu_m, iters, rec)
        assert "order kept" in capfd.readouterr().err
        assert it == it0 and (bits(T) == bits(T0)).all() and (bits(moved) == bits(whole[rows])).all()


def test_recentred_order_under_a_communicator(monkeypatch, capfd):
    """The same step in the device-resident loop of a communicator run.  One rank is all this box
    can start, so the shard is declared part of a larger cloud (n_total > n_local): the mean the
    loop forms from the all-reduced sums is then far from the shard's own, the shard is ordered
    again before the first pass, and the run equals the one without that step up to the order of
    the fp64 sums."""
    from icp_amd import capi, synth
    n = 600_000
    m = synth.cloud(n, seed=33)
    mu_m = capi.mean_host(m)
    mc = capi.center_host(m, mu_m)
    p = synth.apply_fp32(m, synth.ground_truth(0.5))[: n // 2]
    res = {}
    for mode in ("0", "v"):
        monkeypatch.setenv("ICP_HIP_RECENTRE", mode)
        with capi.Context() as ctx:
            ctx.build_tree(mc, 256)
            ctx.comm_init_rank(1, 0, capi.Context.comm_unique_id())
            ctx.set_queries(p, n_total=n)
            capfd.readouterr()
            T, it, err = ctx.run(mu_m, 3, 0.0)
            assert ("shard ordered again" in capfd.readouterr().err) == (mode == "v")
            res[mode] = (T.copy(), it, ctx.get_points())
    assert res["0"][1] == res["v"][1] == 3
    assert np.abs(res["0"][0] - res["v"][0]).max() <= 1e-5
    assert np.abs(res["0"][2] - res["v"][2]).max() <= 1e-5
