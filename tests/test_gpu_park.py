"""The straggler lots of nn_tile_kernel (nn_tile.hip): the last lanes at work in a batch move their
traversal state to a lot in HBM and are run later, a batch of them together, their terms summed
from late rows.  A query's traversal is untouched by it -- the same comparisons in the same order
(libcpu/libcpu/vp-tree.cc:125-175), only in another lane, later -- so with the lots forced on
(ICP_HIP_PARK=always; by default they serve the aligned passes of shards of 6M points and more)
every answer, distance and work counter must be the oracle's bit for bit, the sums must not
depend on which wave ran what, and the registration must be the one without lots up to the order
of the fp64 sums."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu

N = 1_000_000


@pytest.fixture(scope="module")
def world():
    from icp_amd import capi, synth
    m = synth.cloud(N, seed=41)
    mu_m = capi.mean_host(m)
    mc = capi.center_host(m, mu_m)
    p0 = synth.apply_fp32(m, synth.ground_truth(1.0))                             # a point spacing off
    p1 = synth.apply_fp32(p0, synth.expected_transform(1.0).astype(np.float32))   # aligned
    return dict(m=m, mu_m=mu_m, mc=mc, p0=p0, p1=p1)


@pytest.mark.parametrize("park_max,late,chunk", [(16, 8, 0), (4, 1, 8), (32, 32, 1), (1, 1, 2)])
@pytest.mark.parametrize("which", ["p0", "p1"])
def test_parked_queries_get_the_oracles_answers(world, oracle, monkeypatch, which, park_max, late, chunk):
    from icp_amd import capi
    monkeypatch.setenv("ICP_HIP_PARK", "always")
    monkeypatch.setenv("ICP_HIP_PARK_MAX", str(park_max))
    monkeypatch.setenv("ICP_HIP_LATE_PER_BATCH", str(late))
    if chunk:
        monkeypatch.setenv("ICP_HIP_CHUNK_BATCHES", str(chunk))
    p = world[which]
    mu_p = capi.mean_host(p)
    with capi.Context() as ctx:
        ctx.build_tree(world["mc"], 256)
        ctx.set_queries(p)
        got = ctx.nn(mu_p)
        tm = ctx.timing()
        dev = ctx.get_tree()
    q = oracle.subtract(p, mu_p)
    want = oracle.tree_closest(oracle.tree_from_flat(dev), q, counters=True)
    assert (got["row"] == want["row"]).all()
    assert (bits(got["dist"]) == bits(want["dist"])).all()
    assert (bits(got["xyz"]) == bits(want["xyz"])).all()
    assert tm["nodes_visited"] == float(want["n_int"].sum())
    assert tm["points_scanned"] == float(want["n_pts"].sum())
    dev.close()


def test_sums_do_not_depend_on_which_wave_ran_what(world, oracle, monkeypatch):
    """The twelve sums of a pass with the lots on: the exact sum of the same fp32 products
    (libcpu/libcpu/point-3d.cc:220-231) up to fp64 rounding, and the same BITS from run to run --
    which wave runs which chunk, and when it empties its lot, changes with every launch.  The sums
    are taken where a sharded run hands them to its all-reduce (icp_hip_set_allreduce)."""
    from icp_amd import capi
    monkeypatch.setenv("ICP_HIP_PARK", "always")
    monkeypatch.setenv("ICP_HIP_PARK_MAX", "16")
    monkeypatch.setenv("ICP_HIP_LATE_PER_BATCH", "8")
    runs = []
    with capi.Context() as ctx:
        ctx.build_tree(world["mc"], 256)
        for _ in range(3):
            rec = []
            ctx.set_allreduce(lambda a: rec.append(a.copy()))
            ctx.set_queries(world["p0"])
            ctx.run(world["mu_m"], 4, 0.0)
            runs.append(rec)
        dev = ctx.get_tree()
    for other in runs[1:]:
        assert len(other) == len(runs[0])
        for x, y in zip(runs[0], other):
            assert (x.view(np.uint64) == y.view(np.uint64)).all()
    # the first slot of 16 doubles is the NN pass over p0 - mean(p0)
    first = [r for r in runs[0] if len(r) == 16][0]
    mu_p = (np.array([r for r in runs[0] if len(r) == 4][0][1:4]) / N).astype(np.float32)
    q = oracle.subtract(world["p0"], mu_p)
    want = oracle.tree_closest(oracle.tree_from_flat(dev), q, counters=True)
    H = (q[:, :, None] * want["xyz"][:, None, :]).astype(np.float32).astype(np.float64).sum(0).reshape(9)
    assert np.allclose(first[:9], H, rtol=1e-12, atol=1e-9)
    assert first[9] == float(want["n_int"].sum()) and first[10] == float(want["n_pts"].sum())
    dev.close()


def test_registration_with_lots_equals_the_one_without(world, monkeypatch):
    """Whole loops: lots forced on against lots off -- same iteration count, the transform and the
    moved cloud equal up to the order of the fp64 sums; host-driven and device-resident loop
    bit-identical with the lots on."""
    from icp_amd import capi
    res = {}
    for park, loop in (("never", "device"), ("always", "device"), ("always", "host")):
        monkeypatch.setenv("ICP_HIP_PARK", park)
        monkeypatch.setenv("ICP_HIP_LOOP", loop)
        with capi.Context() as ctx:
            ctx.build_tree(world["mc"], 256)
            ctx.set_queries(world["p0"])
            errs = []
            T, it, err = ctx.run(world["mu_m"], 12, 0.0, progress=lambda i, mx, e: errs.append(e))
            res[(park, loop)] = (T.copy(), it, err, ctx.get_points(), errs)
    a, b, c = res[("never", "device")], res[("always", "device")], res[("always", "host")]
    assert a[1] == b[1] == c[1] == 12
    assert np.abs(a[0] - b[0]).max() <= 2e-7 and np.abs(a[3] - b[3]).max() <= 1e-6
    assert (bits(b[0]) == bits(c[0])).all() and (bits(b[3]) == bits(c[3])).all() and b[4] == c[4]
