"""Ragged and smallest inputs through the C ABI, and the error conventions of the entry points.

The reference tests no edge sizes; the cases here are the ones its code defines:
a tree that is one bucket (size < capacity, vp-tree.cc:56), a node of exactly `capacity` points
(internal: the leaf test is strict), clouds that do not fill a wave, a batch or a chunk, clouds of
different sizes on the two sides."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_m,n_p,cap", [
    (1, 1, 3), (2, 5, 3), (3, 1, 4), (3, 2, 3), (8, 8, 8), (9, 700, 8), (64, 63, 32), (64, 65, 64),
    (257, 129, 256), (256, 1, 256), (1000, 1, 16), (1000, 7, 16), (1000, 513, 16), (5000, 4097, 64),
    (70_000, 3, 256), (3, 70_000, 3),
])
def test_ragged_sizes_match_oracle(oracle, n_m, n_p, cap):
    from icp_amd import capi, synth
    m = synth.cloud(n_m, seed=10 * n_m + cap)
    p = synth.cloud(n_p, seed=77 * n_p + 1) * np.float32(0.9)
    mu_m = capi.mean_host(m)
    mc = capi.center_host(m, mu_m)
    ot = oracle.tree_build(cap, mc)
    if oracle.tree_degenerate(ot):
        pytest.skip("the reference's build is undefined for this cloud")
    mu_p = oracle.mean(p)
    want = oracle.tree_closest(ot, oracle.subtract(p, mu_p))
    with capi.Context() as ctx:
        ctx.build_tree(mc, cap)
        ctx.set_queries(p)
        out = ctx.nn(mu_p)
        assert (out["row"] == want["row"]).all()
        assert (bits(out["dist"]) == bits(want["dist"])).all()
        assert (bits(out["xyz"]) == bits(want["xyz"])).all()
        # and the whole loop runs on it: 3 iterations unless the error reaches exactly 0 first
        # (error > threshold, icp.cc:110; a 1-point cloud lands on its match in one step)
        T, it, err = ctx.run(mu_m, 3, 0.0)
        assert (it == 3 or (1 <= it < 3 and err == 0.0)) and np.isfinite(T).all() and np.isfinite(err)
        assert ctx.get_points().shape == (n_p, 3)


def test_call_order_and_argument_errors(clouds):
    from icp_amd import capi
    m, p = clouds["cow_ref"], clouds["cow_tr1"]
    mu_m = capi.mean_host(m)
    with capi.Context() as ctx:
        with pytest.raises(capi.IcpHipError) as e:
            ctx.run(mu_m, 5, 1e-5)              # no tree, no cloud
        assert e.value.status == 1
        ctx.build_tree(capi.center_host(m, mu_m), 32)
        with pytest.raises(capi.IcpHipError) as e:
            ctx.nn(np.zeros(3, np.float32))     # no cloud yet
        assert e.value.status == 1
        with pytest.raises(capi.IcpHipError) as e:
            ctx.set_queries(p, n_total=len(p) - 1)   # n_total < n_local
        assert e.value.status == 1
        with pytest.raises(capi.IcpHipError) as e:
            ctx.icp(m, np.zeros((0, 3), np.float32), 5, 1e-5, 32)   # empty cloud
        assert e.value.status == 1
        with pytest.raises(capi.IcpHipError) as e:
            ctx.build_tree(np.zeros((500, 3), np.float32), 8)   # every distance ties: UB in the reference
        assert e.value.status == 2 and "degenerate" in str(e.value)
        # the context is still good after every refusal
        T, new_p, it, err = ctx.icp(m, p, 200, 1e-5, 256)
        assert it == 1 and np.abs(new_p - m).max() < 5e-3
    with pytest.raises(capi.IcpHipError):
        capi.Context(device_id=4096)


@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf])
@pytest.mark.parametrize("n", [300, 5000, 60_000])
def test_non_finite_coordinates_are_refused(bad, n):
    """A NaN or infinite coordinate makes every comparison of the reference's search false: each
    query then walks the whole tree (and one NaN in the cloud to move makes its mean, hence every
    query, NaN).  The reference would grind through n x n distances; here both clouds are checked
    up front, whichever way they come in (host build, device build, sorted or unsorted upload)."""
    from icp_amd import capi, synth
    m = synth.cloud(n, seed=n)
    spoiled = m.copy()
    spoiled[n // 3, 1] = bad
    mu_m = capi.mean_host(m)
    mc = capi.center_host(m, mu_m)
    with capi.Context() as ctx:
        with pytest.raises(capi.IcpHipError) as e:
            ctx.build_tree(capi.center_host(spoiled, mu_m), 32)
        assert e.value.status == 1 and "non-finite" in str(e.value)
        with pytest.raises(capi.IcpHipError) as e:
            capi.FlatTree(capi.center_host(spoiled, mu_m), 32)
        assert e.value.status == 1 and "non-finite" in str(e.value)
        ctx.build_tree(mc, 32)
        with pytest.raises(capi.IcpHipError) as e:
            ctx.set_queries(spoiled)
        assert e.value.status == 1 and "non-finite" in str(e.value)
        with pytest.raises(capi.IcpHipError) as e:
            ctx.icp(m, spoiled, 5, 1e-5, 32)
        assert e.value.status == 1
        T, new_p, it, err = ctx.icp(m, m, 5, 1e-5, 32)   # the context is still good
        assert it == 1 and err <= 1e-5


@pytest.mark.parametrize("n", [100, 5000])
def test_finite_cloud_with_an_fp32_overflowing_extent_is_accepted(n):
    """Finite coordinates whose extent overflows fp32 (hi - lo = inf) are legal input at every
    size: the set-up order of the larger shards must not mistake them for non-finite ones."""
    from icp_amd import capi, synth
    m = synth.cloud(2000, seed=3)
    p = synth.cloud(n, seed=4)
    p[0] = [3.0e38, -3.0e38, 3.0e38]
    p[1] = [-3.0e38, 3.0e38, -3.0e38]
    mu_m = capi.mean_host(m)
    tree = capi.FlatTree(capi.center_host(m, mu_m), 32)
    with capi.Context() as ctx:
        ctx.set_tree(tree)
        ctx.set_queries(p)   # must not raise
        out = ctx.nn(np.zeros(3, np.float32), want=("row",))
    assert out["row"].shape == (n,) and (out["row"] < len(m)).all()
    tree.close()


@pytest.mark.parametrize("n", [3000, 20000, 700000])
def test_call_orders_other_than_tree_then_queries(n):
    """The shard uploaded before there is a tree (no leaf order, no second ordering pass), the
    tree replaced under a shard that stays, a second run that continues from the moved cloud:
    the registration is the one of the usual order (small, unsorted, sorted and re-sorted sizes)."""
    import icp_amd
    from icp_amd import capi, synth
    m = synth.cloud(n, seed=1)
    p = synth.apply_fp32(m, synth.ground_truth(1.0))
    mu_m = capi.mean_host(m)
    mc = capi.center_host(m, mu_m)
    T0, p0, info = icp_amd.icp(m, p, 30, 1e-5, 64, return_info=True)
    with capi.Context() as ctx:
        ctx.set_queries(p)
        ctx.build_tree(mc, 64)
        T, it, err = ctx.run(mu_m, 30, 1e-5)
        assert it == info["iterations"] and np.abs(T - T0).max() <= 1e-6
        assert np.abs(ctx.get_points() - p0).max() <= 1e-6
        ctx.build_tree(mc, 16)
        ctx.set_queries(p)
        T, it, err = ctx.run(mu_m, 30, 1e-5)
        assert it == info["iterations"] and np.abs(T - T0).max() <= 1e-6
        T, it, err = ctx.run(mu_m, 5, 0.0)
        assert it == 5 and np.abs(T - np.eye(4, dtype=np.float32)).max() <= 1e-5
        assert np.abs(ctx.get_points() - m).max() < 1e-4
