"""GPU tree build (icp_hip_build_tree) against the host build (icp_hip_tree_build), which
tests/test_host_tree.py pins to the reference's compiled libcpu tree bit for bit: every array of
the flat layout must be identical -- node centres, radii, child codes, parents, rows, leaf table,
bucket contents and order (including the far-point swaps), padding."""
import numpy as np
import pytest

from conftest import bits, flat_preorder, golden_nn, tree_sha256

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def device_build(monkeypatch):
    """icp_hip_build_tree hands clouds below 8192 points to the host builder; here the device
    builder is the thing under test (the library reads the variable at icp_hip_create)."""
    monkeypatch.setenv("ICP_HIP_TREE_BUILD", "device")

FLOAT_FIELDS = ["node_cx", "node_cy", "node_cz", "node_radius", "slot_x", "slot_y", "slot_z"]
INT_FIELDS = ["node_inside", "node_outside", "node_parent", "node_row", "leaf_begin", "leaf_count",
              "slot_row"]
SCALARS = ["n_points", "n_slots", "n_nodes", "n_leaves", "height", "capacity", "max_leaf_count",
           "root"]


def same_tree(a, b):
    for k in SCALARS:
        assert getattr(a, k) == getattr(b, k), k
    for k in FLOAT_FIELDS:
        assert (bits(getattr(a, k)) == bits(getattr(b, k))).all(), k
    for k in INT_FIELDS:
        assert (np.asarray(getattr(a, k)) == np.asarray(getattr(b, k))).all(), k


def centred(capi, m):
    return capi.center_host(m, capi.mean_host(m))


@pytest.mark.parametrize("name,cap", [("line_ref", 8), ("line_ref", 32), ("cow_ref", 8),
                                      ("cow_ref", 32), ("cow_ref", 256), ("horse_ref", 32),
                                      ("horse_ref", 256)])
def test_reference_clouds(clouds, name, cap):
    from icp_amd import capi
    mc = centred(capi, clouds[name])
    host = capi.FlatTree(mc, cap)
    with capi.Context() as ctx:
        ctx.build_tree(mc, cap)
        dev = ctx.get_tree()
    same_tree(dev, host)
    dev.validate()
    # directly against the reference: fingerprint of the tree its own compiled libcpu builds
    pname = {"line_ref": "line_translated_2_3_4", "cow_ref": "cow_tr1", "horse_ref": "horse_tr1"}[name]
    g = golden_nn(f"nn_{pname}_cap{cap}")
    assert tree_sha256(flat_preorder(dev)) == str(g["tree_sha256"])
    assert dev.height == int(g["height"]) and dev.n_leaves == int(g["leaves"])
    dev.close()
    host.close()


@pytest.mark.parametrize("n,cap", [(1000, 3), (5000, 8), (20_000, 33), (100_000, 64),
                                   (200_000, 257), (300_000, 1024), (1_000_000, 256),
                                   (2_000_000, 32)])
def test_synthetic(n, cap):
    from icp_amd import capi, synth
    mc = centred(capi, synth.cloud(n, seed=n ^ cap))
    host = capi.FlatTree(mc, cap)
    with capi.Context() as ctx:
        ctx.build_tree(mc, cap)
        dev = ctx.get_tree()
    same_tree(dev, host)
    dev.close()
    host.close()


def test_set_tree_round_trip(clouds):
    """icp_hip_get_tree returns what icp_hip_set_tree uploaded."""
    from icp_amd import capi
    mc = centred(capi, clouds["cow_ref"])
    host = capi.FlatTree(mc, 32)
    with capi.Context() as ctx:
        ctx.set_tree(host)
        back = ctx.get_tree()
    same_tree(back, host)
    back.close()
    host.close()


def test_degenerate_and_argument_errors():
    from icp_amd import capi
    pts = np.zeros((100, 3), np.float32)  # all points equal: every split leaves a side empty
    with capi.Context() as ctx:
        with pytest.raises(capi.IcpHipError) as e:
            ctx.build_tree(pts, 8)
        assert e.value.status == 2
        with pytest.raises(capi.IcpHipError) as e:
            ctx.build_tree(np.random.rand(50, 3).astype(np.float32), 2)
        assert e.value.status == 2
    with pytest.raises(capi.IcpHipError) as e:
        capi.FlatTree(pts, 8)
    assert e.value.status == 2


def test_registration_on_a_device_built_tree(clouds, oracle):
    """The whole loop on a device-built tree: same answer as on the host-built one."""
    from icp_amd import capi
    m, p = clouds["horse_ref"], clouds["horse_tr1"]
    mu_m = capi.mean_host(m)
    mc = capi.center_host(m, mu_m)
    res = []
    for dev in (False, True):
        with capi.Context() as ctx:
            if dev:
                ctx.build_tree(mc, 32)
            else:
                t = capi.FlatTree(mc, 32)
                ctx.set_tree(t)
                t.close()
            ctx.set_queries(p)
            T, it, err = ctx.run(mu_m, 200, 1e-5)
            res.append((T.copy(), it, err))
    assert res[0][1] == res[1][1] == 16
    assert (bits(res[0][0]) == bits(res[1][0])).all() and res[0][2] == res[1][2]


def test_small_clouds_take_the_host_builder_and_give_the_same_tree(clouds, monkeypatch):
    from icp_amd import capi
    mc = centred(capi, clouds["cow_ref"])
    trees = []
    for mode in ("device", "host", None):
        if mode is None:
            monkeypatch.delenv("ICP_HIP_TREE_BUILD")
        else:
            monkeypatch.setenv("ICP_HIP_TREE_BUILD", mode)
        with capi.Context() as ctx:
            ctx.build_tree(mc, 32)
            trees.append(ctx.get_tree())
    same_tree(trees[0], trees[1])
    same_tree(trees[0], trees[2])
    for t in trees:
        t.close()


@pytest.mark.parametrize("cap", [4, 16, 64])
def test_ties_at_a_median_take_the_level_by_level_build(cap):
    """Below 4M points the device build is speculative: the host lays out every level assuming
    that floor(m / 2) points fall strictly inside each median sphere and checks the counts once
    at the end (tree_build.hip).  On a lattice many points are equally far from a vantage point,
    the assumption fails somewhere, and the build must notice and start over level by level: the
    tree is still the host builder's (the reference's, vp-tree.cc:48-52,:92) array by array --
    or the same degenerate-cloud error where the reference's build is undefined."""
    from icp_amd import capi
    g = np.arange(-12, 13, dtype=np.float32)
    lattice = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    rng = np.random.default_rng(cap)
    # a little anisotropy keeps most splits non-degenerate while leaving many exact ties
    m = (lattice * np.array([1.0, 1.5, 2.0], np.float32))[rng.permutation(len(lattice))]
    mc = centred(capi, m)
    try:
        host = capi.FlatTree(mc, cap)
    except capi.IcpHipError as e:
        with capi.Context() as ctx, pytest.raises(capi.IcpHipError) as d:
            ctx.build_tree(mc, cap)
        assert d.value.status == e.status == 2
        return
    with capi.Context() as ctx:
        ctx.build_tree(mc, cap)
        dev = ctx.get_tree()
    same_tree(dev, host)
    dev.close()
    host.close()
