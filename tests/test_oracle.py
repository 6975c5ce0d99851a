"""CPU suite, part 1: the oracle (oracle/icp_oracle.cc) pinned against
  (a) the golden fixtures = outputs of the reference's own compiled libcpu,
  (b) the compiled reference itself when oracle/_ref is present (build container),
  (c) the reference's own unit assertions (libcpu/libcpu/test_point-3d_utils.cc,
      utils/test_matrix.cc) restated as known-answer tests,
  (d) numpy for the SVD step (Eigen is third-party and absent: "parity unpinned" at the last bit).
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, NN_GOLDEN, bits, golden_nn


@pytest.mark.parametrize("mname,pname,cap", NN_GOLDEN)
def test_oracle_matches_reference_golden(oracle, clouds, mname, pname, cap):
    g = golden_nn(f"nn_{pname}_cap{cap}")
    m, p = clouds[mname], clouds[pname]
    mu_m, mu_p = oracle.mean(m), oracle.mean(p)
    assert (bits(mu_m) == bits(g["mu_m"])).all() and (bits(mu_p) == bits(g["mu_p"])).all()
    mc, q = oracle.subtract(m, mu_m), oracle.subtract(p, mu_p)
    t = oracle.tree_build(cap, mc)
    st = oracle.tree_stats(t)
    assert st["height"] == int(g["height"]) and st["leaves"] == int(g["leaves"])
    import hashlib
    d = oracle.tree_dump(t)
    h = hashlib.sha256()
    for k in ("kind", "node_f", "leaf_count", "leaf_xyz"):
        h.update(np.ascontiguousarray(d[k]).tobytes())
    assert h.hexdigest() == str(g["tree_sha256"])
    out = oracle.tree_closest(t, q, counters=True)
    assert (out["row"] == g["row"]).all()
    assert (bits(out["dist"]) == bits(g["dist"])).all()
    assert (out["n_pts"] == g["n_pts"]).all() and (out["n_int"] == g["n_int"]).all()
    assert (bits(oracle.find_covariance(q, out["xyz"])) == bits(g["cov"])).all()


@pytest.mark.parametrize("mname,pname,cap", [c for c in NN_GOLDEN if c[0] != "horse_ref"][:6]
                         + [("horse_ref", "horse_tr2", 256)])
def test_oracle_matches_compiled_reference(oracle, ref, clouds, mname, pname, cap):
    m, p = clouds[mname], clouds[pname]
    assert (bits(ref.mean(m)) == bits(oracle.mean(m))).all()
    mu = ref.mean(m)
    mc = ref.subtract(m, mu)
    assert (bits(mc) == bits(oracle.subtract(m, mu))).all()
    q = ref.subtract(p, ref.mean(p))
    rt, ot = ref.tree_build(cap, mc), oracle.tree_build(cap, mc)
    rd, od = ref.tree_dump(rt), oracle.tree_dump(ot)
    for k in rd:
        assert rd[k].shape == od[k].shape and (rd[k].view(np.uint8) == od[k].view(np.uint8)).all()
    y = ref.tree_closest(rt, q)
    mine = oracle.tree_closest(ot, q)
    assert (bits(y) == bits(mine["xyz"])).all()
    assert (bits(ref.tree_search_dist(rt, q)) == bits(mine["dist"])).all()
    assert (bits(ref.find_covariance(q, y)) == bits(oracle.find_covariance(q, y))).all()


def test_reference_files_parse_like_fixtures(ref, clouds):
    """The committed .npy clouds are the reference's read_csv output."""
    for name, rel in [("cow_ref", "cow/cow_ref.txt"), ("line_scaled_2_3_4", "line/line_scaled_2_3_4.txt")]:
        path = "/root/reference/data/" + rel
        if not os.path.exists(path):
            pytest.skip("reference data not present")
        assert (bits(ref.read_csv(path)) == bits(clouds[name])).all()


def test_small_helpers_match_compiled_reference(oracle, ref):
    rng = np.random.default_rng(3)
    for _ in range(20):
        a, b = rng.standard_normal((4, 4)).astype(np.float32), rng.standard_normal((4, 4)).astype(np.float32)
        assert (bits(ref.matmul4(a, b)) == bits(oracle.matmul4(a, b))).all()


# ---- the reference's own unit assertions, restated (test_point-3d_utils.cc) ----
def test_known_answers_from_reference_unit_tests(oracle):
    # squared_distance((1,1,1),(2,2,2)) == 3 (:15-21) -> distance sqrt(3) through the tree search
    t = oracle.tree_build(8, np.array([[2, 2, 2]], np.float32))
    out = oracle.tree_closest(t, np.array([[1, 1, 1]], np.float32))
    assert out["dist"][0] == np.sqrt(np.float32(3))
    # mean{1,2,3} = 2 (:52-61)
    a = np.array([[1, 1, 1], [2, 2, 2], [3, 3, 3]], np.float32)
    assert np.allclose(oracle.mean(a), 2.0, rtol=0, atol=1e-6)
    # subtract (:63-78)
    assert (oracle.subtract(a, [1, 1, 1]) == a - 1).all()
    # closest(v, v) == v (:35-50)
    t = oracle.tree_build(8, a)
    assert (oracle.tree_closest(t, a)["row"] == np.arange(3)).all()
    # dot(eye, a) == a (:90-98, matrix test :dot with identity)
    eye = np.eye(4, dtype=np.float32)
    b = np.arange(16, dtype=np.float32).reshape(4, 4)
    assert (oracle.matmul4(eye, b) == b).all() and (oracle.matmul4(b, eye) == b).all()


def test_leaf_tie_goes_to_lowest_slot(oracle):
    """Duplicate points: strict < keeps the first slot (vp-tree.cc:136)."""
    pts = np.array([[0, 0, 0], [1, 0, 0], [1, 0, 0], [5, 5, 5]], np.float32)
    t = oracle.tree_build(32, pts)
    out = oracle.tree_closest(t, np.array([[1.1, 0, 0], [0.5, 0, 0]], np.float32))
    assert out["row"][0] == 1
    assert out["row"][1] == 0  # equidistant from rows 0 and 1: the lower slot wins


def test_degenerate_clouds_are_flagged(oracle):
    pts = np.zeros((100, 3), np.float32)  # all points equal: empty inside set (vp-tree.cc:114 UB)
    assert oracle.tree_degenerate(oracle.tree_build(8, pts))
    with pytest.raises(RuntimeError):
        oracle.icp(pts, pts, 5, 1e-5, 8)


# ---- the SVD step against numpy (Eigen absent) ----
def test_svd_against_numpy(oracle):
    rng = np.random.default_rng(11)
    for i in range(300):
        H = (rng.standard_normal((3, 3)) * 10.0 ** rng.uniform(-3, 3)).astype(np.float32)
        U, S, V = oracle.svd3(H)
        Sn = np.linalg.svd(H.astype(np.float64), compute_uv=False)
        assert np.allclose(S, Sn, rtol=2e-5, atol=1e-6 * Sn[0])
        assert np.allclose((U * S) @ V.T, H, rtol=0, atol=3e-6 * Sn[0])
        if Sn[2] > 1e-3 * Sn[0]:  # well-conditioned: V U^T is the unique polar factor
            Un, _, Vtn = np.linalg.svd(H.astype(np.float64))
            T = oracle.alignment_from_cov(H, np.zeros(3), np.zeros(3))
            assert np.abs(T[:3, :3] - (Un @ Vtn).T).max() < 5e-5 * Sn[0] / Sn[2]


def test_svd_on_captured_covariances(oracle, clouds):
    """First-iteration H of horse / cow (golden, from the reference's find_covariance)."""
    for name in ("nn_horse_tr1_cap32", "nn_cow_tr2_cap256", "nn_cow_tr1_cap256"):
        g = golden_nn(name)
        H = g["cov"].reshape(3, 3)
        Un, Sn, Vtn = np.linalg.svd(H.astype(np.float64))
        T = oracle.alignment_from_cov(H, g["mu_p"], g["mu_m"])
        assert np.abs(T[:3, :3] - (Un @ Vtn).T).max() < 2e-6
        assert np.linalg.det(T[:3, :3].astype(np.float64)) == pytest.approx(1.0, abs=1e-5)


def test_rank_one_covariance_gives_reflection_without_fix(oracle, clouds):
    """line: H has rank 1; like the reference there is no det<0 correction (icp.cc:46-55)."""
    g = golden_nn("nn_line_translated_2_3_4_cap8")
    T = oracle.alignment_from_cov(g["cov"], g["mu_p"], g["mu_m"])
    R = T[:3, :3].astype(np.float64)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-5)
    d = np.ones(3) / np.sqrt(3)
    assert np.allclose(R @ d, d, atol=1e-5)  # the line direction is preserved


def test_icp_expected_json_is_reproducible(oracle, clouds):
    with open(os.path.join(GOLDEN, "icp_expected.json")) as f:
        exp = json.load(f)
    for key in ("cow_tr1_cap256", "cow_tr2_cap256", "line_translated_2_3_4_cap8"):
        e = exp[key]
        res = oracle.icp(clouds[e["m"]], clouds[e["p"]], 200, 1e-5, e["capacity"])
        assert res["iterations"] == e["iterations"]
        assert [int(v) for v in bits(res["T"]).reshape(-1)] == e["T_bits"]
        # the reference's functional assertion (test_functionnal.cc:14-18)
        assert np.abs(res["new_p"] - clouds[e["m"]]).max() < 5e-3
        assert np.abs(oracle.apply_alignment(clouds[e["p"]], res["T"]) - clouds[e["m"]]).max() < 5e-3
