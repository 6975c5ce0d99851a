"""N > 1 path on the real kernels: two processes share the one GPU of the box, each with its own
context and half of the cloud; the sums are exchanged through the C ABI's all-reduce callback
(gloo).  The result must match the single-context run: identical iteration count, transform
within 1e-6 (fp64 partial sums differ only in summation order), moved points within 1e-6."""
import numpy as np
import pytest

from test_dist_gloo import run_workers

pytestmark = pytest.mark.gpu


def test_two_contexts_match_one(tmp_path):
    import icp_amd
    from icp_amd import synth
    n, cap = 30011, 64
    m = synth.cloud(n, seed=21)
    p = synth.apply_fp32(m, synth.ground_truth(2.0))
    T1, new_p1, info = icp_amd.icp(m, p, 50, 1e-5, cap, return_info=True)
    res = run_workers("gpu", str(tmp_path / "out"))
    for r in res:
        assert r["iters"] == info["iterations"]
        assert np.abs(np.array(r["T"], np.float32).reshape(4, 4) - T1).max() <= 1e-6
    assert res[0]["T"] == res[1]["T"]
    new_p = np.concatenate([np.array(r["new_p"], np.float32) for r in res])
    assert new_p.shape == new_p1.shape and np.abs(new_p - new_p1).max() <= 1e-6
