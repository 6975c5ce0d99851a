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


def test_rccl_communicator_single_rank():
    """The RCCL leg (icp_hip_comm_init_rank: dlopen of librccl, ncclAllReduce of the sums on the
    context's stream) with a communicator of one rank: same bits as the run without it.  More
    ranks need more GPUs than the test box has; bench.py --gpus N takes this same path."""
    from icp_amd import capi, synth
    n, cap = 50_000, 64
    m = synth.cloud(n, seed=31)
    p = synth.apply_fp32(m, synth.ground_truth(2.0))
    mu_m = capi.mean_host(m)
    tree = capi.FlatTree(capi.center_host(m, mu_m), cap)
    out = []
    for use_comm in (False, True):
        with capi.Context() as ctx:
            ctx.set_tree(tree)
            if use_comm:
                ctx.comm_init_rank(1, 0, capi.Context.comm_unique_id())
            ctx.set_queries(p)
            T, iters, err = ctx.run(mu_m, 30, 1e-5)
            out.append((T.copy(), iters, err, ctx.get_points()))
    tree.close()
    assert out[0][1] == out[1][1] and out[0][2] == out[1][2]
    assert (out[0][0].view(np.uint32) == out[1][0].view(np.uint32)).all()
    assert (out[0][3].view(np.uint32) == out[1][3].view(np.uint32)).all()
