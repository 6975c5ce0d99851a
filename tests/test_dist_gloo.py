"""N > 1 path on CPU: world_size 2 over gloo.

Checks shard bounds, the all-reduce callback the C ABI calls (icp_hip_set_allreduce), and that
the sharded decomposition of an iteration (per-shard fp64 partial sums -> all-reduce -> one SVD
solve) reproduces the single-process iteration.  The oracle stands in for the kernels here; the
same worker runs the real kernels in tests/test_gpu_sharded.py.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from icp_amd import capi, synth
from icp_amd.dist import shard_bounds


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_workers(mode, out, world=2, timeout=300):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"),
                                       mode, out], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [json.load(open(f"{out}.{r}")) for r in range(world)]


def test_shard_bounds_partition():
    for n in (0, 1, 7, 64, 1000, 30011):
        for world in (1, 2, 3, 4, 8):
            b = [shard_bounds(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert all(0 <= e - s for s, e in b)


def test_two_rank_iteration_matches_single_process(tmp_path, oracle):
    res = run_workers("cpu", str(tmp_path / "out"))
    n, cap = 30011, 64
    m = synth.cloud(n, seed=21)
    p = synth.apply_fp32(m, synth.ground_truth(2.0))
    assert [r["shard"] for r in res] == [[0, 15006], [15006, 30011]]
    # both ranks hold identical reduced values
    for k in ("mu_p", "cov", "T", "err"):
        assert res[0][k] == res[1][k], k
    # single-process iteration with the same arithmetic
    mu_m = oracle.mean(m)
    mc = oracle.subtract(m, mu_m)
    tree = oracle.tree_build(cap, mc)
    mu_p = (p.astype(np.float64).sum(0) / n).astype(np.float32)
    assert np.allclose(res[0]["mu_p"], mu_p, rtol=0, atol=1e-7)
    q = oracle.subtract(p, np.array(res[0]["mu_p"], np.float32))
    nn = oracle.tree_closest(tree, q, counters=True)
    rows = np.concatenate([np.array(r["rows"], np.uint32) for r in res])
    assert (rows == nn["row"]).all()  # NN answers do not depend on the sharding
    cov = (q[:, :, None] * nn["xyz"][:, None, :]).astype(np.float32).astype(np.float64).sum(0).reshape(9)
    assert np.allclose(res[0]["cov"][:9], cov, rtol=1e-12, atol=1e-9)
    assert res[0]["cov"][9] == nn["n_int"].sum() and res[0]["cov"][10] == nn["n_pts"].sum()
    T = capi.alignment_host(np.array(res[0]["cov"][:9], np.float32), np.array(res[0]["mu_p"], np.float32), mu_m)
    assert (np.array(res[0]["T"], np.float32).reshape(4, 4) == T).all()
    # and the reference-order fp32 iteration agrees to summation accuracy
    H32 = oracle.find_covariance(q, nn["xyz"])
    assert np.allclose(res[0]["cov"][:9], H32, rtol=1e-4, atol=1e-3)
