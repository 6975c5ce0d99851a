"""N > 1 path on CPU: world_size 2 over gloo.

Checks shard bounds, the all-reduce callback the C ABI calls (icp_hip_set_allreduce), and that
the sharded decomposition of an iteration (per-shard fp64 partial sums -> all-reduce -> one SVD
solve) reproduces the single-process loop, and that bench.py starts its own ranks.  The oracle stands in for the kernels here; the
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


def test_two_rank_iterations_match_single_process(tmp_path, oracle):
    res = run_workers("cpu", str(tmp_path / "out"))
    n, cap = 30011, 64
    m = synth.cloud(n, seed=21)
    p = synth.apply_fp32(m, synth.ground_truth(2.0))
    assert [r["shard"] for r in res] == [[0, 15006], [15006, 30011]]
    # one set-up collective (the first centroid), then ONE of 16 doubles per iteration
    assert res[0]["calls"] == res[1]["calls"] == [4, 16, 16, 16]
    # both ranks hold identical reduced values
    for k in ("covs", "T", "errors"):
        assert res[0][k] == res[1][k], k
    # single-process run of the same loop with the same arithmetic
    mu_m = oracle.mean(m)
    tree = oracle.tree_build(cap, oracle.subtract(m, mu_m))
    mu_p = (p.astype(np.float64).sum(0) / n).astype(np.float32)
    mu_q, cur, T_tot = mu_p.copy(), p.copy(), np.eye(4, dtype=np.float32)
    for it in range(3):
        q = oracle.subtract(cur, mu_q)
        nn = oracle.tree_closest(tree, q, counters=True)
        rows = np.concatenate([np.array(r["rows"][it], np.uint32) for r in res])
        assert (rows == nn["row"]).all()  # NN answers do not depend on the sharding
        cov = (q[:, :, None] * nn["xyz"][:, None, :]).astype(np.float32).astype(np.float64).sum(0).reshape(9)
        got = np.array(res[0]["covs"][it])
        assert np.allclose(got[:9], cov, rtol=1e-12, atol=1e-9)
        assert got[9] == nn["n_int"].sum() and got[10] == nn["n_pts"].sum()
        Tn = capi.alignment_host(got[:9].astype(np.float32), mu_p, mu_m)
        T_tot = oracle.matmul4(Tn, T_tot)
        cur = oracle.apply_alignment(cur, Tn)
        mu_q = mu_m.copy()
        if it < 2:
            d = (nn["xyz"] + mu_m - cur).astype(np.float32)
            err = np.float32(((d * d).astype(np.float32)).astype(np.float64).sum())
            assert np.isclose(res[0]["errors"][it], err, rtol=1e-6)
            mu_p = (cur.astype(np.float64).sum(0) / n).astype(np.float32)
            # the identity the loop rests on: the moved centroid sits on mu_m
            assert np.abs(mu_p - mu_m).max() <= 2e-7
        # the reference-order fp32 covariance agrees to summation accuracy
        assert np.allclose(got[:9], oracle.find_covariance(q, nn["xyz"]), rtol=1e-4, atol=1e-3)
    assert (np.array(res[0]["T"], np.float32).reshape(4, 4) == T_tot).all()
    # the same loop converges like libcpu's (error trajectory of oracle.icp within fp32 noise)
    cpu = oracle.icp(m, p, 50, 1e-5, cap)
    assert np.allclose(res[0]["errors"], cpu["errors"][:2], rtol=1e-3)


def test_bench_launcher_starts_its_ranks():
    """`python bench.py --gpus 2` with no launcher around it must run TWO ranks (the launch path,
    rehearsed without a GPU: rendezvous over gloo, shard split, a sum over the ranks)."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse",
                        "--workload", "1M"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2
    assert line["queries_total"] == 1_000_000 and line["config"]["queries_per_gpu"] == 500_000
    assert line["config"]["parallelism"] == "query-shard x2"
    # one rank, unchanged
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--rehearse",
                        "--workload", "1M"], env=env, capture_output=True, text=True, timeout=300)
    line = json.loads(r.stdout.splitlines()[-1])
    assert r.returncode == 0 and line["n_gpus"] == 1
    # a launcher that started a different number of ranks is an error, not a silent n_gpus: 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--rehearse"],
                       env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_bench_launcher_reports_a_failing_rank():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    # no GPU here: every rank fails loudly, and so does the launcher
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload",
                        "1M", "--steps", "1", "--warmup", "0"], env=env, capture_output=True,
                       text=True, timeout=300)
    if r.returncode == 0:
        pytest.skip("a GPU node with >= 2 devices: the real launch ran")
    assert "exited with status" in r.stderr or "needs GPU" in r.stderr


def test_spatial_shard_partition():
    """icp_amd.dist.spatial_shard: the parts of all ranks are disjoint, cover the cloud, differ in
    size by at most one point per cut, and are compact (a part's bounding box holds about its
    share of the cloud's volume, where a random subset's box is the whole cloud)."""
    from icp_amd import synth
    from icp_amd.dist import spatial_shard
    for n, world in ((10_000, 1), (10_001, 2), (9_999, 3), (40_000, 8), (5, 8), (0, 4)):
        xyz = synth.cloud(n, seed=5) if n else np.zeros((0, 3), np.float32)
        parts = [spatial_shard(xyz, world, r) for r in range(world)]
        allrows = np.concatenate(parts)
        assert len(allrows) == n and (np.sort(allrows) == np.arange(n)).all()
        sizes = [len(q) for q in parts]
        assert max(sizes) - min(sizes) <= 3
        assert all((np.diff(q) > 0).all() for q in parts)
        if n >= 10_000 and world > 1:
            vol = np.prod(xyz.max(0) - xyz.min(0))
            for q in parts:
                box = np.prod(xyz[q].max(0) - xyz[q].min(0))
                assert box < 1.6 * vol / world
    with pytest.raises(ValueError):
        spatial_shard(np.zeros((4, 3), np.float32), 2, 2)


def test_bench_counter_file_matches_the_driver_command():
    """The driver runs `bench.py --gpus 1 --steps 20 --warmup 5`: the committed counter passes
    (profiles/nn_counters_64M.json) must be accepted for it whatever the warm-up count, and
    rejected for another command (steps, capacity, angle, query count)."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for warm in ("0", "3", "5", "11"):
        a = bench.parse(["--gpus", "1", "--steps", "20", "--warmup", warm])
        a.angle = 0.25
        cj = bench.load_counters(a, 64_000_000, 64_000_000)
        assert cj is not None and cj["valu_insts_per_launch"] > 0 and cj["hbm_bytes_per_launch"] > 0
    for argv, nq in ((["--steps", "10"], 64_000_000), (["--capacity", "128"], 64_000_000),
                     (["--steps", "20"], 8_000_000)):
        a = bench.parse(argv)
        a.angle = 0.25
        assert bench.load_counters(a, 64_000_000, nq) is None
    a = bench.parse(["--steps", "20"])
    a.angle = 1.0
    assert bench.load_counters(a, 64_000_000, 64_000_000) is None
