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


def test_spatially_compact_shards_match_one_context(tmp_path):
    """The partition bench.py --gpus N uses (icp_amd.dist.spatial_shard): each rank registers its
    compact part, which the library orders again around the whole cloud's mean; the parts put
    back with out[rows] = new_p are the single-context result."""
    import icp_amd
    from icp_amd import synth
    n = 400_000
    m = synth.cloud(n, seed=23)
    p = synth.apply_fp32(m, synth.ground_truth(1.0))
    T1, new_p1, info = icp_amd.icp(m, p, 30, 1e-5, 256, return_info=True)
    res = run_workers("gpu_spatial", str(tmp_path / "out"))
    out = np.full_like(new_p1, np.nan)
    seen = 0
    for r in res:
        assert r["iters"] == info["iterations"]
        assert np.abs(np.array(r["T"], np.float32).reshape(4, 4) - T1).max() <= 1e-6
        z = np.load(str(tmp_path / "out") + f".{r['rank']}.npz")
        out[z["rows"]] = z["new_p"]
        seen += len(z["rows"])
    assert seen == n and res[0]["T"] == res[1]["T"]
    assert np.abs(out - new_p1).max() <= 1e-6


def test_rows_exchanged_behind_the_abi_match_one_context(tmp_path):
    """icp_hip_set_queries_exchange: each rank hands over a contiguous slice of the (randomly
    ordered) rows, the library sends every point to the rank that owns its region of the tree,
    and icp_hip_get_points returns to every rank the rows it handed over: the single-context
    registration, up to the order of the fp64 sums."""
    import icp_amd
    from icp_amd import synth
    n = 400_000
    m = synth.cloud(n, seed=23)
    p = synth.apply_fp32(m, synth.ground_truth(1.0))
    T1, new_p1, info = icp_amd.icp(m, p, 30, 1e-5, 256, return_info=True)
    res = run_workers("gpu_exchange", str(tmp_path / "out"))
    out = np.full_like(new_p1, np.nan)
    for r in res:
        assert r["iters"] == info["iterations"]
        assert np.abs(np.array(r["T"], np.float32).reshape(4, 4) - T1).max() <= 1e-6
        z = np.load(str(tmp_path / "out") + f".{r['rank']}.npz")
        assert len(z["new_p"]) == len(z["rows"])
        out[z["rows"]] = z["new_p"]
    assert res[0]["T"] == res[1]["T"]
    assert np.abs(out - new_p1).max() <= 1e-6
    # the regions cover the cloud and are about equal (quantiles of a 32768-point sample)
    assert sum(r["queries"] for r in res) == n
    assert max(r["queries"] for r in res) <= 1.03 * n / 2


def test_empty_shard_takes_part_in_every_collective(tmp_path):
    """More ranks than points: a rank with an empty shard issues the same all-reduces and ends
    with the same transform."""
    import icp_amd
    from icp_amd import synth
    n, cap = 30011, 64
    m = synth.cloud(n, seed=21)
    p = synth.apply_fp32(m, synth.ground_truth(2.0))
    T1, new_p1, info = icp_amd.icp(m, p, 50, 1e-5, cap, return_info=True)
    res = run_workers("gpu_empty", str(tmp_path / "out"))
    assert res[0]["iters"] == res[1]["iters"] == info["iterations"]
    assert res[0]["T"] == res[1]["T"]
    # (a sharded run centres on mu_m from the second iteration on, the one-GPU run on mean(new_p):
    # equal up to the rounding of the centroid, loop_steps.hh)
    assert np.abs(np.array(res[0]["T"], np.float32).reshape(4, 4) - T1).max() <= 1e-6
    assert len(res[1]["new_p"]) == 0
    assert np.abs(np.array(res[0]["new_p"], np.float32) - new_p1).max() <= 1e-6


def test_rccl_communicator_rejects_per_rank_loop_overrides(monkeypatch):
    """With a communicator the loop choice must not depend on one rank's environment."""
    from icp_amd import capi, synth
    m = synth.cloud(5000, seed=3)
    mu_m = capi.mean_host(m)
    tree = capi.FlatTree(capi.center_host(m, mu_m), 32)
    monkeypatch.setenv("ICP_HIP_LOOP", "host")
    with capi.Context() as ctx:
        ctx.set_tree(tree)
        ctx.comm_init_rank(1, 0, capi.Context.comm_unique_id())
        ctx.set_queries(m)
        with pytest.raises(capi.IcpHipError) as e:
            ctx.run(mu_m, 5, 1e-5)
        assert e.value.status == 1 and "communicator" in str(e.value)
    tree.close()


def test_rccl_communicator_single_rank():
    """The RCCL leg (icp_hip_comm_init_rank: dlopen of librccl, ncclAllReduce of the sums on the
    context's stream) with a communicator of one rank, device-resident loop, against the same
    sharded protocol driven from the host through an all-reduce callback that has nobody to add
    (icp_hip_set_allreduce): same bits, also when the iteration cap ends the loop and for every
    block position of the end.  More ranks need more GPUs than the test box has; bench.py --gpus N
    takes this same path.  The sharded order: one all-reduce of 16 doubles per iteration, the loop
    condition one slot late, a speculative NN pass at the end, queries centred on mu_m from the
    second iteration on.  Against the one-GPU loop (centred on mean(new_p) like the reference,
    icp.cc:114-115): same iteration count, the transform within the centroid's rounding."""
    from icp_amd import capi, synth
    n, cap = 50_000, 64
    m = synth.cloud(n, seed=31)
    p = synth.apply_fp32(m, synth.ground_truth(2.0))
    mu_m = capi.mean_host(m)
    tree = capi.FlatTree(capi.center_host(m, mu_m), cap)
    out = []
    for mode in ("callback", "rccl", "one GPU"):
        with capi.Context() as ctx:
            ctx.set_tree(tree)
            if mode == "rccl":
                ctx.comm_init_rank(1, 0, capi.Context.comm_unique_id())
            elif mode == "callback":
                ctx.set_allreduce(lambda a: None)
            ctx.set_queries(p)
            T, iters, err = ctx.run(mu_m, 30, 1e-5)
            out.append((T.copy(), iters, err, ctx.get_points()))
    tree.close()
    assert out[0][1] == out[1][1] and out[0][2] == out[1][2]
    assert (out[0][0].view(np.uint32) == out[1][0].view(np.uint32)).all()
    assert (out[0][3].view(np.uint32) == out[1][3].view(np.uint32)).all()
    assert out[2][1] == out[1][1] and np.abs(out[2][0] - out[1][0]).max() <= 1e-6
    assert np.abs(out[2][3] - out[1][3]).max() <= 1e-6
    # the loop ended by the cap, at every position inside the blocks of 1, 2, 4, 8 slots
    tree = capi.FlatTree(capi.center_host(m, mu_m), cap)
    with capi.Context() as plain, capi.Context() as sharded:
        plain.set_tree(tree)
        plain.set_allreduce(lambda a: None)     # the sharded order, driven from the host
        sharded.set_tree(tree)
        sharded.comm_init_rank(1, 0, capi.Context.comm_unique_id())
        for cap_iters in (0, 1, 2, 3, 4, 7, 8, 15, 16):
            res = []
            for ctx in (plain, sharded):
                ctx.set_queries(p)
                errs = []
                T, iters, err = ctx.run(mu_m, cap_iters, 0.0, progress=lambda i, mx, e: errs.append((i, e)))
                res.append((T.copy(), iters, err, errs, ctx.get_points(), ctx.timing()))
            assert res[0][1] == res[1][1] == cap_iters
            assert res[0][3] == res[1][3] and [i for i, _ in res[0][3]] == list(range(1, cap_iters + 1))
            assert (res[0][0].view(np.uint32) == res[1][0].view(np.uint32)).all()
            assert (res[0][4].view(np.uint32) == res[1][4].view(np.uint32)).all()
            for k in ("nn_launches", "queries", "nodes_visited", "points_scanned"):
                assert res[0][5][k] == res[1][5][k], k
            # what the loop rests on: the centroid of the moved cloud sits on mu_m
            if cap_iters:
                assert np.abs(res[0][4].mean(0, dtype=np.float64) - mu_m).max() <= 3e-7
    tree.close()


def test_bench_flow_with_two_ranks_sharing_the_gpu():
    """bench.py --gpus 2 end to end on the one GPU of the box (ICP_BENCH_SHARE_GPU=1: both ranks on
    GPU 0, gloo all-reduce through the host callback, because RCCL refuses a second rank on a
    device): it starts its ranks, hands them compact shards of the larger cloud, registers and
    prints one line for the job."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, ICP_BENCH_SHARE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "custom",
                        "--points", "400000", "--steps", "8", "--warmup", "1", "--cpu-sample", "0"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 8 and j["scaling"] == "strong"
    assert j["config"]["queries_total"] == 400000 and j["config"]["queries_per_gpu"] == 200000
    assert "spatial_shard" in j["config"]["shards"] and "REHEARSAL" in j["config"]["allreduce"]
    assert j["check"]["max_abs_T_minus_expected"] < 2e-5
    assert abs(j["value"] - 400000 * 8 / (j["ms_per_step"] * 8e-3)) < 1e-3 * j["value"]
    # what makes a first N > 1 run diagnosable: one record per rank, a collective latency probe,
    # and a roofline block that never carries a fraction it did not get from counters
    assert [r["rank"] for r in j["per_rank"]] == [0, 1]
    for r in j["per_rank"]:
        assert r["queries"] == 200000 and r["nn_ms_per_step"] > 0 and r["apply_ms_per_step"] > 0
        assert r["wall_ms_per_step"] <= j["ms_per_step"] * 1.001
        assert r["nn_ms_first_iter"] > 0 and r["nn_ms_last_iter"] > 0
    assert j["allreduce_probe"]["us_per_allreduce_16_doubles"] > 0
    rf = j["roofline"]
    assert rf["bound"] == "valu_issue" and rf["avg_launch_ms"] > 0
    assert rf["frac"] is None and rf["traffic"] is None and "no committed counter passes" in rf["note"]


def test_exchange_over_rccl_with_one_rank(monkeypatch):
    """The ncclSend / ncclRecv leg of icp_hip_set_queries_exchange (staging buffers in HBM, one
    group of sends and receives): a communicator of one rank exchanging with itself
    (ICP_HIP_EXCHANGE_SELF=1) -- all this box can start.  Nothing moves, so the registration and
    the rows that come back are those of icp_hip_set_queries under the same communicator, bit for
    bit; per-query outputs of icp_hip_nn are refused on the exchanged shard."""
    from icp_amd import capi, synth
    n = 120_000
    m = synth.cloud(n, seed=37)
    p = synth.apply_fp32(m, synth.ground_truth(1.5))
    mu_m = capi.mean_host(m)
    mc = capi.center_host(m, mu_m)
    out = []
    for exchange in (False, True):
        with capi.Context() as ctx:
            ctx.build_tree(mc, 64)
            ctx.comm_init_rank(1, 0, capi.Context.comm_unique_id())
            if exchange:
                monkeypatch.setenv("ICP_HIP_EXCHANGE_SELF", "1")
                ctx.set_queries_exchange(p, n_total=n)
                assert ctx.shard_size() == (n, n)
                with pytest.raises(capi.IcpHipError):
                    ctx.nn(mu_m)
                ctx.nn(mu_m, want=())
                ctx.set_queries_exchange(p, n_total=n)
            else:
                ctx.set_queries(p)
            T, iters, err = ctx.run(mu_m, 20, 1e-5)
            out.append((T.copy(), iters, err, ctx.get_points()))
    assert out[0][1] == out[1][1] and out[0][2] == out[1][2]
    assert (out[0][0].view(np.uint32) == out[1][0].view(np.uint32)).all()
    assert (out[0][3].view(np.uint32) == out[1][3].view(np.uint32)).all()


def test_bench_rows_exchange_partition():
    """bench.py --gpus 2 --partition rows-exchange on the shared GPU: contiguous row slices in,
    compact regions of about equal size registered, the same transform."""
    import json
    r = _bench_two_ranks({}, ("--partition", "rows-exchange"))
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert "icp_hip_set_queries_exchange" in j["config"]["shards"]
    sizes = [p["queries"] for p in j["per_rank"]]
    assert sum(sizes) == 100000 and max(sizes) <= 52000
    assert j["check"]["max_abs_T_minus_expected"] < 5e-5


def _bench_two_ranks(extra_env, extra_args=()):
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, ICP_BENCH_SHARE_GPU="1", **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload",
                           "custom", "--points", "100000", "--steps", "4", "--warmup", "1",
                           "--cpu-sample", "0", *extra_args],
                          env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)


def test_bench_exits_nonzero_when_a_communicator_fails():
    """A rank whose RCCL communicator does not come up must fail the JOB (every rank, non-zero
    status, no JSON line) unless --allow-fallback is given; with it the line names the fallback.
    The failure is simulated on rank 1 (ICP_BENCH_TEST_COMM_FAILURE) on the shared GPU."""
    import json
    r = _bench_two_ranks({"ICP_BENCH_TEST_COMM_FAILURE": "1"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "did not come up on every rank" in r.stderr and "--allow-fallback" in r.stderr
    r = _bench_two_ranks({"ICP_BENCH_TEST_COMM_FAILURE": "1"}, ("--allow-fallback",))
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert j["config"]["allreduce"].startswith("FALLBACK") and len(j["per_rank"]) == 2
    assert j["check"]["max_abs_T_minus_expected"] < 5e-5
