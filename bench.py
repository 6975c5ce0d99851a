#!/usr/bin/env python3
"""Headline benchmark: ICP over a VP-tree on MI355X, one process per GPU.

Metric (BASELINE.json): NN queries/s (+ ICP iterations/s) on synthetic clouds.

  workload "64M" (default)  config 5 of BASELINE.json: a 64,000,000-point uniform
                            reference cloud (tree replicated on every GPU, capacity 256);
                            the cloud to move is the same cloud under a known rigid motion; its
                            64M points are split into N spatially compact shards of equal
                            size (strong scaling: 8 GPUs x 8M points).
  workload "1M"             config 4: 1,000,000 points (roofline characterisation on one GPU).

A "step" is ONE full ICP iteration over the rank's shard, exactly the reference loop body
(libcpu/libcpu/icp.cc:110-125): NN of every centred point + 3x3 cross-covariance (one fused
kernel), solve, apply + error (one fused kernel); with N > 1 ONE all-reduce of 16 doubles per
iteration over RCCL.  Nothing is skipped and nothing is cached between steps.  The timed region
is K consecutive iterations FROM THE INITIAL misaligned clouds (threshold 0 so the loop cannot
exit early); the W warm-up iterations run the same thing first and the shard is then re-uploaded
(untimed), so inputs are resident in HBM when the clock starts.

Launch: `python bench.py --gpus N` starts its N ranks itself (fresh child processes, made before
anything in this process has touched a GPU); under torchrun (WORLD_SIZE set) it is one of the
ranks.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
SIMDS = 256 * 4            # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9           # max clock
VALU_ISSUE_CYCLES = 2.0    # a wave64 VALU instruction occupies its SIMD-32 for 2 cycles


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="64M", choices=["64M", "1M", "custom"])
    ap.add_argument("--points", type=int, default=0, help="custom: reference cloud size")
    ap.add_argument("--queries-total", type=int, default=0,
                    help="register only the first Q points of the cloud (default: all of it)")
    ap.add_argument("--capacity", type=int, default=256)
    ap.add_argument("--angle", type=float, default=None,
                    help="rigid-motion angle in degrees (default 1.0 at <= 8M points, 0.25 above)")
    ap.add_argument("--cpu-sample", type=int, default=16_000_000,
                    help="queries of the CPU-baseline sample (0 disables the baseline)")
    ap.add_argument("--cpu-repeats", type=int, default=0,
                    help="CPU-baseline repeats (default: 5 at <= 4M points, 3 above; median taken)")
    ap.add_argument("--partition", default="spatial", choices=["spatial", "rows-exchange"],
                    help="N > 1: how a rank gets its shard.  spatial: the caller cuts the cloud "
                         "into compact parts itself (icp_amd.dist.spatial_shard, host numpy).  "
                         "rows-exchange: every rank hands over a contiguous slice of the (randomly "
                         "ordered) rows and the library re-partitions behind the ABI "
                         "(icp_hip_set_queries_exchange)")
    ap.add_argument("--allow-fallback", action="store_true",
                    help="N > 1: if the RCCL communicator does not come up on every rank, let all "
                         "ranks fall back to the host-callback all-reduce instead of exiting non-zero")
    ap.add_argument("--rehearse", action="store_true",
                    help="launcher rehearsal without a GPU: start the ranks, rendezvous over gloo, "
                         "split the shards, print the line's launch fields and exit")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------------
# launcher: `bench.py --gpus N` without a launcher around it starts N ranks
# --------------------------------------------------------------------------------------------
def launch_ranks(n: int) -> int:
    """Start n fresh rank processes of this script (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in their
    environment) and wait for them.  This process never touches a GPU and imports no torch."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)]
                                      + sys.argv[1:], env=env))
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {r} exited with status {code}; stopping the others",
                      file=sys.stderr, flush=True)
                for q in alive:   # the ranks we started, by handle: they would hang in a collective
                    procs[q].terminate()
        time.sleep(0.05)
    return rc if rc >= 0 else 1


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def load_counters(args, n_m, n_queries_total):
    """profiles/nn_counters_<workload>.json if it belongs to THIS command, else None.  The file
    holds the per-launch means of the builder's rocprofv3 --pmc passes; its key is workload, timed
    steps, capacity, angle, query count, one GPU.  The warm-up count is not part of the key:
    set_queries puts the shard back to the initial clouds after the warm-up, so the timed
    launches are the same whatever W was."""
    path = os.path.join(ROOT, "profiles", f"nn_counters_{args.workload}.json")
    try:
        with open(path) as f:
            cj = json.load(f)
        if cj.get("workload") == args.workload and cj.get("n_gpus", 1) == 1 and \
                cj.get("steps") == args.steps and \
                int(cj.get("capacity", 256)) == args.capacity and \
                abs(float(cj.get("angle", 0.25 if n_m > 8_000_000 else 1.0)) - args.angle) < 1e-12 and \
                int(cj.get("queries", n_m)) == n_queries_total and \
                float(cj["valu_insts_per_launch"]) > 0 and float(cj["hbm_bytes_per_launch"]) > 0:
            return cj
    except Exception:
        pass
    return None


def rehearse(args, rank, world):
    """The launch path without a GPU: rendezvous (gloo), shard split, a sum over the ranks."""
    import torch
    import torch.distributed as dist
    from icp_amd.dist import shard_bounds
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    n = args.queries_total or {"64M": 64_000_000, "1M": 1_000_000}.get(args.workload, args.points)
    b, e = shard_bounds(n, world, rank)
    t = torch.tensor([float(e - b), 1.0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"rehearsal": True, "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ranks_seen": int(t[1].item()),
                          "queries_total": int(t[0].item()),
                          "config": {"queries_per_gpu": e - b,
                                     "parallelism": f"query-shard x{world}"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: be the launcher (before any torch / HIP call in this process)
        raise SystemExit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as "
                         f"`python bench.py --gpus N` or under a launcher with --nproc-per-node N")
    if args.rehearse:
        return rehearse(args, rank, world)

    import numpy as np
    import torch  # device plumbing + torch.distributed only
    import torch.distributed as dist
    # ICP_BENCH_SHARE_GPU=1: every rank on GPU 0, sums exchanged over gloo through the host
    # callback -- the whole N-rank flow (partition, shards of a larger cloud, one all-reduce per
    # iteration) on a box with ONE GPU, where RCCL refuses a second rank.  Not a measurement: the
    # line says so in config.allreduce.
    share_gpu = world > 1 and os.environ.get("ICP_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank}, but this node has "
                         f"{torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)
    if share_gpu:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from icp_amd import capi, synth
    from icp_amd.dist import init_rccl, spatial_shard

    def barrier_sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------- workload ----------------
    # Strong scaling: the problem (reference cloud + the whole cloud to move) is fixed and its
    # points are split into `world` shards of equal size.  The reference algorithm takes the
    # translation from the two clouds' centroids (t = mu_m - R mu_p, icp.cc:57), so it only
    # converges to the exact motion when the cloud to move is the WHOLE transformed cloud; a
    # per-GPU subset (weak scaling) would leave a centroid-sampling bias in t.
    if args.workload == "64M":
        n_m = 64_000_000
    elif args.workload == "1M":
        n_m = 1_000_000
    else:
        n_m = args.points or 1_000_000
    if args.angle is None:
        # the ICP basin shrinks with the point spacing (0.02 at 1M, 0.005 at 64M in [-1,1)^3):
        # 1 degree converges in ~5 iterations at 1M but not within 80 at 64M; 0.25 degree does.
        args.angle = 0.25 if n_m > 8_000_000 else 1.0
    scaling = "strong"
    n_queries_total = args.queries_total or n_m
    t0 = time.time()
    m = synth.cloud(n_m)
    gt = synth.ground_truth(args.angle)
    if world == 1:
        begin, end = 0, n_queries_total
        p_shard = synth.apply_fp32(m[begin:end], gt)
        shards = "the whole cloud"
    else:
        # The synthetic cloud's rows are in random order: a slice of rows is a random subset,
        # the worst partition for a tree search (every rank stages every bucket: 1.5-1.75x its
        # share of the one-GPU pass).  A caller partitions a scan by position; so does the bench
        # (icp_amd.dist.spatial_shard: median planes, equal sizes, the same partition on every
        # rank).  Set-up, like the tree build: outside the timed region.
        if args.partition == "spatial":
            rows = spatial_shard(m[:n_queries_total], world, rank)
            begin, end = 0, len(rows)
            p_shard = synth.apply_fp32(m[rows], gt)
            shards = "spatially compact parts of equal size (median planes, icp_amd.dist.spatial_shard)"
            del rows
        else:
            from icp_amd.dist import shard_bounds
            begin, end = shard_bounds(n_queries_total, world, rank)
            p_shard = synth.apply_fp32(m[begin:end], gt)
            shards = ("contiguous slices of the randomly ordered rows, re-partitioned behind the ABI "
                      "into compact regions (icp_hip_set_queries_exchange)")
    t_gen = time.time() - t0

    t0 = time.time()
    mu_m = capi.mean_host(m)
    mc = capi.center_host(m, mu_m)
    t_center = time.time() - t0

    # the reference's tree, built in HBM (icp_hip_build_tree; bit-identical to the host build)
    ctx = capi.Context(local_rank)
    # The per-iteration all-reduce: the context's own RCCL communicator, enqueued on its stream
    # between the kernels (xGMI).  More than one RCCL rank has never run where this was built
    # (one GPU per box).  If the communicator does not come up on EVERY rank the job exits
    # non-zero: a line measured on the other loop must not pass for the RCCL one.  With
    # --allow-fallback all ranks switch together to the host-callback all-reduce through
    # torch.distributed (the path the 2-rank tests cover), and the line says so.
    allreduce_path = "none (one rank)"
    # ICP_BENCH_TEST_COMM_FAILURE=<rank>: that rank pretends its communicator failed (tests of
    # the decision below on a one-GPU box, together with ICP_BENCH_SHARE_GPU=1)
    simulate = os.environ.get("ICP_BENCH_TEST_COMM_FAILURE")
    if world > 1 and (not share_gpu or simulate is not None):
        allreduce_path = "rccl communicator on the context's stream"
        try:
            if simulate is not None and int(simulate) == rank:
                raise RuntimeError("simulated communicator failure (ICP_BENCH_TEST_COMM_FAILURE)")
            if not share_gpu:
                init_rccl(ctx, world, rank)
            ok = 1.0
        except Exception as e:  # capi.IcpHipError, or the rendezvous
            print(f"[bench rank {rank}] RCCL communicator failed: {e}", file=sys.stderr, flush=True)
            ok = 0.0
        flag = torch.tensor([ok], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if flag.item() < 1.0 and not args.allow_fallback:
            ctx.close()
            dist.destroy_process_group()
            raise SystemExit(f"bench.py: rank {rank}: the RCCL communicator did not come up on "
                             f"every rank (see the messages above); --allow-fallback runs the "
                             f"host-callback all-reduce instead")
        if flag.item() < 1.0:
            from icp_amd.dist import torch_allreduce
            ctx.close()
            ctx = capi.Context(local_rank)
            ctx.set_allreduce(torch_allreduce(
                device=None if share_gpu else torch.device("cuda", local_rank)))
            allreduce_path = "FALLBACK: torch.distributed all-reduce from the host loop"
            share_gpu_cb = False
        else:
            share_gpu_cb = share_gpu
    else:
        share_gpu_cb = share_gpu
    if share_gpu_cb:
        from icp_amd.dist import torch_allreduce
        ctx.set_allreduce(torch_allreduce())
        allreduce_path = "REHEARSAL on one shared GPU: gloo all-reduce from the host loop"
    ctx_uses_callback = share_gpu_cb or allreduce_path.startswith("FALLBACK")
    ctx.build_tree(mc[:4096], args.capacity)   # first-call costs (code load) out of the timing
    t0 = time.time()
    ctx.build_tree(mc, args.capacity)
    t_build = time.time() - t0
    tinfo = ctx.tree_info()

    # achievable streaming bandwidth of this GPU beside the 8 TB/s spec peak: a plain
    # device-to-device copy of 1 GiB through torch (plumbing), bytes read + written per second
    copy_gbs = None
    if rank == 0:
        src = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 5 * 2.0 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst
        torch.cuda.empty_cache()

    # ---------------- warm-up ----------------
    def upload():
        if world > 1 and args.partition == "rows-exchange":
            ctx.set_queries_exchange(p_shard, n_total=n_queries_total)
        else:
            ctx.set_queries(p_shard, n_total=n_queries_total)
    if world > 1 and args.partition == "rows-exchange" and ctx_uses_callback:
        # (host-callback runs move the points over the process group too; an RCCL communicator
        # moves them itself, ncclSend / ncclRecv)
        from icp_amd.dist import torch_alltoallv
        ctx.set_alltoallv(world, rank, torch_alltoallv(
            device=None if share_gpu else torch.device("cuda", local_rank)))
    upload()
    if args.warmup > 0:
        ctx.run(mu_m, args.warmup, 0.0)
    upload()  # back to the initial clouds; resets timers
    shard_points = ctx.shard_size()[0]

    # ---------------- timed region: exactly K iterations ----------------
    barrier_sync()
    t0 = time.perf_counter()
    T, iters, err = ctx.run(mu_m, args.steps, 0.0)
    barrier_sync()
    dt = time.perf_counter() - t0
    my_dt = dt
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert iters == args.steps

    tm = ctx.timing()
    trace = ctx.nn_trace()

    # correctness guard: the recovered transform is the inverse of the ground truth
    want = synth.expected_transform(args.angle)
    t_err = float(np.abs(T.astype(np.float64) - want).max())

    # ---------------- roofline of the dominant kernel (NN + covariance) ----------------
    # Measured by this run: the kernel's launch time (HIP events on the library's stream) and its
    # work counters (internal nodes visited, bucket points scanned; equal to the reference
    # traversal's, tests/test_gpu_nn_parity.py and tests/test_gpu_64m.py compare them with the
    # oracle's).  NOT measured by this run: the hardware counters (VALU instructions, HBM bytes);
    # they come from the committed rocprofv3 --pmc passes of this same command on this kernel
    # (profiles/nn_counters_<workload>.json, scripts/profile_round.sh) and are labelled so.
    #
    # What binds the kernel is VALU issue: the shard is stored in leaf order, the queries of a
    # wave share buckets, a bucket is staged once in LDS and scanned by (queries x bucket parts)
    # tiles; `frac` = VALU wave-instructions x 2 cycles / (launch time x 1024 SIMDs x 2.4 GHz).
    # SURVEY 8d's per-query byte model (16 B per query and per node visited, 12 B per bucket point
    # scanned) is kept as `algorithmic_GBs`: it charges every query for bucket bytes that ~50
    # queries share, so it exceeds the HBM peak and is not a fraction of anything.
    launches = max(1, int(tm["nn_launches"]))
    # the loop's work counters ride on the all-reduce: they are the JOB's totals on every rank
    alg_bytes_all = 16.0 * tm["queries"] + 16.0 * tm["nodes_visited"] + 12.0 * tm["points_scanned"]
    bytes_per_launch = alg_bytes_all / launches / world       # one GPU's launch (mean over ranks)
    avg_ms = tm["nn_ms"] / launches
    avg_s = avg_ms * 1e-3
    n_gpu = float(shard_points)
    cpath = os.path.join("profiles", f"nn_counters_{args.workload}.json")
    counters = load_counters(args, n_m, n_queries_total)
    tree_bytes = 12.0 * tinfo["n_slots"] + 32.0 * tinfo["n_nodes"]
    compulsory = 16.0 * n_gpu + min(tree_bytes, bytes_per_launch)
    flops_per_launch = 8.0 * (tm["points_scanned"] + tm["nodes_visited"]) / launches / world
    flop_peak = 256 * 128 * CLOCK_HZ / 1e12    # fp32 lanes x clock: the rate without FMA
    flop_ach = flops_per_launch / avg_s / 1e12
    issue_peak = SIMDS * CLOCK_HZ / VALU_ISSUE_CYCLES / 1e9   # G wave-instructions / s
    roofline = {"kernel": "nn_tile_kernel", "avg_launch_ms": round(avg_ms, 4), "launches": launches,
                "bound": "valu_issue", "achieved": None, "peak": round(issue_peak, 1),
                "unit": "G wave-instr/s", "frac": None,
                "traffic": None, "hbm": None, "counters_from": None,
                "reference_arithmetic": {
                    "TFLOP/s": round(flop_ach, 2), "fp32_no_fma_peak": round(flop_peak, 2),
                    "note": "8 flops per distance of the REFERENCE traversal (points_scanned and "
                            "nodes_visited count what libcpu scans; the ring walk examines 1-2 of "
                            "a bucket's points once the clouds are aligned): a work rate, not a "
                            "utilisation of this kernel"},
                "algorithmic_GBs": round(bytes_per_launch / avg_s / 1e9, 1),
                "algorithmic_bytes_per_launch": round(bytes_per_launch),
                "bytes_per_query": round(alg_bytes_all / max(1.0, tm["queries"]), 1),
                "compulsory_bytes_per_launch": round(compulsory),
                "stack_spills": tm.get("stack_spills", 0.0),
                "measured_copy_GBs": round(copy_gbs, 1) if copy_gbs else None}
    if counters:
        # N == 1: the file's per-launch means are this command's.  N > 1: rank 0 runs its shard of
        # the same cloud against the same tree; instructions and bytes are taken PER QUERY from
        # the one-GPU passes and multiplied by the shard's queries -- an estimate, labelled so
        # (a compact shard costs 1.0-1.15x its share of the one-GPU pass).
        share = n_gpu / float(counters.get("queries", n_m))
        valu = float(counters["valu_insts_per_launch"]) * share
        traffic = float(counters["hbm_bytes_per_launch"]) * share
        ach = valu / avg_s / 1e9
        src = (f"{cpath}: rocprofv3 --pmc passes of this command by the builder "
               f"({counters.get('tag', '?')}), not measured by this run")
        if world > 1:
            src += (f"; scaled by rank 0's share of the queries ({share:.4f}): an ESTIMATE for a "
                    f"shard, the passes were taken on one GPU")
        roofline.update({
            "achieved": round(ach, 1), "frac": round(ach / issue_peak, 4),
            "traffic": round(traffic),
            "hbm": {"achieved": round(traffic / avg_s / 1e9, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(traffic / avg_s / 1e9 / HBM_PEAK_GBS, 4),
                    "frac_of_measured_copy": (round(traffic / avg_s / 1e9 / copy_gbs, 4)
                                              if copy_gbs else None),
                    "traffic_over_compulsory": round(traffic / compulsory, 2)},
            "counters_from": src})
    else:
        # no counter file for this command: no utilisation figure is invented for it
        roofline["note"] = (f"no committed counter passes match this command ({cpath}: workload, "
                            f"steps, capacity, angle, queries): frac / traffic are null")

    # ---------------- per-rank record (N > 1: gathered after the timed region) ----------------
    mine = {"rank": rank, "queries": int(shard_points), "wall_ms_per_step": round(my_dt / args.steps * 1e3, 4),
            "nn_ms_per_step": round(tm["nn_ms"] / launches, 4),
            "apply_ms_per_step": round(tm["apply_ms"] / launches, 4),
            "nn_ms_first_iter": round(float(trace[0]), 4) if len(trace) else None,
            "nn_ms_last_iter": round(float(trace[-1]), 4) if len(trace) else None,
            "setup_s": {"generate_and_partition": round(t_gen, 2), "tree_build": round(t_build, 3)}}
    per_rank, allreduce_probe = [mine], None
    if world > 1:
        # what ONE all-reduce of the loop's 16 doubles costs through this process group (the
        # library's own communicator runs the same collective on its stream): 50 in a row
        probe_dev = "cpu" if share_gpu else "cuda"
        buf = torch.zeros(16, dtype=torch.float64, device=probe_dev)
        for _ in range(5):
            dist.all_reduce(buf)
        barrier_sync()
        tp = time.perf_counter()
        for _ in range(50):
            dist.all_reduce(buf)
        if probe_dev == "cuda":
            torch.cuda.synchronize()
        allreduce_probe = {"us_per_allreduce_16_doubles": round((time.perf_counter() - tp) / 50 * 1e6, 2),
                           "through": "torch.distributed " + ("gloo (host)" if share_gpu else "nccl (RCCL)")}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered

    out = None
    if rank == 0:
        value = n_queries_total * args.steps / dt
        out = {
            "metric": "icp_nn_queries_per_sec", "value": round(value, 1), "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic-{args.workload}: {n_m} reference points, VP-tree "
                                   f"capacity {args.capacity} replicated per GPU, "
                                   f"{n_queries_total} points to register "
                                   f"({end - begin} per GPU), rigid motion {args.angle} deg",
                       "reference_points": n_m, "queries_total": n_queries_total,
                       "queries_per_gpu": end - begin, "capacity": args.capacity,
                       "tree_height": int(tinfo["height"]), "parallelism": f"query-shard x{world}",
                       "shards": shards,
                       "allreduce": allreduce_path},
            "icp_iterations_per_sec": round(args.steps / dt, 3),
            "nn_kernel_queries_per_sec": round(n_queries_total / avg_s, 1),
            "nn_ms_first_iter": round(float(trace[0]), 4) if len(trace) else None,
            "nn_ms_last_iter": round(float(trace[-1]), 4) if len(trace) else None,
            "nn_queries_per_sec_first_iter": (round(n_queries_total / (float(trace[0]) * 1e-3), 1)
                                              if len(trace) and trace[0] > 0 else None),
            "nn_queries_per_sec_last_iter": (round(n_queries_total / (float(trace[-1]) * 1e-3), 1)
                                             if len(trace) and trace[-1] > 0 else None),
            "phase_ms_per_step": {"nn_cov": round(tm["nn_ms"] / launches, 4),
                                  "apply_error": round(tm["apply_ms"] / launches, 4),
                                  "host_loop": round(tm["host_ms"] / launches, 4)},
            "setup_s": {"generate": round(t_gen, 2), "center": round(t_center, 2),
                        "tree_build_device_incl_h2d": round(t_build, 3)},
            "check": {"max_abs_T_minus_expected": t_err, "final_error": float(err)},
            "roofline": roofline,
        }
        if world > 1:
            out["per_rank"] = per_rank
            out["allreduce_probe"] = allreduce_probe

    # ---------------- CPU baseline beside it (rank 0, N == 1 only) ----------------
    # One libcpu iteration on a bounded sample of the shard against the same tree, on the box's
    # host cores: median of >= 3 repeats (>= 5 at 1M), on the geometry of iteration 0 (`value`,
    # next to nn_ms_first_iter) and on the converged geometry (next to nn_ms_last_iter).
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        from oracle import Oracle, Ref  # the checker, timed as the baseline -- never the product
        orc = Oracle()
        s = min(args.cpu_sample, end - begin)
        reps = args.cpu_repeats or (5 if n_m <= 4_000_000 else 3)
        sample = np.ascontiguousarray(p_shard[:s])
        aligned = synth.apply_fp32(sample, want.astype(np.float32))  # the converged cloud
        use_ref = Ref.available() and n_m <= 4_000_000
        # Where the reference's own builder finishes in seconds (<= 4M points; 71 s at 64M) the
        # baseline is the reference itself: its libcpu sources compiled unmodified (oracle/_ref,
        # prebuilt in the build container) do the tree, mean, subtract, closest and
        # find_covariance; the Eigen-dependent solve and the two streaming loops behind it come
        # from the port.  Above that the port (oracle/, bit-identical answers) does all of it.
        t0 = time.perf_counter()
        if use_ref:
            ref = Ref()
            tree = ref.tree_build(args.capacity, mc)
            eng, closest = ref, (lambda q: ref.tree_closest(tree, q))
        else:
            ftree = capi.FlatTree(mc, args.capacity)   # host build of the same tree
            tree = orc.tree_from_flat(ftree)
            eng, closest = orc, (lambda q: orc.tree_closest(tree, q)["xyz"])
        t_tree = time.perf_counter() - t0

        def one_iteration(cloud):
            t0 = time.perf_counter()
            mu_p = eng.mean(cloud)
            q = eng.subtract(cloud, mu_p)
            t1 = time.perf_counter()
            y = closest(q)
            t2 = time.perf_counter()
            H = eng.find_covariance(q, y)
            Tn = orc.alignment_from_cov(H, mu_p, mu_m)
            moved = orc.apply_alignment(cloud, Tn)
            orc.compute_error(y, moved, mu_m)
            t3 = time.perf_counter()
            return t3 - t0, t2 - t1

        first = [one_iteration(sample) for _ in range(reps)]
        last = [one_iteration(aligned) for _ in range(reps)]
        it_first, nn_first = median([a for a, _ in first]), median([b for _, b in first])
        it_last, nn_last = median([a for a, _ in last]), median([b for _, b in last])
        threads = orc.max_threads()
        out["cpu_baseline"] = {
            "value": round(s / it_first, 1), "unit": "queries/s", "cores": threads,
            "cpu_model": cpu_model(), "kind": "reference" if use_ref else "port",
            "repeats": reps,
            "sample": f"one ICP iteration of "
                      f"{'the reference libcpu (compiled unmodified)' if use_ref else 'the libcpu port (oracle/)'}"
                      f" over the first {s} points of the shard against the same {n_m}-point tree, "
                      f"median of {reps}: closest() on {threads} OpenMP threads, mean / subtract / "
                      f"covariance / solve / apply / error serial as in the reference; `value` is "
                      f"on the iteration-0 geometry",
            "nn_queries_per_sec_first_iter": round(s / nn_first, 1),
            "nn_queries_per_sec_last_iter": round(s / nn_last, 1),
            "iteration_queries_per_sec_last_iter": round(s / it_last, 1),
            "tree_build_s": round(t_tree, 2)}
        out["setup_s"]["tree_build_cpu_baseline"] = round(t_tree, 2)
        del tree
    if rank == 0:
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
