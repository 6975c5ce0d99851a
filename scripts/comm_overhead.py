"""What the slot structure of a communicator run costs per iteration: the same registration with
and without an RCCL communicator of one rank (its all-reduce has nobody to talk to, but it is
enqueued, and the loop step runs in a launch of its own after it).
usage: python scripts/comm_overhead.py [n] [iters]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from icp_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
m = synth.cloud(n); mu_m = capi.mean_host(m); mc = capi.center_host(m, mu_m)
p = synth.apply_fp32(m, synth.ground_truth(0.25 if n > 8_000_000 else 1.0))
for use_comm in (False, True, False, True):
    with capi.Context() as ctx:
        ctx.build_tree(mc, 256)
        if use_comm:
            ctx.comm_init_rank(1, 0, capi.Context.comm_unique_id())
        ctx.set_queries(p); ctx.run(mu_m, 3, 0.0); ctx.set_queries(p)
        t0 = time.perf_counter()
        T, it, err = ctx.run(mu_m, iters, 0.0)
        dt = time.perf_counter() - t0
        tm = ctx.timing()
        print(f"{'communicator of one rank' if use_comm else 'no communicator':26s} {dt / it * 1e3:.4f} ms per iteration "
              f"(NN {tm['nn_ms'] / it:.4f}, apply {tm['apply_ms'] / it:.4f})", flush=True)
