"""NN-kernel timing at 64M points (device-built tree): the misaligned first pass and the aligned
pass (queries = the reference cloud itself).  ICP_HIP_TILE_STATS=1 prints the kernel's event counts.
usage: python scripts/nn64m.py [n] [capacity] [angle]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from icp_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ang = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
m = synth.cloud(n); mu_m = capi.mean_host(m); mc = capi.center_host(m, mu_m)
p = synth.apply_fp32(m, synth.ground_truth(ang))
mu_p = p.mean(0, dtype=np.float64).astype(np.float32)
with capi.Context() as ctx:
    ctx.build_tree(mc, cap)
    for name, q, mu in (("misaligned", p, mu_p), ("aligned", m, mu_m)):
        ctx.set_queries(q)
        ms = [ctx.nn(mu, want=())["kernel_ms"] for _ in range(4)]
        tm = ctx.timing()
        print(f"{name}: kernel ms {['%.3f' % v for v in ms]}  nodes/q {tm['nodes_visited']/tm['queries']:.2f} "
              f"pts/q {tm['points_scanned']/tm['queries']:.1f}", flush=True)
