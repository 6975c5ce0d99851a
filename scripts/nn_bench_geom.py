"""NN-kernel time on the bench's geometries: the shard is ordered for the INITIAL pose
(icp_hip_set_queries), then registered for a few iterations; the NN pass is timed on the initial
pose and on the converged one.  ICP_HIP_TILE_STATS=1 prints the kernel's event counts.
usage: python scripts/nn_bench_geom.py [n] [capacity] [angle] [iterations]"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from icp_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ang = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
its = int(sys.argv[4]) if len(sys.argv) > 4 else 8
m = synth.cloud(n); mu_m = capi.mean_host(m); mc = capi.center_host(m, mu_m)
p = synth.apply_fp32(m, synth.ground_truth(ang))
mu_p = p.mean(0, dtype=np.float64).astype(np.float32)

with capi.Context() as ctx:
    ctx.build_tree(mc, cap)
    ctx.set_queries(p)
    ms = [ctx.nn(mu_p, want=())["kernel_ms"] for _ in range(3)]
    tm = ctx.timing()
    print(f"initial pose: kernel ms {['%.3f' % v for v in ms]}  nodes/q {tm['nodes_visited']/tm['queries']:.2f} "
          f"pts/q {tm['points_scanned']/tm['queries']:.1f}", flush=True)
    ctx.run(mu_m, its, 0.0)
    tr = ctx.nn_trace()
    print("loop NN ms per iteration:", " ".join(f"{v:.2f}" for v in tr), flush=True)
    t0 = ctx.timing()
    ms = [ctx.nn(mu_m, want=())["kernel_ms"] for _ in range(3)]
    tm = ctx.timing()
    dq = tm["queries"] - t0["queries"]
    print(f"converged:   kernel ms {['%.3f' % v for v in ms]}  nodes/q {(tm['nodes_visited']-t0['nodes_visited'])/dq:.2f} "
          f"pts/q {(tm['points_scanned']-t0['points_scanned'])/dq:.1f}", flush=True)
