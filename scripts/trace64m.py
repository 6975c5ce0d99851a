"""Per-iteration NN-kernel time and error of the bench workload (64M, 0.25 deg)."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from icp_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
ang = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
m = synth.cloud(n); mu_m = capi.mean_host(m); mc = capi.center_host(m, mu_m)
p = synth.apply_fp32(m, synth.ground_truth(ang))
with capi.Context() as ctx:
    ctx.build_tree(mc, 256)
    ctx.set_queries(p)
    errs = []
    T, it, err = ctx.run(mu_m, iters, 0.0, progress=lambda i, mx, e: errs.append(e))
    tm = ctx.timing()
    print("nn ms:", " ".join("%.2f" % v for v in ctx.nn_trace()))
    print("error:", " ".join("%.3g" % e for e in errs))
    print("nodes/q %.2f pts/q %.1f" % (tm["nodes_visited"] / tm["queries"], tm["points_scanned"] / tm["queries"]))
