"""Experiment: what would a second ordering pass after convergence buy?  Registers the bench's
64M clouds for a few iterations, times the NN pass on the converged cloud in the set-up order,
then uploads the moved cloud again (set-up sorts at the aligned pose) and times the same pass."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from icp_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
ang = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
m = synth.cloud(n); mu_m = capi.mean_host(m); mc = capi.center_host(m, mu_m)
p = synth.apply_fp32(m, synth.ground_truth(ang))
with capi.Context() as ctx:
    ctx.build_tree(mc, 256)
    ctx.set_queries(p)
    ctx.run(mu_m, 8, 0.0)
    print("loop NN ms:", " ".join(f"{v:.2f}" for v in ctx.nn_trace()), flush=True)
    ms = [ctx.nn(mu_m, want=())["kernel_ms"] for _ in range(3)]
    print("converged, set-up order :", ["%.3f" % v for v in ms], flush=True)
    moved = ctx.get_points()
    ctx.set_queries(moved)
    ms = [ctx.nn(mu_m, want=())["kernel_ms"] for _ in range(3)]
    print("converged, re-sorted    :", ["%.3f" % v for v in ms], flush=True)
    ctx.run(mu_m, 3, 0.0)
    print("loop NN ms after re-sort:", " ".join(f"{v:.2f}" for v in ctx.nn_trace()), flush=True)
