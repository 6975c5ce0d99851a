"""The aligned NN pass three ways on one context: inside the loop (nn_trace), icp_hip_nn right after
the loop (the loop's own order), icp_hip_nn after the moved cloud was uploaded again (fresh order).
usage: python scripts/nn_vs_loop.py [n]"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from icp_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
m = synth.cloud(n); mu_m = capi.mean_host(m); mc = capi.center_host(m, mu_m)
p = synth.apply_fp32(m, synth.ground_truth(0.25))
with capi.Context() as ctx:
    ctx.build_tree(mc, 256)
    ctx.set_queries(p)
    ctx.run(mu_m, 20, 0.0)
    tr = ctx.nn_trace()
    print("loop: last NN passes", [round(float(x), 3) for x in tr[-4:]], flush=True)
    a = min(ctx.nn(mu_m, want=())["kernel_ms"] for _ in range(3))
    print(f"icp_hip_nn after the loop, centred on mu_m: {a:.3f} ms", flush=True)
    moved = ctx.get_points()
    mu = capi.mean_host(moved)
    b = min(ctx.nn(mu, want=())["kernel_ms"] for _ in range(3))
    print(f"icp_hip_nn after the loop, centred on mean(moved): {b:.3f} ms; |mu - mu_m| = {np.abs(mu - mu_m).max():.3g}", flush=True)
    ctx.set_queries(moved)
    c2 = min(ctx.nn(mu_m, want=())["kernel_ms"] for _ in range(3))
    print(f"icp_hip_nn after a fresh upload of the moved cloud: {c2:.3f} ms", flush=True)
    print(ctx.timing())
