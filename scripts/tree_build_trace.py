"""Device tree build only (for rocprofv3 --kernel-trace --stats): python3 scripts/tree_build_trace.py [n] [cap]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
from icp_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 256
m = synth.cloud(n); mc = capi.center_host(m, capi.mean_host(m))
with capi.Context() as ctx:
    ctx.build_tree(mc[:4096], cap)
    for _ in range(2):
        t0 = time.perf_counter(); ctx.build_tree(mc, cap); print("build s", time.perf_counter() - t0, flush=True)
