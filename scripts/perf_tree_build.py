import numpy as np, time, sys, os
sys.path.insert(0, os.getcwd())
from icp_amd import capi, synth
for n, cap in [(1_000_000, 256), (8_000_000, 256), (64_000_000, 256), (64_000_000, 32)]:
    m = synth.cloud(n)
    mc = capi.center_host(m, capi.mean_host(m)); del m
    t0 = time.time(); host = capi.FlatTree(mc, cap); th = time.time() - t0
    with capi.Context() as ctx:
        ctx.build_tree(mc, cap)       # warm (first-call allocations, code load)
        t0 = time.time(); ctx.build_tree(mc, cap); td = time.time() - t0
        t0 = time.time(); ctx.set_tree(host); tu = time.time() - t0
    print(f"n={n} cap={cap}: host build {th:.3f}s + upload {tu:.3f}s ; device build (incl. H2D of the cloud) {td:.3f}s ; height {host.height}", flush=True)
    host.close()
