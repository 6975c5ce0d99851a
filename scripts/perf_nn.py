import numpy as np, time, sys, os
sys.path.insert(0, os.getcwd())
from icp_amd import capi, synth
from oracle import Oracle
o = Oracle()
for n, cap, ang in [(1_000_000, 256, 1.0), (1_000_000, 256, 5.0), (1_000_000, 32, 1.0), (8_000_000, 256, 1.0)]:
    m = synth.cloud(n); p = synth.apply_fp32(m, synth.ground_truth(ang))
    mu_m = capi.mean_host(m); mc = capi.center_host(m, mu_m)
    t0=time.time(); tree = capi.FlatTree(mc, cap); tb=time.time()-t0
    mu_p = o.mean(p)
    with capi.Context() as ctx:
        ctx.set_tree(tree); ctx.set_queries(p)
        for rep in range(3):
            out = ctx.nn(mu_p, want=())
        ms = out['kernel_ms']
        # converged-geometry: query the cloud itself
        ctx.set_queries(m)
        for rep in range(3):
            out2 = ctx.nn(mu_m, want=())
        print(f'n={n} cap={cap} ang={ang} build {tb:.2f}s  NN misaligned {ms:.3f} ms = {n/ms/1e3:.1f} Mq/s ; self-query {out2["kernel_ms"]:.3f} ms = {n/out2["kernel_ms"]/1e3:.1f} Mq/s', flush=True)
    tree.close()
