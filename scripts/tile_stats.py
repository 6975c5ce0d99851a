import numpy as np, time, sys, os
sys.path.insert(0, os.getcwd())
from icp_amd import capi, synth
n=int(sys.argv[1]) if len(sys.argv)>1 else 8_000_000
ang=float(sys.argv[2]) if len(sys.argv)>2 else 1.0
cap=int(sys.argv[3]) if len(sys.argv)>3 else 256
m = synth.cloud(n); p = synth.apply_fp32(m, synth.ground_truth(ang))
mu_m = capi.mean_host(m); mc = capi.center_host(m, mu_m)
tree = capi.FlatTree(mc, cap)
mu_p = p.mean(0).astype(np.float32)
with capi.Context() as ctx:
    ctx.set_tree(tree); ctx.set_queries(p)
    for rep in range(2):
        out = ctx.nn(mu_p, want=())
    print('misaligned ms', out['kernel_ms'], ctx.timing(), flush=True)
    ctx.set_queries(m)
    for rep in range(2):
        out = ctx.nn(mu_m, want=())
    print('self ms', out['kernel_ms'], ctx.timing(), flush=True)
