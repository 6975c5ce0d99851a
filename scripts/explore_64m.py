import numpy as np, time, sys, os
sys.path.insert(0, os.getcwd())
from icp_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 8_000_000
m = synth.cloud(n); mu_m = capi.mean_host(m); mc = capi.center_host(m, mu_m)
t0=time.time(); tree = capi.FlatTree(mc, 256); print('build', time.time()-t0, flush=True)
ctx = capi.Context(); ctx.set_tree(tree)
for ang in [1.0, 0.5, 0.25, 0.1]:
    p = synth.apply_fp32(m[:nq], synth.ground_truth(ang))
    ctx.set_queries(p, n_total=nq)
    errs = []
    T, it, err = ctx.run(mu_m, 80, 1e-5, progress=lambda i,mx,e: errs.append(e))
    tr = ctx.nn_trace()
    want = synth.expected_transform(ang)
    print(f'angle {ang}: iters {it} err {err:.3e} |T-want| {np.abs(T-want).max():.2e}  errs', ['%.3g'%e for e in errs[:6]], '...', ['%.3g'%e for e in errs[-3:]], 'nn ms first/last', tr[0], tr[-1], flush=True)
