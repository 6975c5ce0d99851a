"""One NN parity case with progress prints: python scripts/one_case.py n cap kind [seed] [nq]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from icp_amd import capi, synth
from oracle import Oracle
orc = Oracle()
n, cap, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
rng = np.random.default_rng(seed)
m = synth.cloud(n, seed=seed)
mu_m = capi.mean_host(m); mc = capi.center_host(m, mu_m)
if kind == "ulp":
    p = np.ascontiguousarray((m.view(np.int32) + rng.integers(-3, 4, size=m.shape).astype(np.int32)).view(np.float32))
elif kind == "self":
    p = m.copy()
else:
    p = synth.apply_fp32(m, synth.ground_truth(float(kind)))
nq = int(sys.argv[5]) if len(sys.argv) > 5 else n
p = np.ascontiguousarray(p[rng.permutation(n)[:nq]])
t0 = time.time(); ot = orc.tree_build(cap, mc); print("oracle build", time.time() - t0, "degenerate", orc.tree_degenerate(ot), flush=True)
mu_p = orc.mean(p)
t0 = time.time(); want = orc.tree_closest(ot, orc.subtract(p, mu_p)); print("oracle closest", time.time() - t0, flush=True)
with capi.Context() as ctx:
    t0 = time.time(); ctx.build_tree(mc, cap); print("gpu build", time.time() - t0, ctx.tree_info(), flush=True)
    t0 = time.time(); ctx.set_queries(p); print("set_queries", time.time() - t0, flush=True)
    t0 = time.time(); got = ctx.nn(mu_p); print("nn", time.time() - t0, got["kernel_ms"], ctx.timing(), flush=True)
print("equal", (got["row"] == want["row"]).all(), (got["dist"].view(np.uint32) == want["dist"].view(np.uint32)).all())
