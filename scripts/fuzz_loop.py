"""Randomised whole-loop sweep on the GPU: device-resident loop == host-driven loop bit for bit,
and both against the CPU restatement of libcpu::icp (iteration count, transform).
usage: python scripts/fuzz_loop.py [n_cases] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from icp_amd import capi, synth
from oracle import Oracle
orc = Oracle()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def run(mode, m, p, iters, thr, cap):
    os.environ["ICP_HIP_LOOP"] = mode
    errs = []
    mu_m = capi.mean_host(m)
    with capi.Context() as ctx:
        ctx.build_tree(capi.center_host(m, mu_m), cap)
        ctx.set_queries(p)
        T, it, err = ctx.run(mu_m, iters, thr, progress=lambda i, mx, e: errs.append(e))
        return T, it, err, ctx.get_points(), errs


bad = it_diff = 0
worst_T = 0.0
worst_case = ""
over = 0
t0 = time.time()
for k in range(cases):
    n = int(10 ** rng.uniform(1.3, 4.8))
    cap = int(rng.choice([4, 8, 32, 64, 256]))
    ang = float(rng.uniform(0.0, 4.0))
    iters = int(rng.choice([1, 2, 3, 5, 9, 17, 40, 200]))
    thr = float(rng.choice([0.0, 1e-5, 1e-3]))
    m = synth.cloud(n, seed=int(rng.integers(1 << 30)))
    p = synth.apply_fp32(m, synth.ground_truth(ang))
    nq = int(rng.integers(max(4, n // 2), n + 1))
    p = np.ascontiguousarray(p[rng.permutation(n)[:nq]])
    print(f"case {k}: n={n} nq={nq} cap={cap} angle={ang:.2f} iters={iters} thr={thr}", flush=True)
    if orc.tree_degenerate(orc.tree_build(cap, capi.center_host(m, capi.mean_host(m)))):
        continue
    h = run("host", m, p, iters, thr, cap)
    d = run("device", m, p, iters, thr, cap)
    same = h[1] == d[1] and (h[0].view(np.uint32) == d[0].view(np.uint32)).all() and \
        (h[3].view(np.uint32) == d[3].view(np.uint32)).all() and h[4] == d[4] and \
        np.float32(h[2]).view(np.uint32) == np.float32(d[2]).view(np.uint32)
    if not same:
        bad += 1
        print("   HOST != DEVICE", h[1], d[1], flush=True)
    cpu = orc.icp(m, p, iters, thr, cap)
    if cpu["iterations"] != d[1]:
        it_diff += 1
        print(f"   iterations: cpu {cpu['iterations']} gpu {d[1]} (errors cpu {cpu['error']:.3g} gpu {d[2]:.3g})", flush=True)
    else:
        dT = float(np.abs(cpu["T"] - d[0]).max())
        over += dT > 1e-5
        if dT > worst_T:
            worst_T, worst_case = dT, f"n={n} nq={nq} cap={cap} angle={ang:.2f} iterations run={d[1]} thr={thr}"
print(f"loop fuzz: {cases} cases, host!=device in {bad}, iteration count differs from the CPU path in "
      f"{it_diff}, max |T_gpu - T_cpu| over the rest {worst_T:.2e} ({worst_case}), above 1e-5 in {over}, "
      f"{time.time() - t0:.0f}s")
sys.exit(1 if bad else 0)
