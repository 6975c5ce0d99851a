"""Randomised NN parity sweep on the GPU against the oracle (one-off confidence run; the committed
tests hold the fixed cases).  usage: python scripts/fuzz_parity.py [n_cases] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from icp_amd import capi, synth
from oracle import Oracle
orc = Oracle()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # cases before this one only advance the RNG
bad = 0
skipped = 0
queries = 0
t0 = time.time()
for k in range(cases):
    n = int(10 ** rng.uniform(0.5, 6.2))
    cap = int(rng.choice([3, 4, 7, 8, 16, 31, 32, 64, 100, 128, 255, 256, 257, 400, 600]))
    kind = rng.choice(["rigid", "tiny", "ulp", "self", "scaled"])
    if kind in ("rigid", "scaled"):
        n = min(n, 100_000)  # grossly misaligned clouds defeat the pruning: keep those small
    m = synth.cloud(n, seed=int(rng.integers(1 << 30)))
    if rng.random() < 0.3:
        m = (m * np.float32(10 ** rng.uniform(-3, 2))).astype(np.float32)   # other coordinate scales
    mu_m = capi.mean_host(m)
    mc = capi.center_host(m, mu_m)
    if kind == "rigid":
        p = synth.apply_fp32(m, synth.ground_truth(float(rng.uniform(0.05, 20))))
    elif kind == "tiny":
        p = synth.apply_fp32(m, synth.ground_truth(float(rng.uniform(0, 0.01))))
    elif kind == "ulp":
        p = np.ascontiguousarray((m.view(np.int32) + rng.integers(-3, 4, size=m.shape).astype(np.int32)).view(np.float32))
        p = np.where(np.isfinite(p), p, m)   # 0.0 minus one ulp in the integer view is a NaN
    elif kind == "self":
        p = m.copy()
    else:
        p = (m * np.float32(rng.uniform(0.5, 1.5))).astype(np.float32)
    nq = int(rng.integers(1, len(p) + 1)) if rng.random() < 0.3 else len(p)
    p = np.ascontiguousarray(p[rng.permutation(len(p))[:nq]])
    print(f"case {k}: n={n} cap={cap} kind={kind} nq={nq}", flush=True)
    ot = orc.tree_build(cap, mc)
    if orc.tree_degenerate(ot):
        skipped += 1
        continue
    mu_p = orc.mean(p)
    if k >= first:
        want = orc.tree_closest(ot, orc.subtract(p, mu_p))
    env = {}
    if rng.random() < 0.4:
        env["ICP_HIP_BATCH_QUERIES"] = str(rng.choice([64, 32, 16, 8, 4, 2]))
    if rng.random() < 0.2:
        env["ICP_HIP_TILE_CUT"] = str(rng.choice([0, 1, 16, 64]))
    if k < first:
        continue
    print(f"   max|m|={float(np.abs(m).max()):.4g} env={env}", flush=True)
    for a, b in env.items():
        os.environ[a] = b
    try:
        with capi.Context() as ctx:
            ctx.build_tree(mc, cap)
            print("   tree built", ctx.tree_info()["height"], flush=True)
            ctx.set_queries(p)
            print("   queries set", flush=True)
            got = ctx.nn(mu_p)
            print("   nn done %.3f ms" % got["kernel_ms"], flush=True)
    finally:
        for a in env:
            os.environ.pop(a, None)
    queries += nq
    ok = (got["row"] == want["row"]).all() and (got["dist"].view(np.uint32) == want["dist"].view(np.uint32)).all() \
        and (got["xyz"].view(np.uint32) == want["xyz"].view(np.uint32)).all()
    if not ok:
        bad += 1
        print(f"MISMATCH case {k}: n={n} cap={cap} kind={kind} nq={nq} env={env} "
              f"rows differing {(got['row'] != want['row']).sum()}", flush=True)
    if k % 10 == 9:
        print(f"{k + 1} cases, {bad} mismatches, {time.time() - t0:.0f}s", flush=True)
print(f"fuzz parity: {cases - skipped} cases run ({skipped} degenerate skipped), {queries} queries:", "CLEAN" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
