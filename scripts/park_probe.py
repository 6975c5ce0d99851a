"""NN pass time on the bench's two geometries (iteration 0, and the converged cloud freshly ordered)
for settings of the straggler lots: ICP_HIP_PARK_MAX x ICP_HIP_LATE_PER_BATCH (+ chunk size).
usage: python scripts/park_probe.py [n] [settings "pm:lpb:cb" ...]   (ICP_HIP_TILE_STATS=1 for the counters)"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from icp_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
settings = sys.argv[2:] or ["0:1:0", "4:1:0"]
ang = 0.25 if n > 8_000_000 else 1.0
m = synth.cloud(n)
mu_m = capi.mean_host(m)
mc = capi.center_host(m, mu_m)
p0 = synth.apply_fp32(m, synth.ground_truth(ang))
p1 = synth.apply_fp32(p0, synth.expected_transform(ang).astype(np.float32))
for s in settings:
    pm, lpb, cb, *pw = s.split(":")
    os.environ["ICP_HIP_PARK_WAIT"] = pw[0] if pw else "0"
    os.environ["ICP_HIP_PARK"] = "never" if pm == "0" else "always"
    os.environ["ICP_HIP_PARK_MAX"] = pm
    os.environ["ICP_HIP_LATE_PER_BATCH"] = lpb
    if int(cb):
        os.environ["ICP_HIP_CHUNK_BATCHES"] = cb
    else:
        os.environ.pop("ICP_HIP_CHUNK_BATCHES", None)
    with capi.Context() as ctx:
        ctx.build_tree(mc, 256)
        res = []
        for p in (p0, p1):
            ctx.set_queries(p)
            mu = capi.mean_host(p)
            ms = [ctx.nn(mu, want=())["kernel_ms"] for _ in range(4)]
            res.append(min(ms[1:]))
        print(f"park_max {pm:>2} late/batch {lpb} chunk {cb} wait {pw[0] if pw else 0}: misaligned {res[0]:.3f} ms, aligned {res[1]:.3f} ms", flush=True)
