"""NN pass of an 8M-query shard against the 64M-point tree: a random eighth of the cloud (what a
contiguous slice of the input is) vs a compact eighth (a slab), both exactly aligned."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from icp_amd import capi, synth
n = 64_000_000
m = synth.cloud(n); mu_m = capi.mean_host(m); mc = capi.center_host(m, mu_m)
with capi.Context() as ctx:
    ctx.build_tree(mc, 256)
    xs = np.sort(m[::64, 0]); thr = xs[len(xs) // 8]
    for name, q in (("random eighth", m[: n // 8]), ("slab eighth", m[m[:, 0] < thr]), ("all", m)):
        ctx.set_queries(np.ascontiguousarray(q))
        ms = [ctx.nn(mu_m, want=())["kernel_ms"] for _ in range(4)]
        print(f"{name}: {len(q)} queries, kernel ms {['%.3f' % v for v in ms]}  -> {min(ms) / len(q) * 1e6:.3f} ns/query", flush=True)
