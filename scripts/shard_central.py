"""Compact shard whose own mean equals the cloud's (a central slab): the set-up leaf order is then
computed with the right centroid.  Compare with scripts/shard_sizes.py's one-sided slabs."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from icp_amd import capi, synth
n = 64_000_000
m = synth.cloud(n); mu_m = capi.mean_host(m); mc = capi.center_host(m, mu_m)
ax = np.sort(np.abs(m[::64, 0] - mu_m[0]))
with capi.Context() as ctx:
    ctx.build_tree(mc, 256)
    for frac in (2, 8, 16):
        q = np.ascontiguousarray(m[np.abs(m[:, 0] - mu_m[0]) < ax[len(ax) // frac]])
        ctx.set_queries(q)
        ms = [ctx.nn(mu_m, want=())["kernel_ms"] for _ in range(5)]
        print(f"central slab 1/{frac}: {len(q)} queries, kernel ms min {min(ms):.3f} -> {min(ms) / len(q) * 1e6:.4f} ms per M", flush=True)
