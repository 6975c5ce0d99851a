"""Aligned NN pass of subsets of the 64M cloud against the 64M tree: time vs shard size, for random
subsets (a contiguous slice of the input) and for compact slabs (full query density per bucket)."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from icp_amd import capi, synth
n = 64_000_000
m = synth.cloud(n); mu_m = capi.mean_host(m); mc = capi.center_host(m, mu_m)
xs = np.sort(m[::64, 0])
with capi.Context() as ctx:
    ctx.build_tree(mc, 256)
    for frac in (1, 2, 4, 8, 16, 32, 64):
        for kind in ("random", "slab"):
            if kind == "random":
                q = m[: n // frac]
            else:
                q = m if frac == 1 else m[m[:, 0] < xs[len(xs) // frac]]
            ctx.set_queries(np.ascontiguousarray(q))
            ms = [ctx.nn(mu_m, want=())["kernel_ms"] for _ in range(5)]
            print(f"1/{frac} {kind}: {len(q)} queries, kernel ms min {min(ms):.3f} -> {min(ms) / len(q) * 1e6:.4f} ms per M", flush=True)
