"""What one rank of an 8-GPU run costs, measured on one GPU: the NN pass of a contiguous eighth of
the cloud to move (the synthetic cloud's rows are in random order, so that is a random eighth)
against the whole 64M-point tree, at the initial pose and at the converged pose (taken from a
registration of the whole cloud, as the ranks' all-reduced sums would give it), next to the pass
over the whole cloud.  icp_hip_nn is timed, on freshly uploaded shards: their order is the one
computed around the SHARD's mean, which misfits a spatially compact shard (third argument
"spatial": median splits along x, y, z) -- scripts/rank_replay.py replays a rank inside the loop,
where such a shard is ordered again.  usage: python scripts/rank_share.py [n] [share] [rows|spatial]"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from icp_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
share = int(sys.argv[2]) if len(sys.argv) > 2 else 8
how = sys.argv[3] if len(sys.argv) > 3 else "rows"
m = synth.cloud(n); mu_m = capi.mean_host(m); mc = capi.center_host(m, mu_m)
p = synth.apply_fp32(m, synth.ground_truth(0.25))
mu_p = p.mean(0, dtype=np.float64).astype(np.float32)
def t(ctx, mu):
    return min(ctx.nn(mu, want=())["kernel_ms"] for _ in range(3))
with capi.Context() as ctx:
    ctx.build_tree(mc, 256)
    ctx.set_queries(p)
    full0 = t(ctx, mu_p)
    ctx.run(mu_m, 10, 0.0)
    moved = ctx.get_points()
    ctx.set_queries(moved)
    full1 = t(ctx, mu_m)
    k = n // share
    if how == "spatial":
        idx = np.arange(n)
        axis, parts = 0, share
        while parts > 1:  # keep the lower half of a median split, axis after axis
            half = len(idx) // 2
            idx = idx[np.argpartition(p[idx, axis], half)[:half]]
            axis, parts = (axis + 1) % 3, parts // 2
        sel = np.sort(idx)
        k = len(sel)
    else:
        sel = slice(0, k)
    ctx.set_queries(np.ascontiguousarray(p[sel]), n_total=n)
    part0 = t(ctx, mu_p)
    ctx.set_queries(np.ascontiguousarray(moved[sel]), n_total=n)
    part1 = t(ctx, mu_m)
print(f"whole cloud: initial pose {full0:.3f} ms, converged (fresh order) {full1:.3f} ms")
print(f"1/{share} of it ({how}, {k} points): initial pose {part0:.3f} ms ({part0 * share / full0:.2f}x its share), "
      f"converged {part1:.3f} ms ({part1 * share / full1:.2f}x its share)")
