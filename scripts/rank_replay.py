"""One rank of an N-rank registration, replayed on one GPU.  A first run of the whole cloud through
the all-reduce callback records the all-reduced sums of every slot; the shard is then registered
alone, with a callback that puts the recorded sums in place of its own: it sees exactly what it
would see as one of N ranks (the same centroids and transforms), and its NN passes are a rank's.
(Host-driven loop, as always with a callback.)  "tree" = the shard a rank holds after
icp_hip_set_queries_exchange: the points whose near-only descent of p - mean(p) goes one way through
the top log2(share) levels of the tree (rank = `which`).
usage: python scripts/rank_replay.py [n] [share] [rows|spatial|tree] [iters] [which]"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from icp_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
share = int(sys.argv[2]) if len(sys.argv) > 2 else 8
how = sys.argv[3] if len(sys.argv) > 3 else "rows"
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 8
which = int(sys.argv[5]) if len(sys.argv) > 5 else 0
m = synth.cloud(n); mu_m = capi.mean_host(m); mc = capi.center_host(m, mu_m)
p = synth.apply_fp32(m, synth.ground_truth(0.25))
if how == "spatial":
    idx = np.arange(n)
    axis, parts = 0, share
    while parts > 1:  # keep the lower half of a median split, axis after axis
        half = len(idx) // 2
        idx = idx[np.argpartition(p[idx, axis], half)[:half]]
        axis, parts = (axis + 1) % 3, parts // 2
    sel = np.sort(idx)
else:
    sel = np.arange(n // share)
rec = []
with capi.Context() as ctx:
    ctx.build_tree(mc, 256)
    if how == "tree":
        t = ctx.get_tree()
        L = int(np.log2(share))
        mu = (p.astype(np.float64).sum(0) / n).astype(np.float32)
        q = p - mu
        node = np.full(n, int(t.root), np.int64)
        path = np.zeros(n, np.int64)
        thr = np.nextafter(t.node_radius.astype(np.float32) ** 2, np.float32(np.inf))  # ~ the kernel's d2 < thr
        for _ in range(L):
            c = np.stack([t.node_cx[node], t.node_cy[node], t.node_cz[node]], 1)
            d = q - c
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            inside = d2 < thr[node]
            path = path * 2 + (~inside)
            node = np.where(inside, t.node_inside[node], t.node_outside[node]).astype(np.int64)
        sel = np.nonzero(path * share // (1 << L) == which)[0]
        del q, node, path, c, d, d2, inside
        t.close()
    ctx.set_allreduce(lambda a: rec.append(a.copy()))
    ctx.set_queries(p)
    T0, it0, _ = ctx.run(mu_m, iters, 0.0)
    full = np.array(ctx.nn_trace())
    shard = np.ascontiguousarray(p[sel])
    for recentre in ("0", "1"):
        os.environ["ICP_HIP_RECENTRE"] = recentre
        k = [0]
        def replay(a):
            a[:] = rec[k[0]]
            k[0] += 1
        ctx.set_allreduce(replay)
        ctx.set_queries(shard, n_total=n)
        T1, it1, _ = ctx.run(mu_m, iters, 0.0)
        part = np.array(ctx.nn_trace())
        assert it1 == it0 and (T1.view(np.uint32) == T0.view(np.uint32)).all()
        print(f"1/{share} ({how}, {len(sel)} points), ICP_HIP_RECENTRE={recentre}: NN passes "
              + " ".join(f"{x:.3f}" for x in part) + " ms; x its share: "
              + " ".join(f"{x * n / len(sel) / y:.2f}" for x, y in zip(part, full)), flush=True)
print("whole cloud: NN passes " + " ".join(f"{x:.3f}" for x in full) + " ms")
