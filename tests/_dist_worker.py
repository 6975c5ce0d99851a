"""Worker for the world_size-2 tests (launched by tests/test_dist_gloo.py and
tests/test_gpu_sharded.py with RANK/WORLD_SIZE/MASTER_* in the environment)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist  # noqa: E402

from icp_amd import capi, synth  # noqa: E402
from icp_amd.dist import shard_bounds, torch_allreduce  # noqa: E402


def main():
    mode, out_path = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, cap = 30011, 64  # odd size: ragged last shard
    m = synth.cloud(n, seed=21)
    p = synth.apply_fp32(m, synth.ground_truth(2.0))
    b, e = shard_bounds(n, world, rank)
    allreduce = torch_allreduce()
    result = {"rank": rank, "shard": [b, e]}

    if mode == "cpu":
        # The sharded decomposition of the loop with the oracle standing in for the kernels, in
        # the order the library uses: ONE all-reduce of 16 doubles per iteration -- [0..8] the
        # covariance sums and [9..11] the work counters of this iteration's NN pass, [12..15] the
        # error and the moved-cloud sums of the apply pass before it -- summed by the SAME
        # callback type the C ABI calls (icp_hip_set_allreduce).
        from oracle import Oracle
        orc = Oracle()
        mu_m = orc.mean(m)
        mc = orc.subtract(m, mu_m)
        tree = orc.tree_build(cap, mc)
        calls = []

        def reduce(v):
            calls.append(len(v))
            allreduce(v)
        cb = capi.ALLREDUCE_FN(lambda ptr, cnt, _u: (reduce(np.ctypeslib.as_array(ptr, shape=(cnt,))), 0)[1])
        mine = p[b:e].copy()
        first = np.zeros(4)
        first[1:] = mine.astype(np.float64).sum(0)
        assert cb(first.ctypes.data_as(C.POINTER(C.c_double)), 4, None) == 0
        mu_p = (first[1:] / n).astype(np.float32)
        mu_q = mu_p.copy()
        carry = np.zeros(4)
        T_tot = np.eye(4, dtype=np.float32)
        errors, rows, covs = [], [], []
        for it in range(3):
            q = orc.subtract(mine, mu_q)
            nn = orc.tree_closest(tree, q, counters=True)
            sums = np.zeros(16)
            sums[:9] = (q[:, :, None] * nn["xyz"][:, None, :]).astype(np.float32).astype(np.float64).sum(0).reshape(9)
            sums[9], sums[10] = nn["n_int"].sum(), nn["n_pts"].sum()
            sums[12:] = carry
            assert cb(sums.ctypes.data_as(C.POINTER(C.c_double)), 16, None) == 0
            if it > 0:  # the apply pass of the iteration before: its error and the next centroid
                errors.append(float(np.float32(sums[12])))
                mu_p = (sums[13:] / n).astype(np.float32)
            Tn = capi.alignment_host(sums[:9].astype(np.float32), mu_p, mu_m)
            T_tot = orc.matmul4(Tn, T_tot)
            mu_q = mu_m.copy()
            mine = orc.apply_alignment(mine, Tn)
            d = (nn["xyz"] + mu_m - mine).astype(np.float32)
            carry = np.zeros(4)
            carry[0] = ((d * d).astype(np.float32)).astype(np.float64).sum()
            carry[1:] = mine.astype(np.float64).sum(0)
            rows.append(nn["row"].tolist())
            covs.append([float(v) for v in sums[:12]])
        result.update(calls=calls, covs=covs, errors=errors, rows=rows,
                      T=[float(v) for v in T_tot.reshape(-1)])
    elif mode == "gpu_empty":
        # rank 0 holds the whole cloud, rank 1 an empty shard (more ranks than points)
        mu_m = capi.mean_host(m)
        tree = capi.FlatTree(capi.center_host(m, mu_m), cap)
        mine = p if rank == 0 else p[:0]
        with capi.Context(0) as ctx:
            ctx.set_tree(tree)
            ctx.set_queries(mine, n_total=n)
            ctx.set_allreduce(allreduce)
            T, iters, err = ctx.run(mu_m, 50, 1e-5)
            new_p = ctx.get_points()
        result.update(T=[float(v) for v in T.reshape(-1)], iters=int(iters), err=float(err),
                      new_p=new_p.tolist())
    elif mode == "gpu_spatial":
        # spatially compact shards (icp_amd.dist.spatial_shard) of a cloud large enough for the
        # library to order each shard again around the whole cloud's mean (recentre_order)
        from icp_amd.dist import spatial_shard
        n2 = 400_000
        m2 = synth.cloud(n2, seed=23)
        p2 = synth.apply_fp32(m2, synth.ground_truth(1.0))
        rows = spatial_shard(p2, world, rank)
        mu_m = capi.mean_host(m2)
        with capi.Context(0) as ctx:
            ctx.build_tree(capi.center_host(m2, mu_m), 256)
            ctx.set_queries(np.ascontiguousarray(p2[rows]), n_total=n2)
            ctx.set_allreduce(allreduce)
            T, iters, err = ctx.run(mu_m, 30, 1e-5)
            new_p = ctx.get_points()
        np.savez(f"{out_path}.{rank}.npz", rows=rows, new_p=new_p)
        result.update(T=[float(v) for v in T.reshape(-1)], iters=int(iters), err=float(err))
    elif mode == "gpu_exchange":
        # SURVEY 8e's contract: every rank hands over a contiguous slice of the cloud's rows (in
        # whatever order the cloud comes: here a random one) and the library re-partitions
        # behind the ABI (icp_hip_set_queries_exchange, points travelling over gloo through the
        # all-to-all callback); get_points brings every rank ITS rows back
        from icp_amd.dist import torch_alltoallv
        n2 = 400_000
        m2 = synth.cloud(n2, seed=23)
        p2 = synth.apply_fp32(m2, synth.ground_truth(1.0))
        b2, e2 = shard_bounds(n2, world, rank)
        mu_m = capi.mean_host(m2)
        with capi.Context(0) as ctx:
            ctx.build_tree(capi.center_host(m2, mu_m), 256)
            ctx.set_alltoallv(world, rank, torch_alltoallv())
            ctx.set_allreduce(allreduce)
            ctx.set_queries_exchange(p2[b2:e2], n_total=n2)
            T, iters, err = ctx.run(mu_m, 30, 1e-5)
            new_p = ctx.get_points()
            local = ctx.shard_size()[0]  # the size of this rank's region
        np.savez(f"{out_path}.{rank}.npz", rows=np.arange(b2, e2), new_p=new_p)
        result.update(T=[float(v) for v in T.reshape(-1)], iters=int(iters), err=float(err), queries=local)
    else:
        # real GPU, both ranks on device 0, sums exchanged over gloo through the C-ABI callback
        mu_m = capi.mean_host(m)
        tree = capi.FlatTree(capi.center_host(m, mu_m), cap)
        with capi.Context(0) as ctx:
            ctx.set_tree(tree)
            ctx.set_queries(p[b:e], n_total=n)
            ctx.set_allreduce(allreduce)
            T, iters, err = ctx.run(mu_m, 50, 1e-5)
            new_p = ctx.get_points()
        result.update(T=[float(v) for v in T.reshape(-1)], iters=int(iters), err=float(err),
                      new_p=new_p.tolist())
    with open(f"{out_path}.{rank}", "w") as f:
        json.dump(result, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
