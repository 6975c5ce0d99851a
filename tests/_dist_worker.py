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
        # The sharded decomposition of one iteration with the oracle standing in for the kernels:
        # per-shard fp64 partial sums, summed by the SAME callback object the C ABI calls.
        from oracle import Oracle
        orc = Oracle()
        mu_m = orc.mean(m)
        mc = orc.subtract(m, mu_m)
        tree = orc.tree_build(cap, mc)
        sums = np.zeros(4)
        sums[1:] = p[b:e].astype(np.float64).sum(0)
        cb = capi.ALLREDUCE_FN(lambda ptr, cnt, _u: (allreduce(np.ctypeslib.as_array(ptr, shape=(cnt,))), 0)[1])
        assert cb(sums.ctypes.data_as(C.POINTER(C.c_double)), 4, None) == 0
        mu_p = (sums[1:] / n).astype(np.float32)
        q = orc.subtract(p[b:e], mu_p)
        nn = orc.tree_closest(tree, q, counters=True)
        cov = np.zeros(11)
        cov[:9] = (q[:, :, None] * nn["xyz"][:, None, :]).astype(np.float32).astype(np.float64).sum(0).reshape(9)
        cov[9], cov[10] = nn["n_int"].sum(), nn["n_pts"].sum()
        assert cb(cov.ctypes.data_as(C.POINTER(C.c_double)), 11, None) == 0
        T = capi.alignment_host(cov[:9].astype(np.float32), mu_p, mu_m)
        moved = orc.apply_alignment(p[b:e], T)
        d = (nn["xyz"] + mu_m - moved).astype(np.float32)
        err = np.zeros(4)
        err[0] = ((d * d).astype(np.float32)).astype(np.float64).sum()
        assert cb(err.ctypes.data_as(C.POINTER(C.c_double)), 4, None) == 0
        result.update(mu_p=[float(v) for v in mu_p], cov=[float(v) for v in cov],
                      T=[float(v) for v in T.reshape(-1)], err=float(err[0]),
                      rows=nn["row"].tolist())
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
