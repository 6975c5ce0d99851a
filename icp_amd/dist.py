"""One process per GPU: shard bounds and the per-iteration all-reduce of the small sums.

The path shards over the points to move (each is an independent NN query); the only exchange is
ONE all-reduce of 16 doubles per iteration: the NN pass' nine covariance sums and three work
counters together with the error and the centroid sums of the apply pass of the iteration before
(include/icp_hip.h: icp_hip_set_allreduce / icp_hip_comm_init_rank).  The reference has
no counterpart (it is single-device); the parity target is the single-GPU result.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous shard [begin, end) of n rows for `rank` of `world` (concatenation in rank
    order restores the input order, as the moved cloud must come back in input order)."""
    per = (n + world - 1) // world
    return min(n, rank * per), min(n, (rank + 1) * per)


def torch_allreduce(group=None, device=None):
    """Callable summing a float64 numpy array in place over the process group (gloo on CPU
    tensors; nccl == RCCL when `device` is a cuda device)."""
    import torch
    import torch.distributed as dist

    def fn(values: np.ndarray) -> None:
        t = torch.from_numpy(values.copy())
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        values[:] = t.cpu().numpy()
    return fn


def init_rccl(ctx, world: int, rank: int) -> None:
    """Create the context's own RCCL communicator: rank 0 makes the ncclUniqueId, the process
    group distributes it."""
    import torch.distributed as dist
    from . import capi
    uid = [capi.Context.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    ctx.comm_init_rank(world, rank, uid[0])
