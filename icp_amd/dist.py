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


def spatial_shard(xyz, world: int, rank: int):
    """Rows of `xyz` (n x 3; ascending indices) that `rank` of `world` registers when the cloud is
    cut into `world` spatially compact parts of equal size (+-1): planes perpendicular to x, y, z
    in turn, each at the quantile that leaves both sides their share of the ranks.  Every rank
    calls it on the same array and gets its part of one partition (disjoint, covering the cloud).

    Why: a rank searches the whole tree for its shard.  Queries that are neighbours in space
    share buckets, so a compact shard costs a rank about its share of the one-GPU pass (1.0-1.15x
    for an eighth of 64M points), while every bucket of the tree is staged for a random subset
    (1.5-1.75x: DESIGN section 6).  The order of the rows a rank is given is otherwise free: the
    library orders its shard itself.  The moved cloud comes back per shard, in the shard's row
    order: `out[rows] = ctx.get_points()` puts it back in input order."""
    import numpy as np
    idx = np.arange(len(xyz))
    lo, hi, axis = 0, int(world), 0
    if not 0 <= rank < hi:
        raise ValueError("rank outside [0, world)")
    while hi - lo > 1 and len(idx):
        mid = lo + (hi - lo) // 2
        k = len(idx) * (mid - lo) // (hi - lo)
        part = np.argpartition(xyz[idx, axis], min(k, len(idx) - 1))
        if rank < mid:
            idx, hi = idx[part[:k]], mid
        else:
            idx, lo = idx[part[k:]], mid
        axis = (axis + 1) % 3
    return np.sort(idx)


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


def torch_alltoallv(group=None, device=None):
    """Callable for Context.set_alltoallv: variable-size blocks of bytes between the ranks of the
    process group (gloo on CPU tensors; nccl == RCCL when `device` is a cuda device)."""
    import torch
    import torch.distributed as dist

    def fn(send: np.ndarray, send_bytes, recv_bytes) -> np.ndarray:
        world = len(send_bytes)
        ins, at = [], 0
        for b in send_bytes:
            t = torch.from_numpy(np.array(send[at:at + b], np.uint8))
            ins.append(t.to(device) if device is not None else t)
            at += b
        outs = [torch.empty(b, dtype=torch.uint8, device=device) for b in recv_bytes]
        # (gloo has no all-to-all of uneven blocks: world rounds of broadcasts would do too, but
        # point-to-point sends in a fixed order are enough here and never deadlock: everybody
        # posts its receives first)
        rank = dist.get_rank(group)
        reqs = [dist.irecv(outs[r], src=r, group=group) for r in range(world) if r != rank and recv_bytes[r]]
        reqs += [dist.isend(ins[r], dst=r, group=group) for r in range(world) if r != rank and send_bytes[r]]
        outs[rank].copy_(ins[rank])
        for q in reqs:
            q.wait()
        return np.concatenate([o.cpu().numpy() for o in outs]) if world else np.empty(0, np.uint8)
    return fn


def init_rccl(ctx, world: int, rank: int) -> None:
    """Create the context's own RCCL communicator: rank 0 makes the ncclUniqueId, the process
    group distributes it."""
    import torch.distributed as dist
    from . import capi
    uid = [capi.Context.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    ctx.comm_init_rank(world, rank, uid[0])
