"""Host-side mirror of the reference's registration interface, over the C ABI.

``icp`` has the argument meaning, defaults and error behaviour of ``libgpu::icp`` /
``libcpu::icp`` (reference libgpu/libgpu/icp.hh:19-21, libcpu/libcpu/icp.cc:97-128): errors
surface as exceptions (the reference throws std::runtime_error), progress lines go to stderr
exactly as the reference prints them (icp.cc:112-113,124) unless ``quiet``.
"""
from __future__ import annotations

import sys

import numpy as np

from . import capi


def _progress_printer(max_iters):
    def cb(i, m, err):
        # icp.cc:112-113 prints "Starting iter" before the work and "Error:" after it (:124);
        # both lines are emitted here, in that order, once the iteration's error is known.
        sys.stderr.write(f"Starting iter {i}/{m}\n")
        sys.stderr.write(f"Error: {np.float32(err):g}\n")
    return cb


def icp(m, p, iterations: int = 200, threshold: float = 1e-5, vp_threshold: int = 256, *,
        device_id: int = -1, quiet: bool = True, shard=None, allreduce=None, comm=None,
        return_info: bool = False, tree_on: str = "device"):
    """Register cloud ``p`` onto reference cloud ``m`` on one MI355X.

    Returns ``(T, new_p)``: the 4x4 row-major transform mapping the original ``p`` onto ``m``
    and the moved ``p`` in input order.  ``vp_threshold`` is the VP-tree leaf capacity
    (CLI default 256 on gpu, options.cc:175-178).

    Sharded runs (one process per GPU): pass ``shard=(begin, end)`` to own rows [begin, end) of
    ``p`` and either ``allreduce`` (callable summing a float64 array in place over ranks) or
    ``comm=(n_ranks, rank, unique_id)`` for RCCL.  ``new_p`` is then the moved shard.

    ``tree_on``: "device" builds the VP-tree in HBM (icp_hip_build_tree), "host" builds it with
    icp_hip_tree_build and uploads it; the two trees are identical.
    """
    m = capi._pts(m)
    p = capi._pts(p)
    if len(m) == 0 or len(p) == 0:
        raise ValueError("icp: empty point cloud")
    if tree_on not in ("device", "host"):
        raise ValueError("tree_on must be 'device' or 'host'")
    mu_m = capi.mean_host(m)                       # icp.cc:106
    mc = capi.center_host(m, mu_m)                 # icp.cc:108
    with capi.Context(device_id) as ctx:
        if tree_on == "device":
            ctx.build_tree(mc, vp_threshold)
        else:
            tree = capi.FlatTree(mc, vp_threshold)
            try:
                ctx.set_tree(tree)
            finally:
                tree.close()
        del mc
        b, e = (0, len(p)) if shard is None else shard
        ctx.set_queries(p[b:e], n_total=len(p))
        if comm is not None:
            ctx.comm_init_rank(*comm)
        elif allreduce is not None:
            ctx.set_allreduce(allreduce)
        T, iters, err = ctx.run(mu_m, iterations, threshold,
                                None if quiet else _progress_printer(iterations))
        new_p = ctx.get_points()
        ti = ctx.tree_info()
        info = dict(iterations=iters, error=err, timing=ctx.timing(), height=ti["height"],
                    n_nodes=ti["n_nodes"], n_leaves=ti["n_leaves"])
    if return_info:
        return T, new_p, info
    return T, new_p


def closest(m_centered, queries, vp_threshold: int = 256, *, device_id: int = -1):
    """VPTree(vp_threshold, m_centered).closest(queries) (vp-tree.cc:177-187) on the GPU.

    Returns dict(xyz, row, dist, kernel_ms).
    """
    with capi.Context(device_id) as ctx:
        ctx.build_tree(m_centered, vp_threshold)
        ctx.set_queries(queries)
        return ctx.nn(np.zeros(3, np.float32))
