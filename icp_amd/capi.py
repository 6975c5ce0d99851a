"""ctypes binding of include/icp_hip.h (libicp_hip.so).

This is plumbing only: every call goes to the C ABI, which fails loudly (non-zero status ->
``IcpHipError``) when no HIP device or extension is available.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libicp_hip.so")

STATUS = {0: "OK", 1: "ERR_ARG", 2: "ERR_DEGENERATE", 3: "ERR_HIP", 4: "ERR_COMM", 5: "ERR_DEPTH",
          6: "ERR_NOMEM"}

#: every symbol include/icp_hip.h declares
EXPORTS = [
    "icp_hip_tree_build", "icp_hip_tree_free", "icp_hip_tree_validate", "icp_hip_mean_host",
    "icp_hip_center_host", "icp_hip_alignment_host", "icp_hip_create", "icp_hip_destroy", "icp_hip_last_error",
    "icp_hip_last_global_error", "icp_hip_set_tree", "icp_hip_build_tree", "icp_hip_get_tree",
    "icp_hip_get_tree_info", "icp_hip_read_csv", "icp_hip_read_f32", "icp_hip_free_points",
    "icp_hip_set_queries", "icp_hip_nn", "icp_hip_nn_noalloc",
    "icp_hip_set_allreduce", "icp_hip_comm_unique_id", "icp_hip_comm_init_rank", "icp_hip_run",
    "icp_hip_get_points", "icp_hip_get_timing", "icp_hip_get_nn_trace", "icp_hip_icp",
    "icp_hip_set_phase_timers", "icp_hip_set_queries_exchange", "icp_hip_set_alltoallv",
    "icp_hip_get_shard_size",
]


class IcpHipError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"[{STATUS.get(status, status)}] {message}")
        self.status = status


class FlatTreeStruct(C.Structure):
    _fields_ = [
        ("n_points", C.c_uint64), ("n_slots", C.c_uint64), ("n_nodes", C.c_uint32),
        ("n_leaves", C.c_uint32), ("height", C.c_uint32), ("capacity", C.c_uint32),
        ("max_leaf_count", C.c_uint32), ("root", C.c_int32),
        ("node_cx", C.POINTER(C.c_float)), ("node_cy", C.POINTER(C.c_float)),
        ("node_cz", C.POINTER(C.c_float)), ("node_radius", C.POINTER(C.c_float)),
        ("node_inside", C.POINTER(C.c_int32)), ("node_outside", C.POINTER(C.c_int32)),
        ("node_parent", C.POINTER(C.c_int32)), ("node_row", C.POINTER(C.c_uint32)),
        ("leaf_begin", C.POINTER(C.c_uint32)), ("leaf_count", C.POINTER(C.c_uint32)),
        ("slot_x", C.POINTER(C.c_float)), ("slot_y", C.POINTER(C.c_float)),
        ("slot_z", C.POINTER(C.c_float)), ("slot_row", C.POINTER(C.c_uint32)),
    ]


class TimingStruct(C.Structure):
    _fields_ = [("nn_launches", C.c_uint64), ("nn_ms", C.c_double), ("apply_ms", C.c_double),
                ("reduce_ms", C.c_double), ("host_ms", C.c_double), ("queries", C.c_double),
                ("nodes_visited", C.c_double), ("points_scanned", C.c_double),
                ("stack_spills", C.c_double)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int, C.c_void_p)
PROGRESS_FN = C.CFUNCTYPE(None, C.c_uint64, C.c_uint64, C.c_float, C.c_void_p)
ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p,
                           C.POINTER(C.c_uint64), C.c_int, C.c_void_p)

_lib = None


def load() -> C.CDLL:
    """Load libicp_hip.so (built in-tree by icp_amd/csrc/Makefile).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C icp_amd/csrc` "
            "(or __graft_entry__.build()); there is no fallback path")
    L = C.CDLL(LIB_PATH)
    vp, u64, u32, f32p = C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_float)
    treep = C.POINTER(FlatTreeStruct)
    L.icp_hip_tree_build.argtypes = [vp, u64, u32, C.POINTER(treep)]
    L.icp_hip_tree_free.argtypes = [treep]
    L.icp_hip_tree_free.restype = None
    L.icp_hip_tree_validate.argtypes = [treep]
    L.icp_hip_mean_host.argtypes = [vp, u64, vp]
    L.icp_hip_mean_host.restype = None
    L.icp_hip_center_host.argtypes = [vp, u64, vp, vp]
    L.icp_hip_center_host.restype = None
    L.icp_hip_alignment_host.argtypes = [vp, vp, vp, vp]
    L.icp_hip_alignment_host.restype = None
    f32pp = C.POINTER(C.POINTER(C.c_float))
    L.icp_hip_read_csv.argtypes = [C.c_char_p] * 4 + [f32pp, C.POINTER(u64)]
    L.icp_hip_read_f32.argtypes = [C.c_char_p, f32pp, C.POINTER(u64)]
    L.icp_hip_free_points.argtypes = [C.POINTER(C.c_float)]
    L.icp_hip_free_points.restype = None
    L.icp_hip_create.argtypes = [C.POINTER(vp), C.c_int]
    L.icp_hip_destroy.argtypes = [vp]
    L.icp_hip_destroy.restype = None
    L.icp_hip_last_error.argtypes = [vp]
    L.icp_hip_last_error.restype = C.c_char_p
    L.icp_hip_last_global_error.restype = C.c_char_p
    L.icp_hip_set_tree.argtypes = [vp, treep]
    L.icp_hip_build_tree.argtypes = [vp, vp, u64, u32]
    L.icp_hip_get_tree.argtypes = [vp, C.POINTER(treep)]
    L.icp_hip_get_tree_info.argtypes = [vp, treep]
    L.icp_hip_set_queries.argtypes = [vp, vp, u64, u64]
    L.icp_hip_set_queries_exchange.argtypes = [vp, vp, u64, u64]
    L.icp_hip_set_alltoallv.argtypes = [vp, C.c_int, C.c_int, ALLTOALLV_FN, vp]
    L.icp_hip_get_shard_size.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.icp_hip_nn.argtypes = [vp, vp, vp, vp, vp, C.POINTER(C.c_float)]
    L.icp_hip_nn_noalloc.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.icp_hip_set_allreduce.argtypes = [vp, ALLREDUCE_FN, vp]
    L.icp_hip_comm_unique_id.argtypes = [vp, C.c_size_t]
    L.icp_hip_comm_init_rank.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t]
    L.icp_hip_run.argtypes = [vp, vp, u64, C.c_float, vp, C.POINTER(u64), C.POINTER(C.c_float),
                              PROGRESS_FN, vp]
    L.icp_hip_get_points.argtypes = [vp, vp]
    L.icp_hip_icp.argtypes = [vp, vp, u64, vp, u64, u64, C.c_float, u32, vp, vp, C.POINTER(u64),
                              C.POINTER(C.c_float), PROGRESS_FN, vp]
    L.icp_hip_set_phase_timers.argtypes = [vp, C.c_int]
    L.icp_hip_get_timing.argtypes = [vp, C.POINTER(TimingStruct)]
    L.icp_hip_get_nn_trace.argtypes = [vp, vp, u64, C.POINTER(u64)]
    _lib = L
    return L


def _pts(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError("expected an (n, 3) float32 array of points")
    return a


def _check(rc: int, ctx=None):
    if rc != 0:
        L = load()
        msg = (L.icp_hip_last_error(ctx) if ctx else L.icp_hip_last_global_error()) or b""
        raise IcpHipError(rc, msg.decode(errors="replace"))


def _take_points(ptr, n) -> np.ndarray:
    try:
        return np.ctypeslib.as_array(ptr, shape=(n.value, 3)).copy() if n.value else \
            np.zeros((0, 3), np.float32)
    finally:
        load().icp_hip_free_points(ptr)


def read_csv(path: str, x_field="Points_0", y_field="Points_1", z_field="Points_2") -> np.ndarray:
    """libcpu::read_csv (point-3d.cc:105-134) with main's field names as defaults (main.cc:24-27)."""
    ptr, n = C.POINTER(C.c_float)(), C.c_uint64()
    _check(load().icp_hip_read_csv(os.fsencode(path), x_field.encode(), y_field.encode(),
                                   z_field.encode(), C.byref(ptr), C.byref(n)))
    return _take_points(ptr, n)


def read_f32(path: str) -> np.ndarray:
    ptr, n = C.POINTER(C.c_float)(), C.c_uint64()
    _check(load().icp_hip_read_f32(os.fsencode(path), C.byref(ptr), C.byref(n)))
    return _take_points(ptr, n)


def mean_host(xyz) -> np.ndarray:
    """libcpu::mean (point-3d.cc:177-192) on the host, bit-faithful."""
    a = _pts(xyz)
    out = np.zeros(3, np.float32)
    load().icp_hip_mean_host(a.ctypes.data, len(a), out.ctypes.data)
    return out


def center_host(xyz, mu) -> np.ndarray:
    a = _pts(xyz)
    mu = np.ascontiguousarray(mu, np.float32)
    out = np.empty_like(a)
    load().icp_hip_center_host(a.ctypes.data, len(a), mu.ctypes.data, out.ctypes.data)
    return out


def alignment_host(H, mu_p, mu_m) -> np.ndarray:
    """find_alignment after find_covariance (icp.cc:27-60) on the host: H(3x3) -> 4x4."""
    H = np.ascontiguousarray(H, np.float32).reshape(9)
    mu_p = np.ascontiguousarray(mu_p, np.float32)
    mu_m = np.ascontiguousarray(mu_m, np.float32)
    T = np.zeros(16, np.float32)
    load().icp_hip_alignment_host(H.ctypes.data, mu_p.ctypes.data, mu_m.ctypes.data, T.ctypes.data)
    return T.reshape(4, 4)


class FlatTree:
    """Owns an icp_flat_tree built by icp_hip_tree_build; numpy views into its arrays."""

    def __init__(self, xyz_centered, capacity: int, _adopt=None):
        L = load()
        if _adopt is not None:  # a tree the library handed out (icp_hip_get_tree)
            self._ptr = _adopt
            return
        a = _pts(xyz_centered)
        self._ptr = C.POINTER(FlatTreeStruct)()
        _check(L.icp_hip_tree_build(a.ctypes.data, len(a), int(capacity), C.byref(self._ptr)))

    @property
    def struct(self) -> FlatTreeStruct:
        return self._ptr.contents

    def __getattr__(self, name):
        s = self._ptr.contents
        if name in ("n_points", "n_slots", "n_nodes", "n_leaves", "height", "capacity",
                    "max_leaf_count", "root"):
            return getattr(s, name)
        sizes = {"node": s.n_nodes, "leaf": s.n_leaves, "slot": s.n_slots}
        kind = name.split("_")[0]
        if kind in sizes and hasattr(s, name):
            n = sizes[kind]
            if n == 0:
                return np.zeros(0, np.dtype(getattr(s, name)._type_))
            return np.ctypeslib.as_array(getattr(s, name), shape=(n,))
        raise AttributeError(name)

    def validate(self):
        _check(load().icp_hip_tree_validate(self._ptr))

    def close(self):
        if self._ptr:
            load().icp_hip_tree_free(self._ptr)
            self._ptr = C.POINTER(FlatTreeStruct)()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One GPU, one shard of queries, one replica of the tree."""

    def __init__(self, device_id: int = -1, phase_timers: bool = True):
        self._L = load()
        self._h = C.c_void_p()
        _check(self._L.icp_hip_create(C.byref(self._h), int(device_id)))
        self._keep = []
        self.n_local = 0
        # timing() / nn_trace() are part of this wrapper's interface (bench.py, tests); a C caller
        # gets a context without the per-iteration events (icp_hip_set_phase_timers)
        _check(self._L.icp_hip_set_phase_timers(self._h, 1 if phase_timers else 0), self._h)

    def close(self):
        if self._h:
            self._L.icp_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_tree(self, tree: FlatTree):
        _check(self._L.icp_hip_set_tree(self._h, tree._ptr), self._h)

    def build_tree(self, xyz_centered, capacity: int):
        """Build the reference's VP-tree on the device (icp_hip_build_tree)."""
        a = _pts(xyz_centered)
        _check(self._L.icp_hip_build_tree(self._h, a.ctypes.data, len(a), int(capacity)), self._h)

    def tree_info(self) -> dict:
        t = FlatTreeStruct()
        _check(self._L.icp_hip_get_tree_info(self._h, C.byref(t)), self._h)
        return {k: getattr(t, k) for k in ("n_points", "n_slots", "n_nodes", "n_leaves", "height",
                                           "capacity", "max_leaf_count", "root")}

    def get_tree(self) -> "FlatTree":
        ptr = C.POINTER(FlatTreeStruct)()
        _check(self._L.icp_hip_get_tree(self._h, C.byref(ptr)), self._h)
        return FlatTree(None, 0, _adopt=ptr)

    def set_tree_struct(self, struct: FlatTreeStruct):
        _check(self._L.icp_hip_set_tree(self._h, C.byref(struct)), self._h)

    def set_queries(self, xyz, n_total: int | None = None):
        a = _pts(xyz)
        self.n_local = len(a)
        _check(self._L.icp_hip_set_queries(self._h, a.ctypes.data, len(a),
                                           int(n_total if n_total is not None else len(a))),
               self._h)

    def set_alltoallv(self, n_ranks: int, rank: int, fn):
        """fn(send: bytes-like uint8 array, send_bytes: list[int], recv_bytes: list[int]) ->
        uint8 array of sum(recv_bytes) bytes: the blocks from ranks 0 .. n_ranks-1 back to back."""
        if fn is None:
            cb = C.cast(None, ALLTOALLV_FN)
        else:
            def _cb(send, send_bytes, recv, recv_bytes, n, _user):
                try:
                    sb = [int(send_bytes[r]) for r in range(n)]
                    rb = [int(recv_bytes[r]) for r in range(n)]
                    src = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(max(1, sum(sb)),))
                    out = np.asarray(fn(src[:sum(sb)], sb, rb), np.uint8)
                    assert out.size == sum(rb)
                    if out.size:
                        C.memmove(recv, out.ctypes.data, out.size)
                    return 0
                except Exception:  # never let an exception cross the C ABI
                    import traceback
                    traceback.print_exc()
                    return 1
            cb = ALLTOALLV_FN(_cb)
        self._keep.append(cb)
        _check(self._L.icp_hip_set_alltoallv(self._h, int(n_ranks), int(rank), cb, None), self._h)

    def set_queries_exchange(self, xyz, n_total: int):
        """icp_hip_set_queries_exchange: collective; this rank hands over ITS rows, registers the
        points of its region of the tree, and get_points() (collective too) returns its rows."""
        a = _pts(xyz)
        _check(self._L.icp_hip_set_queries_exchange(self._h, a.ctypes.data, len(a), int(n_total)), self._h)
        self.n_local = len(a)          # what get_points() returns
        self._exchanged = True

    def shard_size(self) -> tuple[int, int]:
        """(points this context registers, points of the whole cloud)."""
        a, b = C.c_uint64(), C.c_uint64()
        _check(self._L.icp_hip_get_shard_size(self._h, C.byref(a), C.byref(b)), self._h)
        return a.value, b.value

    def nn(self, mu_p, want=("row", "xyz", "dist")):
        """One NN pass; returns dict with the requested host arrays + kernel_ms."""
        mu = np.ascontiguousarray(mu_p, np.float32)
        n = self.n_local
        row = np.empty(n, np.uint32) if "row" in want else None
        xyz = np.empty((n, 3), np.float32) if "xyz" in want else None
        dist = np.empty(n, np.float32) if "dist" in want else None
        ms = C.c_float()
        _check(self._L.icp_hip_nn(self._h, mu.ctypes.data,
                                  row.ctypes.data if row is not None else None,
                                  xyz.ctypes.data if xyz is not None else None,
                                  dist.ctypes.data if dist is not None else None, C.byref(ms)),
               self._h)
        return dict(row=row, xyz=xyz, dist=dist, kernel_ms=ms.value)

    def nn_noalloc(self, d_px: int, d_py: int, d_pz: int, n: int, mu_p, d_row: int = 0,
                   d_dist: int = 0, d_yx: int = 0, d_yy: int = 0, d_yz: int = 0, d_sums: int = 0,
                   stream: int = 0):
        """icp_hip_nn_noalloc: every d_* is a DEVICE address (int, 0 = NULL) of a buffer the caller
        owns; enqueued on `stream` (a hipStream_t as int), nothing is waited for or allocated."""
        mu = np.ascontiguousarray(mu_p, np.float32)
        _check(self._L.icp_hip_nn_noalloc(self._h, d_px, d_py, d_pz, int(n), mu.ctypes.data,
                                          d_row or None, d_dist or None, d_yx or None,
                                          d_yy or None, d_yz or None, d_sums or None,
                                          stream or None), self._h)

    def set_allreduce(self, fn):
        """fn(np.ndarray[float64]) sums the array in place over all shards."""
        if fn is None:
            cb = C.cast(None, ALLREDUCE_FN)
        else:
            def _cb(ptr, count, _user):
                try:
                    fn(np.ctypeslib.as_array(ptr, shape=(count,)))
                    return 0
                except Exception:  # never let an exception cross the C ABI
                    import traceback
                    traceback.print_exc()
                    return 1
            cb = ALLREDUCE_FN(_cb)
        self._keep.append(cb)
        _check(self._L.icp_hip_set_allreduce(self._h, cb, None), self._h)

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _check(load().icp_hip_comm_unique_id(buf, 128))
        return buf.raw

    def comm_init_rank(self, n_ranks: int, rank: int, unique_id: bytes):
        buf = C.create_string_buffer(unique_id, 128)
        _check(self._L.icp_hip_comm_init_rank(self._h, n_ranks, rank, buf, 128), self._h)

    def run(self, mu_m, max_iters: int, threshold: float, progress=None):
        mu = np.ascontiguousarray(mu_m, np.float32)
        T = np.zeros(16, np.float32)
        it = C.c_uint64()
        err = C.c_float()
        if progress is None:
            cb = C.cast(None, PROGRESS_FN)
        else:
            cb = PROGRESS_FN(lambda i, m, e, _u: progress(i, m, e))
        _check(self._L.icp_hip_run(self._h, mu.ctypes.data, int(max_iters), float(threshold),
                                   T.ctypes.data, C.byref(it), C.byref(err), cb, None), self._h)
        return T.reshape(4, 4), it.value, err.value

    def icp(self, m, p, iterations: int, threshold: float, vp_threshold: int, progress=None):
        """libgpu::icp in one call (icp_hip_icp): returns (T, new_p, iterations, error)."""
        m, p = _pts(m), _pts(p)
        T = np.zeros(16, np.float32)
        new_p = np.empty_like(p)
        it, err = C.c_uint64(), C.c_float()
        cb = C.cast(None, PROGRESS_FN) if progress is None else \
            PROGRESS_FN(lambda i, mx, e, _u: progress(i, mx, e))
        _check(self._L.icp_hip_icp(self._h, m.ctypes.data, len(m), p.ctypes.data, len(p),
                                   int(iterations), float(threshold), int(vp_threshold),
                                   T.ctypes.data, new_p.ctypes.data, C.byref(it), C.byref(err),
                                   cb, None), self._h)
        self.n_local = len(p)
        return T.reshape(4, 4), new_p, it.value, err.value

    def get_points(self) -> np.ndarray:
        out = np.empty((self.n_local, 3), np.float32)
        _check(self._L.icp_hip_get_points(self._h, out.ctypes.data), self._h)
        return out

    def nn_trace(self) -> np.ndarray:
        """NN-kernel ms of each iteration of the last run()."""
        cnt = C.c_uint64()
        _check(self._L.icp_hip_get_nn_trace(self._h, None, 0, C.byref(cnt)), self._h)
        out = np.zeros(cnt.value, np.float32)
        if cnt.value:
            _check(self._L.icp_hip_get_nn_trace(self._h, out.ctypes.data, cnt.value,
                                                C.byref(cnt)), self._h)
        return out

    def timing(self) -> dict:
        t = TimingStruct()
        _check(self._L.icp_hip_get_timing(self._h, C.byref(t)), self._h)
        return {k: getattr(t, k) for k, _ in TimingStruct._fields_}
