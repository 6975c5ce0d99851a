"""ctypes bindings for the CHECKERS -- test infrastructure only.

* ``Oracle``  -> oracle/liboracle.so, the CPU restatement of the reference libcpu path
  (oracle/icp_oracle.cc).
* ``Ref``     -> oracle/_ref/libicp_ref.so, the reference's own libcpu sources compiled
  unmodified (only present where it was built: the build container; it travels to the GPU
  box as a prebuilt file but is never rebuilt there).

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may import this
package.  Nothing under icp_amd/ does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build(ref: bool = True) -> None:
    """Compile the checkers (building the checker is not using it)."""
    subprocess.run(["make", "-s", "-C", _HERE, "liboracle.so"], check=True)
    if ref and os.path.exists("/root/reference/libcpu/libcpu/vp-tree.cc"):
        subprocess.run(["make", "-s", "-C", _HERE, "ref"], check=True)


def _as_pts(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 3
    return a


class _Tree:
    def __init__(self, owner, handle):
        self._owner = owner
        self.handle = handle

    def __del__(self):
        if self.handle:
            self._owner._free(self.handle)
            self.handle = None


class Oracle:
    def __init__(self, path: str | None = None):
        path = path or os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = self.L = C.CDLL(path)
        vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
        L.orc_mean.argtypes = [_f32p, u64, _f32p]
        L.orc_subtract.argtypes = [_f32p, u64, _f32p, _f32p]
        L.orc_find_covariance.argtypes = [_f32p, _f32p, u64, _f32p]
        L.orc_tree_build.argtypes = [u32, _f32p, u64]
        L.orc_tree_build.restype = vp
        L.orc_tree_free.argtypes = [vp]
        L.orc_tree_degenerate.argtypes = [vp]
        L.orc_tree_stats.argtypes = [vp] + [C.POINTER(u64)] * 2 + [C.POINTER(u32)] + [C.POINTER(u64)] * 2
        L.orc_tree_closest.argtypes = [vp, _f32p, u64] + [vp] * 5
        L.orc_tree_dump_sizes.argtypes = [vp] + [C.POINTER(u64)] * 4
        L.orc_tree_dump.argtypes = [vp, _u8p, _f32p, _u32p, _u32p, _f32p, _u32p]
        L.orc_svd3.argtypes = [_f32p] * 4
        L.orc_alignment_from_cov.argtypes = [_f32p] * 4
        L.orc_apply_alignment.argtypes = [_f32p, u64, _f32p]
        L.orc_compute_error.argtypes = [_f32p, _f32p, u64, _f32p]
        L.orc_compute_error.restype = C.c_float
        L.orc_matmul4.argtypes = [_f32p] * 3
        L.orc_icp.argtypes = [_f32p, u64, _f32p, u64, u64, C.c_float, u32, _f32p, vp,
                              C.POINTER(u64), vp, vp]
        L.orc_icp.restype = C.c_int
        L.flat_walk_closest.argtypes = [vp, _f32p, u64, vp, vp, vp, vp]
        L.orc_tree_from_flat.argtypes = [C.c_int32] + [vp] * 13
        L.orc_tree_from_flat.restype = vp

    def max_threads(self) -> int:
        return int(self.L.orc_max_threads())

    def tree_from_flat(self, ft):
        """Pointer tree for the recursive search from an icp_amd.capi.FlatTree (no rebuild)."""
        s = ft.struct
        ptr = lambda name: C.cast(getattr(s, name), C.c_void_p)
        h = self.L.orc_tree_from_flat(
            s.root, ptr("node_cx"), ptr("node_cy"), ptr("node_cz"), ptr("node_radius"),
            ptr("node_inside"), ptr("node_outside"), ptr("node_row"), ptr("leaf_begin"),
            ptr("leaf_count"), ptr("slot_x"), ptr("slot_y"), ptr("slot_z"), ptr("slot_row"))
        return _Tree(self, h)

    def flat_walk_closest(self, flat_tree_ptr, q):
        """Host emulation of the kernel's traversal over an icp_flat_tree (oracle/flat_walk.cc).

        flat_tree_ptr: ctypes pointer to the icp_flat_tree struct."""
        q = _as_pts(q)
        n = len(q)
        ids = np.empty(n, np.uint32)
        row = np.empty(n, np.uint32)
        dist = np.empty(n, np.float32)
        xyz = np.empty((n, 3), np.float32)
        self.L.flat_walk_closest(C.cast(flat_tree_ptr, C.c_void_p), q, n, ids.ctypes.data,
                                 row.ctypes.data, dist.ctypes.data, xyz.ctypes.data)
        return dict(id=ids, row=row, dist=dist, xyz=xyz)

    def _free(self, h):
        self.L.orc_tree_free(h)

    def mean(self, a):
        a = _as_pts(a)
        out = np.zeros(3, np.float32)
        self.L.orc_mean(a, len(a), out)
        return out

    def subtract(self, a, mu):
        a = _as_pts(a)
        out = np.empty_like(a)
        self.L.orc_subtract(a, len(a), np.ascontiguousarray(mu, np.float32), out)
        return out

    def find_covariance(self, p, y):
        p, y = _as_pts(p), _as_pts(y)
        out = np.zeros(9, np.float32)
        self.L.orc_find_covariance(p, y, len(p), out)
        return out

    def tree_build(self, capacity, pts):
        pts = _as_pts(pts)
        h = self.L.orc_tree_build(capacity, pts, len(pts))
        if not h:
            raise ValueError("empty cloud")
        return _Tree(self, h)

    def tree_degenerate(self, t) -> bool:
        return bool(self.L.orc_tree_degenerate(t.handle))

    def tree_stats(self, t):
        a, b, mn, mx = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        h = C.c_uint32()
        self.L.orc_tree_stats(t.handle, a, b, h, mn, mx)
        return dict(internal=a.value, leaves=b.value, height=h.value, min_leaf=mn.value,
                    max_leaf=mx.value)

    def tree_closest(self, t, q, counters=False):
        """Returns dict(xyz, row, dist[, n_int, n_pts])."""
        q = _as_pts(q)
        n = len(q)
        xyz = np.empty((n, 3), np.float32)
        row = np.empty(n, np.uint32)
        dist = np.empty(n, np.float32)
        n_int = np.zeros(n, np.uint32) if counters else None
        n_pts = np.zeros(n, np.uint32) if counters else None
        self.L.orc_tree_closest(t.handle, q, n, xyz.ctypes.data, row.ctypes.data, dist.ctypes.data,
                                n_int.ctypes.data if counters else None,
                                n_pts.ctypes.data if counters else None)
        out = dict(xyz=xyz, row=row, dist=dist)
        if counters:
            out.update(n_int=n_int, n_pts=n_pts)
        return out

    def tree_dump(self, t):
        nk, ni, nl, nlp = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.L.orc_tree_dump_sizes(t.handle, nk, ni, nl, nlp)
        kind = np.zeros(nk.value, np.uint8)
        node_f = np.zeros((ni.value, 4), np.float32)
        node_row = np.zeros(ni.value, np.uint32)
        leaf_count = np.zeros(nl.value, np.uint32)
        leaf_xyz = np.zeros((nlp.value, 3), np.float32)
        leaf_row = np.zeros(nlp.value, np.uint32)
        self.L.orc_tree_dump(t.handle, kind, node_f, node_row, leaf_count, leaf_xyz, leaf_row)
        return dict(kind=kind, node_f=node_f, node_row=node_row, leaf_count=leaf_count,
                    leaf_xyz=leaf_xyz, leaf_row=leaf_row)

    def svd3(self, H):
        H = np.ascontiguousarray(H, np.float32).reshape(9)
        U, S, V = np.zeros(9, np.float32), np.zeros(3, np.float32), np.zeros(9, np.float32)
        self.L.orc_svd3(H, U, S, V)
        return U.reshape(3, 3), S, V.reshape(3, 3)

    def alignment_from_cov(self, H, mu_p, mu_m):
        T = np.zeros(16, np.float32)
        self.L.orc_alignment_from_cov(np.ascontiguousarray(H, np.float32).reshape(9),
                                      np.ascontiguousarray(mu_p, np.float32),
                                      np.ascontiguousarray(mu_m, np.float32), T)
        return T.reshape(4, 4)

    def apply_alignment(self, p, T):
        p = _as_pts(p).copy()
        self.L.orc_apply_alignment(p, len(p), np.ascontiguousarray(T, np.float32).reshape(16))
        return p

    def compute_error(self, y, p, mu_m):
        y, p = _as_pts(y), _as_pts(p)
        return float(self.L.orc_compute_error(y, p, len(p), np.ascontiguousarray(mu_m, np.float32)))

    def matmul4(self, a, b):
        out = np.zeros(16, np.float32)
        self.L.orc_matmul4(np.ascontiguousarray(a, np.float32).reshape(16),
                           np.ascontiguousarray(b, np.float32).reshape(16), out)
        return out.reshape(4, 4)

    def icp(self, m, p, iterations=200, threshold=1e-5, capacity=32, want_rows=False):
        """libcpu::icp restated.  Returns dict(T, new_p, iterations, errors[, rows])."""
        m, p = _as_pts(m), _as_pts(p)
        T = np.zeros(16, np.float32)
        new_p = np.empty_like(p)
        errors = np.full(max(int(iterations), 1), np.nan, np.float32)
        rows = np.zeros(len(p), np.uint32) if want_rows else None
        it = C.c_uint64()
        rc = self.L.orc_icp(m, len(m), p, len(p), int(iterations), float(threshold), int(capacity),
                            T, new_p.ctypes.data, it, errors.ctypes.data,
                            rows.ctypes.data if want_rows else None)
        if rc != 0:
            raise RuntimeError(f"orc_icp failed with status {rc} (degenerate tree)")
        out = dict(T=T.reshape(4, 4), new_p=new_p, iterations=it.value, errors=errors[:it.value])
        if want_rows:
            out["rows"] = rows
        return out


class Ref:
    """The real reference libcpu (build container only)."""

    PATH = os.path.join(_HERE, "_ref", "libicp_ref.so")

    @classmethod
    def available(cls) -> bool:
        return os.path.exists(cls.PATH)

    def __init__(self):
        L = self.L = C.CDLL(self.PATH)
        vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
        L.ref_read_csv.argtypes = [C.c_char_p, vp, u64]
        L.ref_read_csv.restype = C.c_int64
        L.ref_squared_distance.argtypes = [_f32p, _f32p]
        L.ref_squared_distance.restype = C.c_float
        L.ref_mean.argtypes = [_f32p, u64, _f32p]
        L.ref_subtract.argtypes = [_f32p, u64, _f32p, _f32p]
        L.ref_find_covariance.argtypes = [_f32p, _f32p, u64, _f32p]
        L.ref_matmul4.argtypes = [_f32p] * 3
        L.ref_translation.argtypes = [_f32p] * 4
        L.ref_tree_build.argtypes = [u32, _f32p, u64]
        L.ref_tree_build.restype = vp
        L.ref_tree_free.argtypes = [vp]
        L.ref_tree_closest.argtypes = [vp, _f32p, u64, _f32p]
        L.ref_tree_search_dist.argtypes = [vp, _f32p, u64, _f32p]
        L.ref_tree_dump_sizes.argtypes = [vp] + [C.POINTER(u64)] * 4
        L.ref_tree_dump.argtypes = [vp, _u8p, _f32p, _u32p, _f32p]

    def _free(self, h):
        self.L.ref_tree_free(h)

    def read_csv(self, path: str) -> np.ndarray:
        n = self.L.ref_read_csv(path.encode(), None, 0)
        if n < 0:
            raise RuntimeError(f"reference read_csv failed on {path}")
        out = np.empty((n, 3), np.float32)
        assert self.L.ref_read_csv(path.encode(), out.ctypes.data, n) == n
        return out

    def mean(self, a):
        a = _as_pts(a)
        out = np.zeros(3, np.float32)
        self.L.ref_mean(a, len(a), out)
        return out

    def subtract(self, a, mu):
        a = _as_pts(a)
        out = np.empty_like(a)
        self.L.ref_subtract(a, len(a), np.ascontiguousarray(mu, np.float32), out)
        return out

    def find_covariance(self, p, y):
        p, y = _as_pts(p), _as_pts(y)
        out = np.zeros(9, np.float32)
        self.L.ref_find_covariance(p, y, len(p), out)
        return out

    def matmul4(self, a, b):
        out = np.zeros(16, np.float32)
        self.L.ref_matmul4(np.ascontiguousarray(a, np.float32).reshape(16),
                           np.ascontiguousarray(b, np.float32).reshape(16), out)
        return out.reshape(4, 4)

    def translation(self, r, mu_p, mu_m):
        t = np.zeros(3, np.float32)
        self.L.ref_translation(np.ascontiguousarray(r, np.float32).reshape(9),
                               np.ascontiguousarray(mu_p, np.float32),
                               np.ascontiguousarray(mu_m, np.float32), t)
        return t

    def tree_build(self, capacity, pts):
        pts = _as_pts(pts)
        return _Tree(self, self.L.ref_tree_build(capacity, pts, len(pts)))

    def tree_closest(self, t, q):
        q = _as_pts(q)
        out = np.empty_like(q)
        self.L.ref_tree_closest(t.handle, q, len(q), out)
        return out

    def tree_search_dist(self, t, q):
        q = _as_pts(q)
        out = np.empty(len(q), np.float32)
        self.L.ref_tree_search_dist(t.handle, q, len(q), out)
        return out

    def tree_dump(self, t):
        nk, ni, nl, nlp = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.L.ref_tree_dump_sizes(t.handle, nk, ni, nl, nlp)
        kind = np.zeros(nk.value, np.uint8)
        node_f = np.zeros((ni.value, 4), np.float32)
        leaf_count = np.zeros(nl.value, np.uint32)
        leaf_xyz = np.zeros((nlp.value, 3), np.float32)
        self.L.ref_tree_dump(t.handle, kind, node_f, leaf_count, leaf_xyz)
        return dict(kind=kind, node_f=node_f, leaf_count=leaf_count, leaf_xyz=leaf_xyz)
