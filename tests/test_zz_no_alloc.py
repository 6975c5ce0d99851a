"""icp_hip_nn_noalloc, the entry point that allocates nothing (static device scratch), through
the C ABI: its rows, distances, matched points and sums must equal icp_hip_nn's bit for bit, on
two calls in a row (the second with a smaller input: the scratch array keeps the first call's
contents), and beyond its stated limits it must refuse.  The caller-side device buffers and the
stream come straight from the HIP runtime (ctypes): the caller of this entry point is plain HIP
code, and the test process stays free of a second framework's runtime."""
import ctypes as C

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


class Hip:
    """The few HIP runtime calls a caller of the entry point makes itself."""

    def __init__(self):
        from icp_amd import capi
        capi.load()  # libicp_hip.so brings the runtime in
        self.rt = self._runtime()
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.rt.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.rt.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.rt.hipStreamDestroy.argtypes = [C.c_void_p]
        self.rt.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        self.bufs = []

    @staticmethod
    def _runtime():
        """The HIP runtime libicp_hip.so is linked against (whatever its soname on this ROCm)."""
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.split()[-1]
                if "libamdhip64.so" in path:
                    return C.CDLL(path)
        import ctypes.util
        return C.CDLL(ctypes.util.find_library("amdhip64") or "libamdhip64.so")

    def ok(self, rc):
        assert rc == 0, f"HIP error {rc}"

    def upload(self, a) -> int:
        a = np.ascontiguousarray(a)
        return self._fill(a.nbytes, src=a)

    def filled(self, nbytes: int, byte: int) -> int:
        return self._fill(nbytes, byte=byte)

    def _fill(self, nbytes, src=None, byte=0) -> int:
        p = C.c_void_p()
        self.ok(self.rt.hipMalloc(C.byref(p), max(nbytes, 4)))
        self.bufs.append(p)
        if src is not None:
            self.ok(self.rt.hipMemcpy(p, src.ctypes.data, nbytes, 1))
        else:
            self.ok(self.rt.hipMemset(p, byte, nbytes))
        return p.value

    def download(self, ptr: int, dtype, count: int):
        out = np.empty(count, dtype)
        self.ok(self.rt.hipMemcpy(out.ctypes.data, C.c_void_p(ptr), out.nbytes, 2))
        return out

    def stream(self) -> int:
        # one stream for the whole module: the first launch on a new stream makes the runtime
        # allocate that queue's own buffers (8 MB), which is not the entry point's doing
        if getattr(self, "_stream", None) is None:
            s = C.c_void_p()
            self.ok(self.rt.hipStreamCreate(C.byref(s)))
            self._stream = s.value
        return self._stream

    def sync(self, stream: int):
        self.ok(self.rt.hipStreamSynchronize(C.c_void_p(stream)))

    def free_bytes(self) -> int:
        free, total = C.c_size_t(), C.c_size_t()
        self.ok(self.rt.hipMemGetInfo(C.byref(free), C.byref(total)))
        return free.value

    def release(self):
        for p in self.bufs:
            self.rt.hipFree(p)
        self.bufs = []


@pytest.fixture(scope="module")
def hip():
    h = Hip()
    yield h
    h.release()


_warm = []


def _run_noalloc(hip, ctx, p, mu_p, stream):
    n = len(p)
    px, py, pz = (hip.upload(p[:, k]) for k in range(3))
    row, dist = hip.filled(4 * n, 0xFF), hip.filled(4 * n, 0xFF)
    y = [hip.filled(4 * n, 0xFF) for _ in range(3)]
    sums = hip.filled(8 * 12, 0xFF)
    free_before = hip.free_bytes()
    ctx.nn_noalloc(px, py, pz, n, mu_p, row, dist, y[0], y[1], y[2], sums, stream)
    hip.sync(stream)
    # nothing was allocated on the device (the first launch of a kernel in a process is left
    # out: the HIP runtime may then set up the hardware queue's private segment).  Free memory
    # may GROW across the call: the runtime hands back, at a synchronisation, blocks that
    # contexts destroyed by earlier tests had released (seen: +2 MiB after other test files).
    if _warm:
        assert hip.free_bytes() >= free_before
    _warm.append(1)
    out = dict(row=hip.download(row, np.uint32, n), dist=hip.download(dist, np.float32, n),
               xyz=np.stack([hip.download(t, np.float32, n) for t in y], 1),
               sums=hip.download(sums, np.float64, 12))
    hip.release()
    return out


def test_noalloc_matches_icp_hip_nn_twice_in_a_row(hip, oracle):
    from icp_amd import capi, synth
    n_m, cap = 400_000, 256
    m = synth.cloud(n_m, seed=77)
    mu_m = capi.mean_host(m)
    mc = capi.center_host(m, mu_m)
    p_all = synth.apply_fp32(m, synth.ground_truth(3.0))
    stream = hip.stream()
    with capi.Context() as ctx:
        ctx.build_tree(mc, cap)
        # second call smaller than the first; third: fewer queries than one wave
        for n in (300_000, 70_001, 37):
            p = p_all[:n]
            mu_p = p.mean(0, dtype=np.float64).astype(np.float32)
            got = _run_noalloc(hip, ctx, p, mu_p, stream)
            ctx.set_queries(p)
            want = ctx.nn(mu_p)
            tm = ctx.timing()
            assert (got["row"] == want["row"]).all()
            assert (bits(got["dist"]) == bits(want["dist"])).all()
            assert (bits(got["xyz"]) == bits(want["xyz"])).all()
            assert got["sums"][9] == tm["nodes_visited"] and got["sums"][10] == tm["points_scanned"]
            # the covariance sums: the same fp32 products, summed in fp64 in another order
            q = p - mu_p
            H = (q[:, :, None] * want["xyz"][:, None, :]).astype(np.float32).astype(np.float64).sum(0)
            assert np.allclose(got["sums"][:9], H.reshape(9), rtol=1e-12, atol=1e-9)
        # and against the oracle directly
        p = p_all[:5000]
        mu_p = oracle.mean(p)
        got = _run_noalloc(hip, ctx, p, mu_p, stream)
        ref = oracle.tree_closest(oracle.tree_build(cap, mc), oracle.subtract(p, mu_p))
        assert (got["row"] == ref["row"]).all() and (bits(got["dist"]) == bits(ref["dist"])).all()


def test_covariance_sums_against_the_reference_golden(hip, clouds):
    """find_covariance (libcpu/libcpu/point-3d.cc:212-236) on the GPU, H itself: the nine sums the
    NN kernel returns for horse_tr1 against the covariance the compiled reference computed for the
    same clouds (golden; sequential fp32 sums, hence the drift bound) and against the exact sum of
    the same fp32 products."""
    import os
    from conftest import GOLDEN
    from icp_amd import capi
    g = np.load(os.path.join(GOLDEN, "nn_horse_tr1_cap32.npz"))
    m, p = clouds["horse_ref"], clouds["horse_tr1"]
    mu_m = capi.mean_host(m)
    mc = capi.center_host(m, mu_m)
    stream = hip.stream()
    with capi.Context() as ctx:
        ctx.build_tree(mc, 32)
        got = _run_noalloc(hip, ctx, p, g["mu_p"], stream)
    assert (got["row"] == g["row"]).all()
    H = got["sums"][:9]
    q = (p - g["mu_p"]).astype(np.float32)
    y = mc[g["row"]]
    exact = (q[:, :, None] * y[:, None, :]).astype(np.float32).astype(np.float64).sum(0).reshape(9)
    assert np.allclose(H, exact, rtol=1e-12, atol=1e-9)
    assert np.abs(H - g["cov"].astype(np.float64).reshape(9)).max() < 2e-3  # fp32 sequential drift
    assert got["sums"][9] == float(g["n_int"].astype(np.int64).sum())
    assert got["sums"][10] == float(g["n_pts"].astype(np.int64).sum())


def test_noalloc_outputs_are_optional_and_limits_are_enforced(hip):
    from icp_amd import capi, synth
    m = synth.cloud(50_000, seed=5)
    mu_m = capi.mean_host(m)
    mc = capi.center_host(m, mu_m)
    with capi.Context() as ctx:
        ctx.build_tree(mc, 64)
        px, py, pz = (hip.upload(m[:, k]) for k in range(3))
        row = hip.filled(4 * len(m), 0)
        ctx.nn_noalloc(px, py, pz, len(m), mu_m, d_row=row)  # the default stream
        hip.sync(0)
        # the cloud against its own tree: every point finds itself (distance 0), or a duplicate
        got = hip.download(row, np.uint32, len(m))
        assert (bits(mc[got]) == bits(mc)).all()
        big = (1 << 20) + 1
        with pytest.raises(capi.IcpHipError) as e:
            ctx.nn_noalloc(px, py, pz, big, mu_m, d_row=row)
        assert e.value.status == 1 and "1048576" in str(e.value)
        with pytest.raises(capi.IcpHipError):
            ctx.nn_noalloc(px, py, pz, 10, mu_m, d_yx=row)
    with capi.Context() as empty:
        with pytest.raises(capi.IcpHipError):
            empty.nn_noalloc(px, py, pz, 10, mu_m)
    hip.release()


def _chain_tree(levels: int):
    """A hand-made tree `levels` edges deep (no build makes one: the reference splits at the median):
    node i sits at x = 10 i with radius 5, holds one point 1 away inside and everything further out
    outside.  Returns (struct, arrays kept alive, the points by row)."""
    from icp_amd import capi
    nn, nl = levels, levels + 1
    f32, i32, u32 = np.float32, np.int32, np.uint32
    keep = dict(
        node_cx=(10.0 * np.arange(nn)).astype(f32), node_cy=np.zeros(nn, f32), node_cz=np.zeros(nn, f32),
        node_radius=np.full(nn, 5.0, f32),
        node_inside=(-1 - np.arange(nn)).astype(i32),                       # leaf i
        node_outside=np.append(np.arange(1, nn), -1 - nn).astype(i32),      # node i + 1; the last: leaf nn
        node_parent=(np.arange(nn) - 1).astype(i32), node_row=np.arange(nn).astype(u32),
        leaf_begin=(16 * np.arange(nl)).astype(u32), leaf_count=np.ones(nl, u32),
        slot_x=np.full(16 * nl, np.inf, f32), slot_y=np.full(16 * nl, np.inf, f32),
        slot_z=np.full(16 * nl, np.inf, f32), slot_row=np.full(16 * nl, 0xFFFFFFFF, u32))
    keep["slot_x"][::16] = np.append(10.0 * np.arange(nn) + 1.0, 10.0 * nn).astype(f32)
    keep["slot_y"][::16] = 0.0
    keep["slot_z"][::16] = 0.0
    keep["slot_row"][::16] = nn + np.arange(nl)
    s = capi.FlatTreeStruct(n_points=nn + nl, n_slots=16 * nl, n_nodes=nn, n_leaves=nl, height=levels,
                            capacity=2, max_leaf_count=1, root=0)
    for name, arr in keep.items():
        setattr(s, name, arr.ctypes.data_as(type(getattr(s, name))))
    pts = np.zeros((nn + nl, 3), f32)
    pts[:nn, 0] = keep["node_cx"]
    pts[nn:, 0] = keep["slot_x"][::16]
    return s, keep, pts


def test_noalloc_refuses_a_tree_deeper_than_its_static_stack(hip):
    """Height 33: icp_hip_nn serves it (its overflow stack is allocated by depth, the answers are
    the brute-force ones), icp_hip_nn_noalloc says ICP_HIP_ERR_DEPTH (2 on-chip entries + 30 in
    the static array = 32 levels) and names icp_hip_nn."""
    from icp_amd import capi
    s, keep, pts = _chain_tree(33)
    assert capi.load().icp_hip_tree_validate(C.byref(s)) == 0
    rng = np.random.default_rng(3)
    q = np.stack([rng.uniform(-5.0, 340.0, 500), rng.normal(0, 2.0, 500), rng.normal(0, 2.0, 500)],
                 1).astype(np.float32)
    mu = np.zeros(3, np.float32)
    with capi.Context() as ctx:
        ctx.set_tree_struct(s)
        ctx.set_queries(q)
        got = ctx.nn(mu)
        d2 = ((q[:, None, :].astype(np.float64) - pts[None, :, :]) ** 2).sum(2)
        assert (got["row"] == d2.argmin(1)).all()
        px, py, pz = (hip.upload(q[:, k]) for k in range(3))
        row = hip.filled(4 * len(q), 0)
        with pytest.raises(capi.IcpHipError) as e:
            ctx.nn_noalloc(px, py, pz, len(q), mu, d_row=row)
        assert e.value.status == 5 and "32 levels" in str(e.value) and "icp_hip_nn" in str(e.value)
    s32, keep32, _ = _chain_tree(32)
    with capi.Context() as ctx:
        ctx.set_tree_struct(s32)
        ctx.nn_noalloc(px, py, pz, len(q), mu, d_row=row)  # 32 levels: the limit itself is served
        hip.sync(0)
        ctx.set_queries(q)
        assert (hip.download(row, np.uint32, len(q)) == ctx.nn(mu)["row"]).all()
    hip.release()
