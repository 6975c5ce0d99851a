"""The registration loop's scalar steps (icp_amd/csrc/loop_steps.hh: init, solve, finish, advance,
nn_idle -- product code, the same functions the device runs as kernel tails) driven on the CPU
around the oracle's NN pass, in both orders the product uses:

  one GPU   NN(k) -> solve(k) -> apply(k) -> finish(k)                      (libcpu/libcpu/icp.cc:110-125)
  sharded   NN(k) -> ONE all-reduce of 16 doubles [NN sums of k | apply sums of k-1]
                  -> advance = finish(k-1), solve(k) -> apply(k)            k iterations = k + 1 slots

What the NN pass subtracts: on one GPU mean(p) of the cloud as it stands, every iteration, as the
reference does (icp.cc:114-115); a sharded state (LoopState::sharded) takes mu_m from the second
iteration on, because its pass starts before the moved cloud has been summed over the ranks.

The sums are added exactly (math.fsum), so a sharded state stepped in either order and any split
into shards must give the same bits; the one-GPU state must agree with the CPU path (oracle.icp)
on the iteration count and within 1e-5 on T, and differ from the sharded one by no more than the
rounding of the centroid.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "_build", "libloop_steps_host.so")


def build_harness():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared",
                    "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "icp_amd", "csrc"),
                    os.path.join(HERE, "csrc", "loop_steps_host.cc"), "-o", SO], check=True)


@pytest.fixture(scope="module")
def ls():
    src = os.path.join(HERE, "csrc", "loop_steps_host.cc")
    hdr = os.path.join(ROOT, "icp_amd", "csrc", "loop_steps.hh")
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        if not os.path.isdir("/opt/rocm/include"):
            pytest.skip("no HIP headers to build the harness with")
        build_harness()
    lib = C.CDLL(SO)
    lib.ls_init.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_uint, C.c_double, C.c_int]
    for f in (lib.ls_solve, lib.ls_finish, lib.ls_advance):
        f.argtypes = [C.c_void_p, C.c_void_p]
    lib.ls_nn_idle.argtypes = [C.c_void_p]
    lib.ls_get.argtypes = [C.c_void_p] * 10
    assert lib.ls_iter_sums() == 16 and lib.ls_apply_at() == 12
    return lib


class State:
    def __init__(self, lib, sums4, mu_m, thr, cap, n_total, sharded):
        self.lib = lib
        self.buf = C.create_string_buffer(lib.ls_state_size())
        s = np.asarray(sums4, np.float64)
        lib.ls_init(self.buf, s.ctypes.data, np.asarray(mu_m, np.float32).ctypes.data, thr, cap, float(n_total),
                    int(sharded))

    def call(self, fn, sums):
        s = np.ascontiguousarray(sums, np.float64)
        fn(self.buf, s.ctypes.data)

    def get(self):
        T, Tn, mq, mp = (np.zeros(k, np.float32) for k in (16, 12, 3, 3))
        it, st, dn = C.c_uint(), C.c_uint(), C.c_uint()
        err = C.c_float()
        work = np.zeros(3)
        self.lib.ls_get(self.buf, T.ctypes.data, Tn.ctypes.data, mq.ctypes.data, mp.ctypes.data, C.byref(it),
                        C.byref(st), C.byref(dn), C.byref(err), work.ctypes.data)
        return dict(T=T.reshape(4, 4), Tn=Tn.reshape(3, 4), mu_q=mq, mu_p=mp, it=it.value, started=st.value,
                    done=dn.value, error=err.value, work=work)


def fsum_cols(a):
    return [math.fsum(a[:, k].astype(np.float64).tolist()) for k in range(a.shape[1])]


def nn_terms(oracle, otree, p, mu_q):
    """Per-point terms of the NN pass' twelve sums, and the matched points."""
    q = (p - mu_q.astype(np.float32)).astype(np.float32)
    r = oracle.tree_closest(otree, q, counters=True)
    y = r["xyz"]
    prod = (q[:, :, None] * y[:, None, :]).astype(np.float32).reshape(-1, 9)
    terms = np.concatenate([prod.astype(np.float64), r["n_int"][:, None].astype(np.float64),
                            r["n_pts"][:, None].astype(np.float64), np.zeros((len(p), 1))], axis=1)
    return terms, y


def apply_terms(p, y, Tn, mu_m):
    """p <- Tn p as apply_error_kernel does it (fp32, left to right, no FMA) and the per-point
    terms of its four sums."""
    x, yy, z = p[:, 0], p[:, 1], p[:, 2]
    new = np.empty_like(p)
    for r in range(3):
        new[:, r] = ((x * Tn[r, 0] + yy * Tn[r, 1]) + z * Tn[r, 2]) + Tn[r, 3]
    e = (y + mu_m.astype(np.float32)) - new
    err = (e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]) + e[:, 2] * e[:, 2]
    return new, np.concatenate([err[:, None], new], axis=1).astype(np.float64)


def run_plain(lib, oracle, otree, p, mu_m, cap, thr, sharded):
    st = State(lib, [0.0] + fsum_cols(p), mu_m, thr, cap, len(p), sharded)
    p = p.copy()
    errors = []
    while not st.get()["done"]:
        mu_q = st.get()["mu_q"]
        mean_p = (np.array(fsum_cols(p)) / float(len(p))).astype(np.float32)
        if not sharded or not errors:       # mean(new_p), icp.cc:114
            assert (mu_q.view(np.uint32) == mean_p.view(np.uint32)).all()
        else:                               # mu_m, where t = mu_m - r mu_p has put the centroid
            assert (mu_q.view(np.uint32) == np.asarray(mu_m, np.float32).view(np.uint32)).all()
            assert np.abs(mean_p - mu_q).max() <= 3e-7 * max(1.0, np.abs(p).max())
        terms, y = nn_terms(oracle, otree, p, mu_q)
        st.call(lib.ls_solve, fsum_cols(terms))
        p, at = apply_terms(p, y, st.get()["Tn"], mu_m)
        st.call(lib.ls_finish, fsum_cols(at))
        errors.append(st.get()["error"])
    return st.get(), p, errors, None


def run_sharded(lib, oracle, otree, p, mu_m, cap, thr, cuts):
    shards = [p[a:b].copy() for a, b in zip([0] + cuts, cuts + [len(p)])]
    st = State(lib, [0.0] + fsum_cols(p), mu_m, thr, cap, len(p), True)
    pending = [0.0] * 4          # apply sums of the iteration before (zeros in the first slot)
    slots = 0
    errors = []
    while not st.get()["done"]:
        s0 = st.get()
        if lib.ls_nn_idle(st.buf):
            nn, ys = [0.0] * 12, None     # the cap is reached: only the last error is outstanding
        else:
            parts = [nn_terms(oracle, otree, sh, s0["mu_q"]) for sh in shards]
            nn = fsum_cols(np.concatenate([t for t, _ in parts]))
            ys = [y for _, y in parts]
        st.call(lib.ls_advance, nn + pending)      # the one all-reduce of the slot
        slots += 1
        s1 = st.get()
        if s1["it"] > s0["it"]:
            errors.append(s1["error"])
        if s1["done"]:
            break
        applied = [apply_terms(sh, y, s1["Tn"], mu_m) for sh, y in zip(shards, ys)]
        shards = [a for a, _ in applied]
        pending = fsum_cols(np.concatenate([t for _, t in applied]))
    return st.get(), np.concatenate(shards), errors, slots


@pytest.mark.parametrize("pair,capacity,cap_iters,thr", [
    ("cow_tr2", 32, 200, 1e-5), ("cow_tr1", 256, 200, 1e-5), ("cow_tr2", 32, 4, 0.0), ("cow_tr2", 32, 1, 0.0),
    ("cow_tr2", 32, 0, 1e-5), ("cow_tr2", 32, 200, float("inf")),
])
def test_both_orders_give_the_same_bits(ls, oracle, clouds, pair, capacity, cap_iters, thr):
    m, p = clouds["cow_ref"], clouds[pair]
    mu_m = oracle.mean(m)
    otree = oracle.tree_build(capacity, oracle.subtract(m, mu_m))
    ref, pref, eref, _ = run_plain(ls, oracle, otree, p, mu_m, cap_iters, thr, False)
    a, pa, ea, _ = run_plain(ls, oracle, otree, p, mu_m, cap_iters, thr, True)
    # the two centring rules: same iteration count, the same transform up to the centroid's rounding
    assert ref["it"] == a["it"] and np.abs(ref["T"] - a["T"]).max() <= 2e-6
    assert np.abs(pref - pa).max() <= 2e-6 * max(1.0, np.abs(pa).max())
    for cuts in ([1000], [5, 2000, 2001]):      # two shards; four, one of them a single point
        b, pb, eb, slots = run_sharded(ls, oracle, otree, p, mu_m, cap_iters, thr, list(cuts))
        assert a["it"] == b["it"] and ea == eb
        assert (a["T"].view(np.uint32) == b["T"].view(np.uint32)).all()
        assert (pa.view(np.uint32) == pb.view(np.uint32)).all()
        assert (a["work"] == b["work"]).all()
        # k iterations take k + 1 slots; none at all when the loop never starts
        assert slots == (a["it"] + 1 if a["it"] else 0)
    if thr == 1e-5 and cap_iters == 200:
        cpu = oracle.icp(m, p, cap_iters, thr, capacity)
        for got, moved in ((ref, pref), (a, pa)):
            assert got["it"] == cpu["iterations"]
            assert np.abs(got["T"] - cpu["T"]).max() <= 1e-5
            assert np.abs(moved - m).max() < 5e-3          # test_functionnal.cc's tolerance
    else:
        assert a["it"] == (0 if (cap_iters == 0 or thr == float("inf")) else cap_iters)
        if a["it"] == 0:
            assert (a["T"] == np.eye(4, dtype=np.float32)).all() and math.isinf(a["error"])
