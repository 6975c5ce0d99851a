"""The reference-side binding, executed: integration/libgpu_icp.cc (libgpu::icp with the reference's
signature, libgpu/libgpu/icp.hh:19-21) linked with the reference's own point-3d.cc / vp-tree.cc and
libicp_hip.so into small drivers (integration/Makefile `driver`, built in the build container into
oracle/_ref/ because they contain compiled reference code; only the binaries travel).

CPU: the friend-class flattener (integration/flatten_from_libcpu.hh, the `friend class
libgpu::GPUVPTree` of libcpu/libcpu/vp-tree.hh:8-17) turns a real libcpu::VPTree into an
icp_flat_tree that validates and equals icp_hip_tree_build's tree array by array.
GPU: a caller that only knows libgpu::icp registers the reference's clouds and gets the transform
the C ABI gives, with the iteration counts of the CPU path.
"""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

REF_DIR = os.path.join(ROOT, "oracle", "_ref")
DIRECT = os.path.join(REF_DIR, "run_libgpu_icp")
VIA_LIBCPU = os.path.join(REF_DIR, "run_libgpu_icp_libcpu_tree")


def need(path):
    if not os.path.exists(path):
        pytest.skip(f"{os.path.relpath(path, ROOT)} not built (make -C integration driver)")


def raw(tmp_path, clouds, name):
    path = str(tmp_path / (name + ".f32"))
    np.ascontiguousarray(clouds[name], np.float32).tofile(path)
    return path


def run(binary, *args):
    env = dict(os.environ, OMP_NUM_THREADS="4")
    return subprocess.run([binary, *args], capture_output=True, text=True, timeout=600, env=env)


@pytest.mark.parametrize("name,cap", [("line_ref", 8), ("cow_ref", 8), ("cow_ref", 256),
                                      ("horse_ref", 32), ("horse_ref", 256)])
def test_friend_class_flattener_gives_the_library_tree(tmp_path, clouds, name, cap):
    need(VIA_LIBCPU)
    m = raw(tmp_path, clouds, name)
    r = run(VIA_LIBCPU, m, m, str(cap), "--check-tree")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "flattened libcpu::VPTree == icp_hip_tree_build" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("binary", [DIRECT, VIA_LIBCPU])
@pytest.mark.parametrize("mname,pname,cap", [("cow_ref", "cow_tr1", 256), ("cow_ref", "cow_tr2", 256),
                                             ("horse_ref", "horse_tr1", 32),
                                             ("line_ref", "line_translated_2_3_4", 8)])
def test_libgpu_icp_through_the_binding(tmp_path, clouds, binary, mname, pname, cap):
    need(binary)
    import icp_amd
    expected = json.load(open(os.path.join(GOLDEN, "icp_expected.json")))[f"{pname}_cap{cap}"]
    r = run(binary, raw(tmp_path, clouds, mname), raw(tmp_path, clouds, pname), str(cap))
    assert r.returncode == 0, r.stdout + r.stderr
    # the reference's progress lines on stderr (icp.cu:146-147,157)
    starts = [l for l in r.stderr.splitlines() if l.startswith("Starting iter ")]
    assert len(starts) == expected["iterations"] and starts[0] == "Starting iter 1/200"
    assert sum(l.startswith("Error: ") for l in r.stderr.splitlines()) == expected["iterations"]
    out = {l.split()[0]: l.split()[1:] for l in r.stdout.splitlines() if l and l[0] in "Tm"}
    T = np.array(out["T"], np.float32).reshape(4, 4)
    # the same call through the Python mirror of the C ABI: same bits
    T2, new_p, info = icp_amd.icp(clouds[mname], clouds[pname], 200, 1e-5, cap, return_info=True)
    assert info["iterations"] == expected["iterations"]
    assert (T.view(np.uint32) == T2.view(np.uint32)).all()
    # test_functionnal.cc:14-20's criterion on the moved cloud
    assert float(out["max_abs_moved_minus_reference"][0]) < 5e-3
