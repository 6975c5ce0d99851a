"""The NN kernel's ways back up the tree (icp_amd/csrc/nn_tile.hip: jump to the start of the
descent, jump to the one level that can matter, or climb level by level), restated on the host in
scripts/emulate_level_skip.cc and compared, query by query, with the plain climb of
libcpu/libcpu/vp-tree.cc:125-175: same answer (slot and distance bits) and the same count of
nodes visited, on a misaligned, a nearly aligned and a noise-level aligned cloud, with buckets
small enough that far-side searches nest.  The same program runs the kernel's ring walk (the
bucket scan that starts at the ring of the query's distance to the bucket's parent centre and
stops when the gap to the next ring exceeds the best distance found; ring_walk in nn_tile.hip,
with the kernel's arithmetic: ring distances cut to 16 mantissa bits, its slack, its stop rule)
on every query's first bucket against the full scan of that bucket.  The emulation exits non-zero
on any mismatch."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EXE = os.path.join(HERE, "_build", "emulate_way_up")
SRC = os.path.join(ROOT, "scripts", "emulate_level_skip.cc")
LIB = os.path.join(ROOT, "icp_amd", "lib")


def build_emulation():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    subprocess.run(["g++", "-std=c++17", "-O2", "-fopenmp", "-I" + os.path.join(ROOT, "include"), SRC,
                    "-L" + LIB, "-licp_hip", "-Wl,-rpath," + LIB, "-Wl,-rpath,/opt/rocm/lib",
                    "-o", EXE], check=True)


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(os.path.join(LIB, "libicp_hip.so")):
        pytest.skip("libicp_hip.so is not built")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(SRC):
        build_emulation()
    return EXE


@pytest.mark.parametrize("args", [
    ("200000", "16", "2.0"),                              # misaligned: nested far-side searches
    ("300000", "64", "0.05"),                             # nearly aligned
    ("300000", "256", "0", "1.25", "0", "0.000002"),      # aligned up to noise: the one-level jumps
])
def test_every_way_up_gives_the_plain_climbs_answer(exe, args):
    env = dict(os.environ, OMP_NUM_THREADS="4")
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches 0" in r.stdout
    assert "different from the full scan: 0" in r.stdout
    assert "heap-ordered 1" in r.stdout
