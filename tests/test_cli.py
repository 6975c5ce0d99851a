"""cli/icp: the reference's command line (main/main/main.cc, options.cc) over the C ABI.

CPU part: argument handling, messages and exit codes as the reference's parse_options produces
them (options.cc:16-180).  GPU part: a run on the cow clouds written back to CSV; stdout carries
"Transformation: ", the 4x4 and "Points" (main.cc:42-51), stderr the parameters, the per-iteration
lines, the timing and ten point pairs; the printed transform is icp_amd.icp's."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

CLI = os.path.join(ROOT, "cli", "icp")


def run(*args):
    return subprocess.run([CLI, *args], capture_output=True, text=True, timeout=120)


def write_csv(path, pts, extra=False):
    with open(path, "w") as f:
        f.write("Points_0,Points_1,Points_2" + (",Points_Magnitude,Point ID" if extra else "") + "\n")
        for i, p in enumerate(pts):
            f.write(",".join("%.9g" % v for v in p) + (",1.0,%d" % i if extra else "") + "\n")


@pytest.mark.skipif(not os.path.exists(CLI), reason="cli/icp not built (make -C cli)")
class TestArguments:
    def test_usage_when_positionals_are_missing(self):
        for args in ((), ("only_one.txt",)):
            r = run(*args)
            assert r.returncode == 1
            assert r.stderr.startswith("Usage: ")
            for flag in ("--device cpu|gpu", "--capacity <capacity>", "--iterations <iterations>",
                         "--error <error>"):
                assert flag in r.stderr

    @pytest.mark.parametrize("args,msg", [
        (("--device",), "Missing device."), (("--device", "tpu"), "Invalid device: tpu"),
        (("--capacity",), "Missing capacity."), (("--capacity", "-3"), "Invalid capacity: -3"),
        (("--capacity", "abc"), "Invalid capacity: abc"),
        (("--iterations", "-1"), "Invalid iterations: -1"), (("--error", "x"), "Invalid error: x"),
        (("--error", "-1"), "Invalid error: -1"), (("--bogus",), "Invalid argument: --bogus"),
    ])
    def test_bad_options(self, args, msg):
        r = run("a.txt", "b.txt", *args)
        assert r.returncode == 1
        assert r.stderr.splitlines()[0] == msg
        assert "Usage: " in r.stderr

    def test_cpu_device_defaults_then_refused(self):
        r = run("a.txt", "b.txt", "--device", "cpu", "--iterations", "7")
        assert r.returncode == 1
        lines = r.stderr.splitlines()
        assert lines[:5] == ["Parameters:", "    Device: CPU", "    Capacity: 32",
                             "    Max iterations: 7", "    Error threshold: 1e-05"]


@pytest.mark.gpu
def test_cli_registers_cow(tmp_path, clouds):
    import icp_amd
    m, p = clouds["cow_ref"], clouds["cow_tr2"]
    write_csv(tmp_path / "ref.txt", m, extra=True)
    write_csv(tmp_path / "tr.txt", p)
    r = run(str(tmp_path / "ref.txt"), str(tmp_path / "tr.txt"))
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout.splitlines()
    assert out[0] == "Transformation: " and out[5] == "Points" and len(out) == 6
    T_cli = np.array([[float(v) for v in line.split(" ")] for line in out[1:5]])
    T, new_p, info = icp_amd.icp(m, p, 200, 1e-5, 256, return_info=True)
    assert np.allclose(T_cli, T, rtol=2e-5, atol=1e-6)  # printed with 6 significant digits
    err = r.stderr.splitlines()
    assert err[:5] == ["Parameters:", "    Device: GPU", "    Capacity: 256",
                       "    Max iterations: 200", "    Error threshold: 1e-05"]
    assert err[5] == "Starting iter 1/200" and err[6].startswith("Error: ")
    assert sum(l.startswith("Starting iter ") for l in err) == info["iterations"] == 11
    assert any(l.startswith("Execution time: ") and l.endswith("ms") for l in err)
    pairs = [l for l in err if l.startswith("(") and ") - (" in l]
    assert len(pairs) == 10
    first = pairs[0].replace("(", "").replace(")", "").split(" - ")
    a = np.array([float(v) for v in first[0].split(", ")])
    b = np.array([float(v) for v in first[1].split(", ")])
    assert np.allclose(a, m[0], rtol=1e-5) and np.abs(a - b).max() < 5e-3


@pytest.mark.gpu
def test_cli_reports_loader_errors(tmp_path):
    (tmp_path / "bad.txt").write_text("x,y,z\n1,2,3\n")
    r = run(str(tmp_path / "bad.txt"), str(tmp_path / "bad.txt"))
    assert r.returncode != 0
    assert "could not find one of the coordinates" in r.stderr
