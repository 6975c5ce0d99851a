"""cli/icp_bench: the reference's benchmark harness (bench/bench/benchmark.cc) over the C ABI.

CPU part: the sweep it registers is the reference's (capacity 8..1024 x five file pairs,
CustomArguments benchmark.cc:21-26), names and labels are Google Benchmark's as in the reference's
logs (analysis/logs/v18.csv), and without a GPU every case reports an error instead of a time.
GPU part: the timed call converges in the iteration counts the CPU path takes."""
import csv
import io
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT, has_gpu

BENCH = os.path.join(ROOT, "cli", "icp_bench")
DATA = os.path.join(GOLDEN, "data")
COLUMNS = ["name", "iterations", "real_time", "cpu_time", "time_unit", "bytes_per_second",
           "items_per_second", "label", "error_occurred", "error_message"]  # v18.csv:1
LABELS = ["line_translated_2_3_4 -> line_ref", "cow_tr1 -> cow_ref", "cow_tr2 -> cow_ref",
          "horse_tr1 -> horse_ref", "horse_tr2 -> horse_ref"]

pytestmark = pytest.mark.skipif(not os.path.exists(BENCH), reason="cli/icp_bench not built")


def run(*args):
    r = subprocess.run([BENCH, "--npy", "--data", DATA, *args], capture_output=True, text=True,
                       timeout=300)
    return r, list(csv.DictReader(io.StringIO(r.stdout)))


def test_default_sweep_is_the_references():
    if has_gpu():
        pytest.skip("a GPU is present: the full sweep is the GPU test's business")
    r, rows = run()
    assert r.returncode == 0
    assert list(rows[0].keys())[:len(COLUMNS)] == COLUMNS
    want = [f"BM_GPU/{c}/{i}/real_time" for c in (8, 16, 32, 64, 128, 256, 512, 1024)
            for i in range(5)]
    assert [row["name"] for row in rows] == want
    assert [row["label"] for row in rows[:5]] == LABELS
    # no GPU here: nothing is timed, nothing falls back to the CPU
    assert all(row["error_occurred"] == "true" and "HIP device" in row["error_message"]
               for row in rows)


def test_arguments():
    r = subprocess.run([BENCH, "--device", "cpu"], capture_output=True, text=True)
    assert r.returncode == 1 and "libcpu" in r.stderr
    r = subprocess.run([BENCH, "--bogus"], capture_output=True, text=True)
    assert r.returncode == 1 and "Invalid argument: --bogus" in r.stderr
    r = subprocess.run([BENCH, "--data", "/nonexistent", "--files", "0"], capture_output=True,
                       text=True)
    assert r.returncode == 1 and "ould not open" in r.stderr  # read_csv's message


@pytest.mark.gpu
def test_sweep_runs_and_converges_like_the_cpu_path():
    r, rows = run("--capacities", "32,256", "--min-time", "0.05", "--synthetic", "100000:1")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(rows) == 12
    assert not any(row["error_occurred"] for row in rows)
    its = {row["name"]: int(row["icp_iterations"]) for row in rows}
    # tests/golden/icp_expected.json: horse_tr1 16 (cap 32), cow_tr1 1, cow_tr2 11, line 1
    assert its["BM_GPU/32/3/real_time"] == 16 and its["BM_GPU/256/1/real_time"] == 1
    assert its["BM_GPU/256/2/real_time"] == 11 and its["BM_GPU/32/0/real_time"] == 1
    assert rows[5]["label"] == "synthetic_100000_rot1 -> synthetic_100000"
    assert all(float(row["real_time"]) > 0 and row["time_unit"] == "ms" for row in rows)
    # reuse of one context over all cases gives the same iteration counts
    r2, rows2 = run("--capacities", "32,256", "--min-time", "0.05", "--synthetic", "100000:1",
                    "--reuse-context")
    assert r2.returncode == 0
    assert [row["icp_iterations"] for row in rows2] == [row["icp_iterations"] for row in rows]
