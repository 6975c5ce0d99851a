#!/usr/bin/env python3
"""HBM traffic of the NN kernel from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE),
collected and corrected as MI355X_MICROARCH.md (HBM section) prescribes:
  * counters are reported in KiB -> x1024;
  * on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a coalesced streaming read; the
    factor is CALIBRATED in the same run on sum_points_kernel, which reads exactly 12*n bytes with
    the NN kernel's own access shape (one dword per lane, 256 B per wave-instruction);
  * WRITE_SIZE is taken as is.
With --sq-csv (the SQ pass of the same command) the VALU wave-instructions per launch are added:
the result is profiles/nn_counters_<workload>.json, which bench.py reads for its roofline block
when workload, steps, capacity, angle and query count match the command it is running.
Usage: collect_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> \
           [--sq-csv <sq_counter_collection.csv>] --points N --workload 64M --steps 20 --warmup 3 \
           --tag r02a --out profiles/nn_counters_64M.json
"""
import argparse
import csv
import json

ap = argparse.ArgumentParser()
ap.add_argument("fetch_csv")
ap.add_argument("write_csv")
ap.add_argument("--points", type=int, required=True, help="points in the shard (sum_points reads 12 B each)")
ap.add_argument("--workload", default="64M")
ap.add_argument("--out", default=None)
ap.add_argument("--steps", type=int, default=0, help="average over the last K launches (bench.py timed region)")
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--sq-csv", default=None, help="pass with SQ_INSTS_VALU (and friends) of the same command")
ap.add_argument("--tag", default="?")
ap.add_argument("--capacity", type=int, default=256)
ap.add_argument("--angle", type=float, default=None, help="default: bench.py's (0.25 above 8M points, else 1.0)")
a = ap.parse_args()


def load(path, counter, scale=1024.0):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            out.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]) * scale)
    return out


f, w = load(a.fetch_csv, "FETCH_SIZE"), load(a.write_csv, "WRITE_SIZE")
cal = [k for k in f if "sum_points_kernel" in k][0]
factor = 12.0 * a.points / (sum(f[cal]) / len(f[cal]))
nn = [k for k in f if "nn_tile_kernel" in k or "nn_cov_kernel" in k][0]
fetch = [v * factor for v in f[nn]]
write = w[nn]
if a.steps:
    fetch, write = fetch[-a.steps:], write[-a.steps:]
res = {
    # the key bench.py matches its command against (the warm-up count is NOT part of it: the shard
    # is reset after the warm-up, the timed launches are the same)
    "workload": a.workload, "n_gpus": 1, "steps": a.steps, "capacity": a.capacity,
    "angle": a.angle if a.angle is not None else (0.25 if a.points > 8_000_000 else 1.0),
    "queries": a.points, "warmup_of_the_profiled_run": a.warmup, "tag": a.tag,
    "kernel": nn,
    "fetch_size_calibration": {"kernel": cal, "known_bytes": 12 * a.points,
                               "reported_bytes": sum(f[cal]) / len(f[cal]), "factor": factor},
    "launches": len(fetch),
    "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
    "hbm_bytes_per_launch": (sum(fetch) + sum(write)) / len(fetch),
}
if a.sq_csv:
    for name in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM",
                 "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"):
        v = load(a.sq_csv, name, 1.0).get(nn)
        if v:
            v = v[-a.steps:] if a.steps else v
            key = "valu_insts_per_launch" if name == "SQ_INSTS_VALU" else name.lower() + "_per_launch"
            res[key] = sum(v) / len(v)
print(json.dumps(res, indent=1))
if a.out:
    json.dump(res, open(a.out, "w"), indent=1)
