#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel calls / total / average, plus the average
over the LAST `--steps` dispatches of the NN kernel (= bench.py's timed region; the earlier ones
are its warm-up iterations).  Usage: summarize_profile.py <kernel_trace.csv> [--steps K]"""
import argparse
import csv
import collections
import json

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--bench-json", default=None)
a = ap.parse_args()

rows = list(csv.DictReader(open(a.trace)))
per = collections.defaultdict(list)
meta = {}
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    per[name].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    meta[name] = {k: r[k] for k in ("LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count",
                                    "SGPR_Count", "Workgroup_Size_X", "Grid_Size_X")}
total = sum(d for v in per.values() for _, d in v)
print(f"{'kernel':60s} {'calls':>6s} {'total_ms':>12s} {'avg_ms':>10s} {'share':>7s}  resources")
for name, v in sorted(per.items(), key=lambda kv: -sum(d for _, d in kv[1])):
    t = sum(d for _, d in v)
    print(f"{name[:60]:60s} {len(v):6d} {t/1e6:12.3f} {t/len(v)/1e6:10.4f} {100*t/total:6.2f}%  {meta[name]}")
nn = [k for k in per if "nn_tile_kernel" in k or "nn_cov_kernel" in k]
for k in nn:
    v = sorted(per[k])[-a.steps:]
    avg = sum(d for _, d in v) / len(v) / 1e6
    print(f"\n{k}: average over the last {len(v)} dispatches (timed region) = {avg:.4f} ms")
if a.bench_json:
    j = json.loads(open(a.bench_json).read().strip().splitlines()[-1])
    print(f"bench.py (same command) roofline.avg_launch_ms = {j['roofline']['avg_launch_ms']} ms, "
          f"{j['roofline'].get('bound')}: achieved {j['roofline']['achieved']} {j['roofline'].get('unit')}, frac {j['roofline']['frac']}")
