#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (scripts/profile_round.sh) into the tracked profiles/<tag>_* files:
the kernel-trace summary and stats, the traced bench line, the HBM-traffic JSON and, per PMC pass,
the rows of the NN kernel and of the FETCH_SIZE calibration kernel (sum_points_kernel).
Usage: publish_profile.py <tag> [64M|1M]"""
import csv
import os
import shutil
import sys

tag = sys.argv[1]
wl = sys.argv[2] if len(sys.argv) > 2 else "64M"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}_{wl}")
dst = os.path.join(root, "profiles")
shutil.copy(os.path.join(src, "kernel_trace_summary.txt"), os.path.join(dst, f"{tag}_bench{wl}_kernel_trace_summary.txt"))
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, f"{tag}_bench{wl}_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_traced.json"), os.path.join(dst, f"{tag}_bench{wl}_traced.json"))
shutil.copy(os.path.join(src, f"nn_counters_{wl}.json"), os.path.join(dst, f"{tag}_nn_counters_{wl}.json"))
shutil.copy(os.path.join(src, f"nn_counters_{wl}.json"), os.path.join(dst, f"nn_counters_{wl}.json"))
keep = ("sum_points_kernel", "nn_tile_kernel", "nn_cov_kernel")
cols = ["Dispatch_Id", "Kernel", "Grid_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value"]
for p in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2", "pmc_l2"):
    path = os.path.join(src, p, "pmc_counter_collection.csv")
    if not os.path.exists(path):
        continue
    with open(path) as f, open(os.path.join(dst, f"{tag}_{wl}_{p}.csv"), "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(cols)
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0]
            if any(k in name for k in keep):
                w.writerow([r["Dispatch_Id"], name, r["Grid_Size"], r["LDS_Block_Size"], r["VGPR_Count"],
                            r["SGPR_Count"], r["Counter_Name"], r["Counter_Value"]])
print("published", tag)
