#!/bin/bash
# Row f-4: leaf-capacity sweep of the synthetic workloads (bench.py, CPU baseline off).
# Usage (on the GPU box): bash scripts/capacity_study.sh  -> gpurun_out/capstudy_<workload>_<cap>.json
set -e
mkdir -p gpurun_out
for wl in 1M 64M; do
  for cap in 32 64 128 256 512 1024; do
    timeout -k 10 120 python bench.py --workload $wl --capacity $cap --cpu-sample 0 \
        > gpurun_out/capstudy_${wl}_${cap}.json 2> gpurun_out/capstudy_${wl}_${cap}.err
    echo "$wl $cap done"
  done
done
