#!/bin/bash
# Runs on the GPU box: kernel trace of the default bench + separate PMC passes of the same command.
# Usage: scripts/profile_round.sh <tag> [64M|1M]   -> gpurun_out/prof_<tag>_<workload>/
set -u
tag=${1:-rXX}
wl=${2:-64M}
points=64000000
[ "$wl" = "1M" ] && points=1000000
out=gpurun_out/prof_${tag}_$wl
mkdir -p $out
export TMPDIR=/tmp
# the same command everywhere: 3 warm-up + 20 timed launches, no CPU baseline leg
short="python3 bench.py --workload $wl --cpu-sample 0"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/trace -o trace --output-format csv -- $short > $out/bench_traced.json 2> $out/trace.err || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o pmc --output-format csv -- $short > $out/pmc_fetch.json 2> $out/pmc_fetch.err || exit 2
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o pmc --output-format csv -- $short > $out/pmc_write.json 2> $out/pmc_write.err || exit 3
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM -d $out/pmc_sq1 -o pmc --output-format csv -- $short > $out/pmc_sq1.json 2> $out/pmc_sq1.err || exit 4
timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM -d $out/pmc_sq2 -o pmc --output-format csv -- $short > $out/pmc_sq2.json 2> $out/pmc_sq2.err || exit 5
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE -d $out/pmc_l2 -o pmc --output-format csv -- $short > $out/pmc_l2.json 2> $out/pmc_l2.err || exit 6
python3 scripts/summarize_profile.py $out/trace/trace_kernel_trace.csv --steps 20 --bench-json $out/bench_traced.json > $out/kernel_trace_summary.txt
python3 scripts/collect_traffic.py $out/pmc_fetch/pmc_counter_collection.csv $out/pmc_write/pmc_counter_collection.csv --sq-csv $out/pmc_sq1/pmc_counter_collection.csv --points $points --workload $wl --steps 20 --warmup 3 --tag $tag --out $out/nn_counters_$wl.json > /dev/null
cat $out/kernel_trace_summary.txt | cut -c1-150
