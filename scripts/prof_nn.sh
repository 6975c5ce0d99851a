#!/bin/bash
# PMC passes over scripts/nn64m.py (4 misaligned + 4 aligned NN launches): instruction mix, waits,
# matrix-pipe and LDS activity of nn_tile_kernel.  usage: scripts/prof_nn.sh <tag> [n] [cap] [angle]
set -u
tag=${1:-x}; n=${2:-16000000}; cap=${3:-256}; ang=${4:-0.25}
out=gpurun_out/prof_nn_$tag
mkdir -p $out
export TMPDIR=/tmp
run="python3 scripts/nn64m.py $n $cap $ang"
timeout -k 10 200 $run > $out/plain.log 2>&1 || exit 3
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM -d $out/sq1 -o pmc --output-format csv -- $run > $out/sq1.log 2>&1 || exit 4
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA -d $out/sq2 -o pmc --output-format csv -- $run > $out/sq2.log 2>&1 || exit 5
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC -d $out/sq3 -o pmc --output-format csv -- $run > $out/sq3.log 2>&1 || exit 6
python3 - "$out" <<'PY'
import csv, sys, collections
out = sys.argv[1]
for p in ("sq1", "sq2", "sq3"):
    acc = collections.defaultdict(list)
    try:
        for r in csv.DictReader(open(f"{out}/{p}/pmc_counter_collection.csv")):
            if "nn_tile_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    except FileNotFoundError:
        continue
    for k, v in acc.items():
        print(p, k, " ".join(f"{x:.4g}" for x in v))
PY
cat $out/plain.log
