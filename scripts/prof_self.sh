#!/bin/bash
# PMC passes over scripts/tile_stats.py (8M misaligned x2 launches, then converged x2 launches)
set -u
out=gpurun_out/prof_self
mkdir -p $out
export TMPDIR=/tmp
run="python3 scripts/tile_stats.py 8000000 1.0 256"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM -d $out/sq1 -o pmc --output-format csv -- $run > $out/sq1.log 2>&1 || exit 4
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA -d $out/sq2 -o pmc --output-format csv -- $run > $out/sq2.log 2>&1 || exit 5
timeout -s KILL 200 rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU -d $out/sq3 -o pmc --output-format csv -- $run > $out/sq3.log 2>&1 || exit 6
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum -d $out/l2 -o pmc --output-format csv -- $run > $out/l2.log 2>&1 || exit 7
