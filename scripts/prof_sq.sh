#!/bin/bash
# GPU box: the SQ counters of one bench configuration alone (two passes).  Usage: scripts/prof_sq.sh TAG [bench args]
set -o pipefail
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_sq -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $OUT/bench_sq.log 2>&1 || exit 1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $OUT/bench_sq2.log 2>&1 || exit 1
python3 scripts/prof_summarize.py $OUT > $OUT/summary.txt 2>&1
