#!/bin/bash
# GPU box: kernel-trace stats + PMC counters for one bench configuration.  Usage: scripts/prof_kernels.sh TAG [bench args]
# Writes gpurun_out/prof_TAG/{stats,pmc1,pmc2}/...; copy the summaries worth keeping into profiles/.
set -o pipefail
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $OUT/bench_stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc1 -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $OUT/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM --output-format csv -d $OUT/pmc2 -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $OUT/bench_pmc2.log 2>&1
python3 scripts/prof_summarize.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
