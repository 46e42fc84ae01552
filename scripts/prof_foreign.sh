#!/bin/bash
# GPU box: rocprofv3 evidence for the decode of a foreign stream in pieces (scripts/foreign_stream_rate.py MiB): kernel stats, HBM traffic and SQ
# counters in separate passes -> gpurun_out/prof_foreign/{stats,rd,wr,sq}
MIB=${1:-512}
OUT=$PWD/gpurun_out/prof_foreign; mkdir -p $OUT
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/scripts/foreign_stream_rate.py $MIB > $OUT/stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/rd -- python3 $R/scripts/foreign_stream_rate.py $MIB > $OUT/rd.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/wr -- python3 $R/scripts/foreign_stream_rate.py $MIB > $OUT/wr.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/sq -- python3 $R/scripts/foreign_stream_rate.py $MIB > $OUT/sq.log 2>&1 || exit 1
cd $R && python3 scripts/prof_summarize.py $OUT > $OUT/summary.txt 2>&1
grep -v "^[WEI]2026" $OUT/stats.log | tail -2
