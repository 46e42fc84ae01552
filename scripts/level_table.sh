#!/bin/bash
# GPU box: every level on both synthetic workloads, HBM-resident, one bench line each -> gpurun_out/level_table.txt
OUT=gpurun_out/level_table.txt; : > $OUT
for wl in silesia-mix log-text; do
  for lv in 1 2 3 4 5 6 7 8 9; do
    st=3; [ $lv -ge 8 ] && st=2
    python3 bench.py --workload $wl --level $lv --steps $st --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['stage_ms_per_step']
print('%-11s L%d  %7.2f GiB/s  %8.1f ms/step  ratio %.3f  stages %s' % ('$wl', $lv, d['value'], d['ms_per_step'], d['config']['compression_ratio'], s))" >> $OUT
  done
  python3 bench.py --workload $wl --op inflate --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-11s inflate of the L6 stream  %7.2f GiB/s of output  %8.1f ms/step' % ('$wl', d['value'], d['ms_per_step']))" >> $OUT
done
cat $OUT
