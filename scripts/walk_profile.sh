#!/bin/bash
# GPU box: where walk_kernel's waves spend their cycles and what they do, per level (builds: scripts/build_variant.sh wtime -DZGPU_WALK_TIME, wstats -DZGPU_WALK_STATS)
OUT=${1:-gpurun_out/walk_profile.txt}; : > $OUT
for lv in 4 6 9; do
  ZAMD_GPU_LIB=$PWD/build/variants/wtime.so timeout -k 10 200 python3 scripts/walk_time.py $lv 0 2>/dev/null >> $OUT
  ZAMD_GPU_LIB=$PWD/build/variants/wstats.so timeout -k 10 200 python3 scripts/walk_stats.py $lv 0 2>/dev/null >> $OUT
done
cat $OUT
