#!/bin/bash
# GPU box: A/B the engine builds under build/variants/ (ZAMD_GPU_LIB override) on the same 1 GiB workload
for f in zlib_amd/libzamd_gpu.so build/variants/*.so; do
  r=$(ZAMD_GPU_LIB=$PWD/$f timeout -k 10 400 python bench.py --gib 1 --steps 2 --warmup 1 --no-cpu-baseline $BENCH_EXTRA 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*\|"stage_ms_per_step": {[^}]*}')
  echo "$f $r" | tr '\n' ' '; echo
done
