#!/bin/bash
# GPU box: cumulative time of sort3_kernel's passes (builds with -DZGPU_S3_STOP=N leave the kernel after pass N; scripts/build_variant.sh s3stopN -DZGPU_S3_STOP=N)
for n in 0 1 2 3; do
  ZAMD_GPU_LIB=$PWD/build/variants/s3stop$n.so python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('sort3 up to stop $n: %.2f ms per 4 GiB' % d['roofline']['stage_ms_per_step']['chain'])"
done
python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('sort3 whole: %.2f ms per 4 GiB' % d['roofline']['stage_ms_per_step']['chain'])"
