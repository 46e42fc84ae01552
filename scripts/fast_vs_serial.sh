#!/bin/bash
# GPU box: levels 1-3, the head/prev loop (serial) against deflate_fast on the sorted buckets (fast), at 4 GiB and at 256 MiB
for gib in 4 0.25; do for lv in 1 2 3; do for lz in serial fast; do
  python bench.py --level $lv --lz $lz --gib $gib --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('gib $gib level $lv $lz: %.2f GiB/s %.1f ms' % (d['value'], d['ms_per_step']))"
done; done; done
