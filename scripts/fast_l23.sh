run() { python3 bench.py --workload silesia-mix --level $1 --lz $2 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['stage_ms_per_step']
print('L$1 $2  %7.2f GiB/s  %8.1f ms/step  stages %s' % (d['value'], d['ms_per_step'], s))"; }
for lv in 3 2 1; do run $lv fast; run $lv auto; done
