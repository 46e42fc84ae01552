run() { python3 bench.py --workload $1 --level $2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['stage_ms_per_step']
print('$1 L$2 $3  %7.2f GiB/s  %8.1f ms/step  checked %s  stages %s' % (d['value'], d['ms_per_step'], d['config']['chunks_checked_against_reference_hashes'], s))"; }
for lv in 1 2 3; do
ZGPU_HAND_ON=0 run silesia-mix $lv loop-only
run silesia-mix $lv hand-on
done
ZGPU_HAND_ON=0 run log-text 1 loop-only
run log-text 1 hand-on
