for lv in 1 2 3; do
python3 bench.py --workload silesia-mix --level $lv --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['stage_ms_per_step']
print('L$lv  %7.2f GiB/s  %8.1f ms/step  ratio %.3f  stages %s' % (d['value'], d['ms_per_step'], d['config']['compression_ratio'], s))"
done
python3 bench.py --workload log-text --level 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['stage_ms_per_step']
print('log L1  %7.2f GiB/s  %8.1f ms/step  ratio %.3f  stages %s' % (d['value'], d['ms_per_step'], d['config']['compression_ratio'], s))"
python3 scripts/size_table.py
