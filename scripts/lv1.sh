for v in fw1 fw2 fw3; do
ZAMD_GPU_LIB=$PWD/build/variants/$v.so python3 bench.py --workload silesia-mix --level 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['stage_ms_per_step']
print('$v L1  %7.2f GiB/s  %8.1f ms/step  stages %s' % (d['value'], d['ms_per_step'], s))"
done
