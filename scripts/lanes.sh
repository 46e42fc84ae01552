for lv in 1; do for ln in 12 16 20 24; do
ZGPU_SERIAL_LANES=$ln python3 bench.py --workload silesia-mix --level $lv --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('L$lv lanes $ln  %7.2f GiB/s  %8.1f ms/step  lz %.1f' % (d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_step']['lz_serial']))"
done; done
