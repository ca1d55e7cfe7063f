run() { python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs --no-profile 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0]); print('  %s %.1f img/s %.3f ms (median %.3f)' % ('$1', l['value'], l['ms_per_step'], l['step_ms']['median']))"; }
for i in 1 2 3; do
  CP_WINO24_RULE=32,16,512 run 512
  CP_WINO24_RULE=32,16,256 run 256
  CP_WINO24_RULE=32,16,128 run 128
done
