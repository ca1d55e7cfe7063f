#!/bin/bash
# same-box A/B of two builds of the library: kernels (tools/w24_ab.py shapes) and the bench step, alternating.
# usage: tools/gpu_ab_lib.sh OLD.so [rounds]      (NEW = centerpose_amd/libcenterpose_hip.so)
OLD=$1; R=${2:-3}
NEW=centerpose_amd/libcenterpose_hip.so
bench() { python tools/with_lib.py $1 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs --no-profile 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0]); print('  bench $2 %.1f img/s %.3f ms (median %.3f)' % (l['value'], l['ms_per_step'], l['step_ms']['median']))"; }
for i in $(seq $R); do
  echo "--- old"; python tools/with_lib.py $OLD tools/w24_ab.py CP_NONE 0
  echo "--- new"; python tools/with_lib.py $NEW tools/w24_ab.py CP_NONE 0
done
for i in $(seq $R); do bench $OLD old; bench $NEW new; done
