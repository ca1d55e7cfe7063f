#!/bin/bash
# same-box A/B of an environment switch on the bench step, alternating, 3 rounds.  usage: tools/gpu_ab_env.sh VAR v1 v2 [v3 ...]
VAR=$1; shift
run() { env $VAR=$1 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs --no-profile 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0]); print('  $VAR=$1 %.1f img/s %.3f ms (median %.3f)' % (l['value'], l['ms_per_step'], l['step_ms']['median']))"; }
for i in 1 2 3; do for v in "$@"; do run $v; done; done
