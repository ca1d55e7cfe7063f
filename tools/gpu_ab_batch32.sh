#!/bin/bash
# is ONE plan at twice the batch faster than two B=16 instances in one graph?  (what a concatenating process_many would run)
OUT=gpurun_out/r6c6; mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "16 2" "32 1" "16 1" "32 2"; do set -- $cfg
  timeout 600 python bench.py --batch $1 --in-flight $2 --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-profile 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0]); print('B=$1 in-flight $2:', l['value'], 'img/s', l['ms_per_step'], 'ms/step', l['step_ms'])"
done | tee $OUT/batch32.txt
for cfg in "8 2" "16 1"; do set -- $cfg
  timeout 600 python bench.py --arch hrnet --batch $1 --in-flight $2 --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-profile 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0]); print('hrnet B=$1 in-flight $2:', l['value'], 'img/s', l['ms_per_step'], 'ms/step')"
done | tee -a $OUT/batch32.txt
