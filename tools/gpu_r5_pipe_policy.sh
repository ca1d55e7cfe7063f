#!/bin/bash
run() { env $1 python bench.py --arch $2 --batch $3 --steps 60 --warmup 12 --no-cpu-baseline --no-profile --no-other-configs 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith(chr(123))][0]); print('  %-8s %-28s %.1f img/s %.3f ms' % ('$2', '$1', l['value'], l['ms_per_step']))"; }
for rep in 1 2; do for a in "dla_34 16" "res_50 8" "hrnet 8"; do set -- $a; run CP_PIPE_POLICY=sched $1 $2; run CP_PIPE_POLICY=instance $1 $2; done; done
