#!/bin/bash
run() { env $1 python bench.py --arch $2 --batch $3 --steps 60 --warmup 12 --no-cpu-baseline --no-profile --no-other-configs 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith(chr(123))][0]); print('  %-8s %-24s %.1f img/s %.3f ms' % ('$2', '$1', l['value'], l['ms_per_step']))"; }
for rep in 1 2; do for k in 0 3 8 20 45; do run CP_PIPE_STAGGER=$k dla_34 16; done; done
for k in 0 10 40; do run CP_PIPE_STAGGER=$k res_50 8; run CP_PIPE_STAGGER=$k hrnet 8; done
