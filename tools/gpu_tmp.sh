#!/bin/bash
for age in 4 8 12 16 20; do for cfg in "dla_34 16" "hrnet 8" "res_50 8"; do set -- $cfg
  CP_BUFFER_MIN_AGE=$age timeout 200 python bench.py --arch $1 --batch $2 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys; l=json.loads(sys.stdin.read()); print('age $age $1 B=$2', l['value'], 'img/s', l['ms_per_step'], 'ms; act MB', l['activation_mb'])"
done; done
