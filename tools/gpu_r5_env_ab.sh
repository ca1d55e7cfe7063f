#!/bin/bash
# round 5: with two steps in flight the holes a launch leaves are filled by the other step -- do the rules that traded kernel efficiency for
# fill (split-K / split-C launches, 64-wide DCN tiles) still pay?  same box, alternating.   usage: tools/gpu_r5_env_ab.sh "arch batch" ...
run() { env $2 python bench.py --arch $ARCH --batch $B --steps 60 --warmup 12 --no-cpu-baseline --no-profile --no-other-configs 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith(chr(123))][0]); print('  %-8s %-44s %.1f img/s %.3f ms' % ('$ARCH', '$1', l['value'], l['ms_per_step']))"; }
for cfg in "$@"; do
  set -- $cfg; ARCH=$1; B=$2
  for rep in 1 2; do
    run "default" "CP_NOP=1"
    run "CP_DCN_TILE=6064128" "CP_DCN_TILE=6064128"
    run "no split launches" "CP_DCN_SPLITK=0 CP_WINO_SPLITC=0 CP_CONV_SPLITK=0"
    run "CP_SCHED_PREFER=0" "CP_SCHED_PREFER=0"
  done
done
