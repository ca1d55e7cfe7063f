#!/bin/bash
# same-box A/B of the F(2x4,3x3) layer-selection rule (CP_WINO24_RULE = min cin, min cout, min blocks); usage: tools/gpu_ab_wino24.sh [arch batch]
ARCH=${1:-dla_34}; B=${2:-16}
run() {
  python bench.py --arch $ARCH --batch $B --steps 60 --warmup 15 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0]); r=l['roofline']
t=r['templates']
print('  %7.1f img/s %6.3f ms | ' % (l['value'], l['ms_per_step']) + ' '.join('%s %d x %.3f ms %.3f' % (k.replace('conv3x3_','').replace('_kernel',''), v['launches'], v['ms_per_step'], v['frac']) for k,v in t.items() if 'wino' in k))"
}
echo "CP_WINO24=0"; CP_WINO24=0 run
for rule in 128,64,512 64,64,512 32,16,512 32,16,256 32,16,64; do echo "rule $rule"; CP_WINO24_RULE=$rule run; done
