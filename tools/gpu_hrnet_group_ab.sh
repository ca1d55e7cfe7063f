run() { env CP_GROUP=$1 python bench.py --arch hrnet --batch 8 --steps 60 --warmup 15 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0]); r=l['roofline']
print('  CP_GROUP=$1 %.1f img/s %.3f ms | in-seq %.3f ms |' % (l['value'], l['ms_per_step'], r['all_kernels_ms_per_step']), {k:(v['launches'], v['ms_per_step']) for k,v in r['templates'].items()})"; }
for i in 1 2; do run 0; run 1; done
