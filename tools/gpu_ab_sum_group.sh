#!/bin/bash
# grouped up-sampling sums of HRNet: kernel test, engine tests of hrnet, then the same-box A/B (CP_SUM_GROUP=0 / 1)
OUT=gpurun_out/r6c11; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "sum_up or hrnet or c_plan or plan" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
for rep in 1 2; do for sw in 1 0; do
  CP_SUM_GROUP=$sw timeout 600 python bench.py --arch hrnet --batch 8 --steps 60 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0]); r=l['roofline']
print('CP_SUM_GROUP=$sw hrnet B=8: %.1f img/s (%.3f ms) | one step %.1f | in-sequence %.3f ms | launches %d | sums: %s' % (l['value'], l['ms_per_step'], l['one_step_in_flight']['images_per_sec'], r['all_kernels_ms_per_step'], sum(v['launches'] for v in r['kernels'].values()), {k: (v['launches'], v['ms_per_step']) for k, v in r['kernels'].items() if k.startswith('sum_up')}))"
done; done | tee $OUT/sum_group_ab.txt
