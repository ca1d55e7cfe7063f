#!/bin/bash
# pointwise kernel in the networks: whole GPU suite, then res_50 B=8 / hrnet B=8 with and without it (CP_POINTWISE=0), one box
OUT=gpurun_out/r6c10; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
for rep in 1 2; do for sw in 1 0; do for cfg in "res_50 8" "hrnet 8"; do set -- $cfg
  CP_POINTWISE=$sw timeout 600 python bench.py --arch $1 --batch $2 --steps 60 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0]); r=l['roofline']
print('CP_POINTWISE=$sw $1 B=$2: %.1f img/s (%.3f ms) | one step %.1f | in-sequence %.3f ms | all-MFMA %.4f | pw: %s' % (l['value'], l['ms_per_step'], l['one_step_in_flight']['images_per_sec'], r['all_kernels_ms_per_step'], r['all_mfma_kernels']['executed_frac'], {k: v['ms_per_step'] for k, v in r['kernels'].items() if k.startswith('pw_')}))"
done; done; done | tee $OUT/pointwise_ab.txt
