#!/bin/bash
OUT=gpurun_out/r2m; mkdir -p $OUT
tools/gpu_profile.sh r2m pmc > $OUT/profile.log 2>&1
timeout 200 python tools/layer_profile.py res_50 16 > $OUT/layers_res50.txt 2>&1
timeout 200 python tools/layer_profile.py hrnet 16 > $OUT/layers_hrnet.txt 2>&1
cp $OUT/pmc_traffic.json profiles/r2_pmc_traffic.json
timeout 300 python bench.py > $OUT/bench_default.json 2>/dev/null
for cfg in "res_50 8" "res_50 16" "hrnet 8" "hrnet 16" "mobilenetv3 16" "shufflenetV2 16"; do set -- $cfg
  timeout 200 python bench.py --arch $1 --batch $2 --no-cpu-baseline 2>/dev/null > $OUT/bench_$1_$2.json
  python -c "
import json; l=json.load(open('$OUT/bench_$1_$2.json')); r=l['roofline']; print('$1 B=$2', l['value'], 'img/s', l['ms_per_step'], 'ms; dom', r['kernel'][:40], r['frac'], 'all-mfma exe frac', r['all_mfma_kernels']['executed_frac'])"
done
python -c "
import json; l=json.load(open('$OUT/bench_default.json')); print(l['value'], l['ms_per_step'], l['roofline']['kernel'], l['roofline']['frac'], l['roofline']['traffic'], l['cpu_baseline']['value'])"
