#!/bin/bash
OUT=gpurun_out/tmp; mkdir -p $OUT
for thr in 4096 2048 1024 512 256 4096; do
for cfg in "dla_34 16" "res_50 8" "hrnet 8"; do
  set -- $cfg
  CP_IGEMM_T128=$thr timeout 200 python bench.py --arch $1 --batch $2 --no-cpu-baseline > $OUT/t_$1_$thr.json 2> $OUT/t.err
  python - <<PY
import json
l=json.load(open("$OUT/t_$1_$thr.json"))
ks=l["roofline"]["kernels"]
ig=[(k,v["ms_per_step"],v["launches"]) for k,v in ks.items() if k.startswith("igemm_conv_kernel<64, 64") or k.startswith("igemm_conv_kernel<128, 64")]
print("thr $thr $1 B=$2", l["value"], "img/s", l["ms_per_step"], "ms;", ig)
PY
done; done
