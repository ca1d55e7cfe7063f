#!/bin/bash
OUT=gpurun_out/tmp; mkdir -p $OUT
timeout 200 python bench.py --arch hrnet --batch 8 --no-cpu-baseline > $OUT/h8.json 2> $OUT/t.err
python - <<'PY'
import json
l=json.load(open("gpurun_out/tmp/h8.json"))
r=l["roofline"]
print(l["value"], l["ms_per_step"], "sum", r["all_kernels_ms_per_step"])
for k,v in list(r["kernels"].items())[:10]: print("   %-52s %3d %7.3f ms %5.1f%% alg %6.1f exe %6.1f TF %.2f TB/s" % (k[:52], v["launches"], v["ms_per_step"], 100*v["share"], v["algorithmic_tflops"], v["executed_tflops"], v["compulsory_tbps"]))
PY
timeout 200 python tools/layer_profile.py hrnet 8 > $OUT/layers_h8.txt 2>&1
sort -k4 -n -r $OUT/layers_h8.txt | awk '{print $1, $2, $3, $5, $7, $9, $10, $12}' | head -30
