#!/bin/bash
# A/B of the DCNv2 kernel variants on the bench workload.  usage: tools/gpu_dcn_ab.sh <tag>
TAG=${1:-ab}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 300 python -m pytest tests/test_conv_hip.py -m gpu -x -q -k "dcn" > $OUT/pytest_dcn.log 2>&1; tail -3 $OUT/pytest_dcn.log
for t in ${TILES:-0 64064 64128 128064}; do
  CP_DCN_TILE=$t timeout 200 python bench.py --no-cpu-baseline > $OUT/bench_dcn$t.json 2> $OUT/bench_dcn$t.err
  python - <<PY
import json
l=json.load(open("$OUT/bench_dcn$t.json"))
print("CP_DCN_TILE=$t", l["value"], "img/s", l["ms_per_step"], "ms; dom", l["roofline"]["kernel"], l["roofline"]["frac"])
for k,v in l["roofline"]["kernels"].items(): print("   %-52s %7.3f ms %5.1f%% alg %6.1f exe %6.1f TF" % (k[:52], v["ms_per_step"], 100*v["share"], v["algorithmic_tflops"], v["executed_tflops"]))
print("   decode", l["decode"])
PY
done
