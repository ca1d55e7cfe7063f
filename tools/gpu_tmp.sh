#!/bin/bash
OUT=gpurun_out/tmp; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_hip.py tests/test_engine_hip.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
for cfg in "dla_34 16" "res_50 8" "hrnet 8"; do
  set -- $cfg
  timeout 200 python bench.py --arch $1 --batch $2 --no-cpu-baseline > $OUT/t.json 2> $OUT/t.err || tail -3 $OUT/t.err
  python - <<PY
import json
l=json.load(open("$OUT/t.json"))
ks=l["roofline"]["kernels"]
print("$1 B=$2", l["value"], "img/s", l["ms_per_step"], "ms;", [(k[:24],v["ms_per_step"],v["executed_tflops"]) for k,v in ks.items() if k.startswith(("dcn","igemm_conv_kernel<64, 64"))])
PY
done; done
