#!/bin/bash
OUT=gpurun_out/tmp; mkdir -p $OUT
for cfg in "res_50 8" "res_50 16" "hrnet 8" "hrnet 16" "mobilenetv3 16" "shufflenetV2 16" "dla_34 1" "dla_34 4" "res_50 1" "hrnet 1"; do
  set -- $cfg
  timeout 200 python bench.py --arch $1 --batch $2 --no-cpu-baseline --no-profile > $OUT/t.json 2> $OUT/t.err || tail -3 $OUT/t.err
  python - <<PY
import json
l=json.load(open("$OUT/t.json"))
print("$1 B=$2", l["value"], "img/s", l["ms_per_step"], "ms")
PY
done
