#!/bin/bash
# Other configurations at HEAD (same kernels): throughput of the BASELINE configs' per-GPU shapes and batch-1 latency.
# usage: tools/gpu_configs.sh <tag>
TAG=${1:-cfg}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for spec in "res_50 8" "res_50 16" "hrnet 8" "hrnet 16" "mobilenetv3 16" "shufflenetV2 16" "dla_34 1" "dla_34 4" "res_50 1" "hrnet 1"; do
  set -- $spec
  timeout 300 python bench.py --arch $1 --batch $2 --steps 50 --warmup 10 --no-cpu-baseline --no-profile > $OUT/bench_$1_$2.json 2> $OUT/bench_$1_$2.err
  python - <<PY
import json
l = json.load(open("$OUT/bench_$1_$2.json"))
print("%-14s B=%-3s %8.1f img/s  %7.3f ms/step  capture %s" % ("$1", "$2", l["value"], l["ms_per_step"], l["graph_capture"]))
PY
done | tee $OUT/configs.txt
