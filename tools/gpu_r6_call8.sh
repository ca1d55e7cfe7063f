#!/bin/bash
OUT=gpurun_out/r6c8; mkdir -p $OUT
export TMPDIR=/tmp
for a in "dla_34 16" "res_50 8" "hrnet 8"; do timeout 600 python tools/cplan_pipeline_bench.py $a 2 60 2>/dev/null | grep "img/s"; done | tee $OUT/cplan_pipeline.txt
