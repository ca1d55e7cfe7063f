#!/bin/bash
# round 5, GPU session 3: pipelined split-bf16 loop (micro), HRNet grouped fuse layers (tests + per-launch table + A/B vs CP_FUSE_GROUP=0),
# parity suite with CP_SPLIT_BF16=1
OUT=gpurun_out/r5c3; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 tools/micro/bf16x3_loop > $OUT/bf16x3_loop.txt 2>&1; echo "micro rc=$?"; cat $OUT/bf16x3_loop.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
grep -E "passed|failed|error|Error" $OUT/pytest.log | tail -8
for g in 1 0; do
  CP_FUSE_GROUP=$g timeout 600 python bench.py --arch hrnet --batch 8 --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs > $OUT/bench_hrnet_group$g.json 2> $OUT/bench_hrnet_group$g.err
  python - <<PY
import json
l = json.load(open("$OUT/bench_hrnet_group$g.json"))
print("hrnet B=8 CP_FUSE_GROUP=$g:", l["value"], "img/s", l["ms_per_step"], "ms; in sequence", l["roofline"]["all_kernels_ms_per_step"], "ms;", {k: (v["launches"], v["ms_per_step"]) for k, v in l["roofline"]["templates"].items()})
PY
done
timeout 300 python tools/layer_profile.py hrnet 8 > $OUT/layers_hrnet.txt 2>&1; head -3 $OUT/layers_hrnet.txt | tail -2
CP_SPLIT_BF16=1 timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_split_bf16.log 2>&1; echo "pytest (CP_SPLIT_BF16=1) rc=$?" | tee -a $OUT/pytest_split_bf16.log
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_split_bf16.log | tail -12
