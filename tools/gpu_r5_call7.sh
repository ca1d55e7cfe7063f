#!/bin/bash
OUT=gpurun_out/r5c7; mkdir -p $OUT; export TMPDIR=/tmp
python tools/bf16x3_ab.py 2>&1 | grep -v amdgpu.ids > $OUT/bf16x3_ab_v2.txt; cut -c1-34,200- $OUT/bf16x3_ab_v2.txt
CP_SPLIT_BF16=1 timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -3
CP_SPLIT_BF16=1 CP_SPLIT_BF16_MINBLOCKS=0 timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
l = json.load(open("$OUT/bench.json"))
print(l["value"], "img/s; one", l["one_step_in_flight"]["images_per_sec"])
for k, v in l["other_configs"].items():
    print(k, v.get("images_per_sec"), v.get("one_step_in_flight"), {t: (tv["launches"], tv["ms_per_step"], tv["frac"]) for t, tv in v.get("templates", {}).items() if "igemm" in t})
PY
