#!/bin/bash
# round 5, GPU session 4: F(2x4) patch-plane padding A/B (old vs new library), new unit tests, bench line with the split-bf16 launch rule
OUT=gpurun_out/r5c4; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_hip.py -m gpu -q --timeout 600 -k "group or wino or split_bf16" > $OUT/pytest_subset.log 2>&1; echo "pytest subset rc=$?"; tail -3 $OUT/pytest_subset.log
timeout 1200 bash tools/gpu_ab_lib.sh centerpose_amd/csrc/build/ab/libold.so 2 > $OUT/w24_plane_pad_ab.txt 2>&1; cat $OUT/w24_plane_pad_ab.txt | grep -v amdgpu.ids
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
l = json.load(open("$OUT/bench.json"))
print(l["value"], "img/s", l["ms_per_step"], "ms", l.get("sustained", {}).get("images_per_sec"))
for k, v in l["other_configs"].items():
    print(k, v.get("images_per_sec"), v.get("ms_per_step"), v.get("error"), {t: (tv["launches"], tv["ms_per_step"], tv["frac"]) for t, tv in v.get("templates", {}).items()})
PY
