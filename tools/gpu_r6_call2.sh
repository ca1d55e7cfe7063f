#!/bin/bash
# round 6, call 2: whole GPU suite (new: 8-rank rehearsal, hrnet 2-rank, DCN argument space, dw_deconv2 A/B) + the driver-style bench line
OUT=gpurun_out/r6c2; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<PY
import json
l = json.load(open("$OUT/bench.json"))
print(l["value"], "img/s", l["ms_per_step"], "ms; one:", l["one_step_in_flight"]["images_per_sec"], "sus", l["sustained"]["images_per_sec"])
print({k: (v.get("traffic_ratio"), v["share"], v["frac"]) for k, v in l["roofline"]["templates"].items()})
for k, v in l.get("other_configs", {}).items():
    print("   other", k, {q: v.get(q) for q in ("images_per_sec", "ms_per_step", "one_step_in_flight", "all_mfma_executed_frac", "error", "wall_ms_per_image_median", "fps", "stage_ms_median")})
c = l["cpu_baseline"]
print("cpu", c["value"], c["cores"], c.get("res_50_best"), c.get("res_50_multiprocess"))
PY
