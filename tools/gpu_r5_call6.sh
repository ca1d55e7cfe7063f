#!/bin/bash
# round 5, GPU session 6: two steps in ONE captured graph (engine.EnginePipeline): tests + bench
OUT=gpurun_out/r5c6; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_engine_hip.py tests/test_dist_gpu.py -m gpu -q --timeout 900 -k "in_flight or bench" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest.log | tail -8
timeout 900 python -X faulthandler bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 600 $OUT/bench.err
python - <<PY
import json
l = json.load(open("$OUT/bench.json"))
print(l["value"], "img/s", l["ms_per_step"], "ms; one in flight", l["one_step_in_flight"], "sustained", l["sustained"]["images_per_sec"], l["step_ms"], l["pipeline_graph_capture"])
for k, v in l["other_configs"].items():
    print(k, v.get("images_per_sec"), v.get("ms_per_step"), v.get("one_step_in_flight"), v.get("pipeline_graph_capture"), v.get("error"))
PY
