#!/bin/bash
# round 6, call 1: the steps-in-flight entry points (process_stream) -- new parity test at the timed size, the bench line through the
# product API, then the whole GPU suite.
OUT=gpurun_out/r6c1; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_hip.py -m gpu -x -q -s -k "two_steps_in_flight or steps_in_flight_same_bits" > $OUT/pytest_inflight.log 2>&1; echo "inflight rc=$?"; tail -5 $OUT/pytest_inflight.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<PY
import json
l = json.load(open("$OUT/bench.json"))
print(l["value"], "img/s", l["ms_per_step"], "ms; one:", l["one_step_in_flight"], "sus", l["sustained"]["images_per_sec"])
print({k: (v.get("traffic_ratio"), v["share"], v["frac"]) for k, v in l["roofline"]["templates"].items()})
for k, v in l.get("other_configs", {}).items():
    print("   other", k, {q: v.get(q) for q in ("images_per_sec", "ms_per_step", "one_step_in_flight", "all_mfma_executed_frac", "error")})
PY
timeout 1800 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log
