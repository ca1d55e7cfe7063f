#!/bin/bash
# round 5, GPU session 5: steps in flight (engine.EnginePipeline + bench --in-flight): full GPU suite, bench line at depth 2, depth sweep
OUT=gpurun_out/r5c5; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest.log | tail -8
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 400 $OUT/bench.err
python - <<PY
import json
l = json.load(open("$OUT/bench.json"))
print(l["value"], "img/s", l["ms_per_step"], "ms; one in flight", l["one_step_in_flight"], "sustained", l["sustained"]["images_per_sec"], l["step_ms"])
for k, v in l["other_configs"].items():
    print(k, v.get("images_per_sec"), v.get("ms_per_step"), v.get("one_step_in_flight"), v.get("error"))
PY
for d in 1 3 4; do
  timeout 600 python bench.py --in-flight $d --no-cpu-baseline --no-profile --no-other-configs 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0]); print('in flight $d: %.1f img/s %.3f ms' % (l['value'], l['ms_per_step']))"
done
