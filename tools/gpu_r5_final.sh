#!/bin/bash
# round 5, final evidence session: the bench line as the driver runs it (with the CPU baseline) + the whole GPU suite
OUT=gpurun_out/r5final; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest.log | tail -6
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 300 $OUT/bench.err
timeout 600 python bench.py --gpus 1 --force-gather --steps 20 --warmup 5 --no-cpu-baseline --no-profile --gather-check > $OUT/bench_force_gather.json 2> $OUT/bench_force_gather.err; echo "force-gather rc=$?"
python - <<PY
import json
l = json.load(open("$OUT/bench.json"))
print(l["value"], "img/s", l["ms_per_step"], "ms; one in flight", l["one_step_in_flight"], "sustained", l["sustained"]["images_per_sec"])
r = l["roofline"]; print("dominant", r["kernel"], r["frac"], r["min_bound_frac"])
for k, v in l["other_configs"].items():
    print(k, v.get("images_per_sec"), v.get("one_step_in_flight"), v.get("all_mfma_executed_frac"), v.get("error"))
c = l["cpu_baseline"]; print("cpu", c["value"], c["cores"], c["sample"][:160])
g = json.loads(open("$OUT/bench_force_gather.json").readline()); print("force-gather:", g["value"], g["backend"], g["ranks"], g["gather"])
PY
