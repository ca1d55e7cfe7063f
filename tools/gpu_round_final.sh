#!/bin/bash
# final evidence session of a round: the whole GPU suite, the bench line as the driver runs it (with the CPU baseline), the forced
# one-rank RCCL line, the 8-rank dress rehearsal line.  usage: tools/gpu_round_final.sh <tag>
TAG=${1:-r6}; OUT=gpurun_out/${TAG}final; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --timeout 1500 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest.log | tail -6
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 300 $OUT/bench.err
timeout 600 python bench.py --gpus 1 --force-gather --steps 20 --warmup 5 --no-cpu-baseline --no-profile --gather-check > $OUT/bench_force_gather.json 2> $OUT/bench_force_gather.err; echo "force-gather rc=$?"
CP_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --batch 2 --steps 5 --warmup 2 --no-cpu-baseline --no-profile --gather-check > $OUT/bench_8rank_rehearsal.json 2> $OUT/bench_8rank_rehearsal.err; echo "8-rank rc=$?"
python - <<PY
import json
l = json.load(open("$OUT/bench.json"))
print(l["value"], "img/s", l["ms_per_step"], "ms; one in flight", l["one_step_in_flight"], "sustained", l["sustained"]["images_per_sec"])
r = l["roofline"]; print("dominant", r["kernel"], r["frac"], r["min_bound_frac"])
for k, v in l["other_configs"].items():
    print(k, v.get("images_per_sec"), v.get("one_step_in_flight"), v.get("all_mfma_executed_frac"), v.get("fps"), v.get("error"))
c = l["cpu_baseline"]; print("cpu", c["value"], c["cores"], c.get("cgroup_cpu_limit"), c.get("res_50_best"))
g = json.loads([x for x in open("$OUT/bench_force_gather.json") if x.startswith("{")][0]); print("force-gather:", g["value"], g["backend"], g["ranks"], g["gather"])
g = json.loads([x for x in open("$OUT/bench_8rank_rehearsal.json") if x.startswith("{")][0]); print("8-rank rehearsal:", g["value"], g["backend"], g["ranks"], g["gather"]["check"])
PY
