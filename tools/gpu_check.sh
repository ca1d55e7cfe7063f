#!/bin/bash
# One GPU-box session: bench line (-> profiles/bench_line.json), parity tests, per-launch table, and the check that the kernels
# test_timed_configuration_parity ran (gpurun_out/tested_kernels_dla_34.json) are the kernels the bench line timed.
# usage: tools/gpu_check.sh <tag> [pytest args]
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
l = json.load(open("$OUT/bench.json"))
print(l["value"], "img/s", l["ms_per_step"], "ms; capture", l["graph_capture"], "; dom", l["roofline"]["kernel"], l["roofline"]["frac"])
for k, v in l["roofline"]["kernels"].items():
    print("   %-52s %2d x %7.3f ms %5.1f%% alg %6.1f exe %6.1f TF" % (k[:52], v["launches"], v["ms_per_step"], 100 * v["share"], v["algorithmic_tflops"], v["executed_tflops"]))
for k, v in l.get("other_configs", {}).items():
    print("   other", k, {q: v.get(q) for q in ("images_per_sec", "ms_per_step", "all_mfma_executed_frac", "dominant_kernel", "dominant_frac", "error")})
c = l.get("cpu_baseline", {})
print("   cpu value", c.get("value"), "cores", c.get("cores"), "|", c.get("sample"))
for k, v in c.items():
    if isinstance(v, dict):
        print("   cpu", k, v.get("images_per_sec"), "img/s @", v.get("threads"), "threads", v.get("sweep_images_per_sec"))
    elif isinstance(v, list):
        print("   cpu", k, [(q.get("processes"), q.get("images_per_sec", q.get("error"))) for q in v])
PY
cp $OUT/bench.json profiles/bench_line.json
timeout 1800 python -m pytest tests -m gpu -x -q -s "$@" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
grep -E "passed|failed|error|worst error|detections compared|equal names" $OUT/pytest.log | tail -14
python - <<PY
import json, os
p = "gpurun_out/tested_kernels_dla_34.json"
if os.path.exists(p):
    tested = sorted(json.load(open(p))["kernels"])
    timed = sorted(json.load(open("$OUT/bench.json"))["roofline"]["kernels"])
    print("tested kernels == timed kernels:", tested == timed, sorted(set(tested) ^ set(timed)))
else:
    print("no tested-kernel list (test_timed_configuration_parity did not run)")
PY
timeout 200 python tools/layer_profile.py dla_34 16 > $OUT/layers_dla34.txt 2>&1; head -2 $OUT/layers_dla34.txt | tail -1
