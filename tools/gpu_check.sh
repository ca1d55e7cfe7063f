#!/bin/bash
# One GPU-box session: bench line (-> profiles/bench_line.json, which test_timed_configuration_parity compares its kernel set
# with), parity tests, per-launch table.  usage: tools/gpu_check.sh <tag> [pytest args]
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
l = json.load(open("$OUT/bench.json"))
print(l["value"], "img/s", l["ms_per_step"], "ms; capture", l["graph_capture"], "; dom", l["roofline"]["kernel"], l["roofline"]["frac"])
for k, v in l["roofline"]["kernels"].items():
    print("   %-52s %2d x %7.3f ms %5.1f%% alg %6.1f exe %6.1f TF" % (k[:52], v["launches"], v["ms_per_step"], 100 * v["share"], v["algorithmic_tflops"], v["executed_tflops"]))
print("   cpu", {k: (v["images_per_sec"], v["threads"], v["sweep_images_per_sec"]) for k, v in l["cpu_baseline"].items() if isinstance(v, dict)})
PY
cp $OUT/bench.json profiles/bench_line.json
timeout 1500 python -m pytest tests -m gpu -x -q -s "$@" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
grep -E "passed|failed|error|worst error|detections compared" $OUT/pytest.log | tail -12
timeout 200 python tools/layer_profile.py dla_34 16 > $OUT/layers_dla34.txt 2>&1; head -1 $OUT/layers_dla34.txt
