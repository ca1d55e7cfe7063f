#!/bin/bash
# round 3, call c: split-K DCNv2 for small-M layers + pipelined wave-private-A DCN kernel
OUT=gpurun_out/r3c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_hip.py -m gpu -x -q -k "dcn_v2" > $OUT/pytest_kernels.log 2>&1; tail -3 $OUT/pytest_kernels.log
timeout 300 python tools/bench_conv.py d64_128,d64_128,d128_64,d512_16 64064,9000064 > $OUT/dcn_micro.txt 2>&1; cat $OUT/dcn_micro.txt
run_bench() {   # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
l=json.load(open("$OUT/bench_$tag.json"))
print("$tag", l["value"], "img/s", l["ms_per_step"], "ms; dom", l["roofline"]["kernel"], l["roofline"]["frac"])
for k,v in l["roofline"]["kernels"].items():
    if "dcn" in k or "splitk" in k: print("   %-52s %2d x %7.3f ms %5.1f%% alg %6.1f exe %6.1f TF" % (k[:52], v["launches"], v["ms_per_step"], 100*v["share"], v["algorithmic_tflops"], v["executed_tflops"]))
PY
}
run_bench default CP_X=0
run_bench nosplit CP_DCN_SPLITK=0
run_bench wp CP_DCN_TILE=9000064
cp $OUT/bench_default.json profiles/bench_line.json
timeout 900 python -m pytest tests/test_engine_hip.py -m gpu -x -q -s -k "timed_configuration or plan_roundtrip or c_plan_handle or critical_path" > $OUT/pytest_engine.log 2>&1; grep -E "passed|failed|worst|Error|error" $OUT/pytest_engine.log | tail
timeout 200 python tools/layer_profile.py dla_34 16 > $OUT/layers_dla34.txt 2>&1; grep -E "dcn|splitk" $OUT/layers_dla34.txt
