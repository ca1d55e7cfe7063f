#!/bin/bash
# round 3, call b: fused hps / hm_hp heads + wave-private-A DCN experiment
OUT=gpurun_out/r3b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_hip.py -m gpu -x -q -k "head3x3 or dcn_v2_vs_scalar" > $OUT/pytest_kernels.log 2>&1; tail -3 $OUT/pytest_kernels.log
timeout 300 python tools/bench_conv.py d64_128,d128_64,d512_16 0,64064,9000064 > $OUT/dcn_micro.txt 2>&1; cat $OUT/dcn_micro.txt
for t in 0 9000064; do
  CP_DCN_TILE=$t timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_dcn$t.json 2> $OUT/bench_dcn$t.err
  python - <<PY
import json
l=json.load(open("$OUT/bench_dcn$t.json"))
print("CP_DCN_TILE=$t", l["value"], "img/s", l["ms_per_step"], "ms; dom", l["roofline"]["kernel"], l["roofline"]["frac"])
for k,v in l["roofline"]["kernels"].items(): print("   %-52s %2d x %7.3f ms %5.1f%% alg %6.1f exe %6.1f TF" % (k[:52], v["launches"], v["ms_per_step"], 100*v["share"], v["algorithmic_tflops"], v["executed_tflops"]))
PY
done
cp $OUT/bench_dcn0.json profiles/bench_line.json
timeout 900 python -m pytest tests/test_engine_hip.py -m gpu -x -q -s -k "timed_configuration or batch_invariance or c_plan_handle" > $OUT/pytest_engine.log 2>&1; grep -E "passed|failed|worst" $OUT/pytest_engine.log | tail
# is level0 memory-bound?  B=4: its 67 MB input fits the 256 MB Infinity Cache (hot = every launch repeated back to back)
timeout 200 python tools/layer_profile.py dla_34 4 > $OUT/layers_b4_seq.txt 2>&1; CP_PROFILE_HOT=1 timeout 200 python tools/layer_profile.py dla_34 4 > $OUT/layers_b4_hot.txt 2>&1
head -5 $OUT/layers_b4_seq.txt; head -5 $OUT/layers_b4_hot.txt
