#!/bin/bash
# round-2 profiles at HEAD
OUT=gpurun_out/r2p; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
bash tools/gpu_profile.sh r2p pmc > $OUT/profile.log 2>&1
CP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/rocprof1 -o r2 -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile > $OUT/bench_under_rocprof_1stream.json 2> $OUT/rocprof1.err
timeout 120 python tools/rocpd_summary.py $(find $OUT/rocprof1 -name "*.db" | head -1) > $OUT/kernel_stats_1stream.md 2>&1
cp $OUT/pmc_traffic.json profiles/r2_pmc_traffic.json
timeout 400 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 200 python tools/layer_profile.py res_50 16 > $OUT/layers_res50.txt 2>&1
timeout 200 python tools/layer_profile.py hrnet 16 > $OUT/layers_hrnet.txt 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2p/bench_line.json'))
r=d['roofline']
print(d['value'], d['ms_per_step'], d['step_ms'])
print({k:v for k,v in r.items() if k not in ('kernels','definition')})
PY
