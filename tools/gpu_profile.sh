#!/bin/bash
# rocprofv3 kernel stats of the bench command (two capture streams and CP_STREAMS=1) + layer table + PMC traffic / MFMA
# utilisation / instruction mix.  usage: tools/gpu_profile.sh <tag> [pmc]
TAG=${1:-prof}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp 2>/dev/null; cd - >/dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/rocprof -o r4 -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
timeout 120 python tools/rocpd_summary.py $(find $OUT/rocprof -name "*.db" | head -1) > $OUT/kernel_stats.md 2>&1; head -12 $OUT/kernel_stats.md
CP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/rocprof1 -o r4 -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile > $OUT/bench_under_rocprof_1stream.json 2> $OUT/rocprof1.err
timeout 120 python tools/rocpd_summary.py $(find $OUT/rocprof1 -name "*.db" | head -1) > $OUT/kernel_stats_1stream.md 2>&1; head -12 $OUT/kernel_stats_1stream.md
rm -rf $OUT/rocprof $OUT/rocprof1
timeout 200 python tools/layer_profile.py dla_34 16 > $OUT/layers_dla34.txt 2>&1; head -2 $OUT/layers_dla34.txt
if [ "$2" = "pmc" ]; then
  timeout 600 python tools/pmc_traffic.py $OUT/pmc_traffic.json | tail -3
  timeout 600 python tools/pmc_mfma_util.py $OUT/mfma_util.json
  rm -rf gpurun_out/pmc_* gpurun_out/pmc
fi
