#!/bin/bash
# rocprofv3 kernel stats of the bench command + layer table + PMC traffic.  usage: tools/gpu_profile.sh <tag> [pmc]
TAG=${1:-prof}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp 2>/dev/null; cd - >/dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/rocprof -o r2 -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
ls -R $OUT/rocprof | head -20
timeout 120 python tools/rocpd_summary.py $(find $OUT/rocprof -name "*.db" | head -1) > $OUT/kernel_stats.md 2>&1; head -30 $OUT/kernel_stats.md
timeout 200 python tools/layer_profile.py dla_34 16 > $OUT/layers_dla34.txt 2>&1; head -3 $OUT/layers_dla34.txt
if [ "$2" = "pmc" ]; then timeout 600 python tools/pmc_traffic.py $OUT/pmc_traffic.json; fi
