#!/bin/bash
# rocprofv3 kernel stats of the bench command with ONE capture stream (no overlap: per-kernel durations comparable with the in-sequence table)
OUT=gpurun_out/r2n; mkdir -p $OUT; export TMPDIR=/tmp
CP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/rocprof1 -o r2 -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile > $OUT/bench_under_rocprof_1stream.json 2> $OUT/rocprof1.err
timeout 120 python tools/rocpd_summary.py $(find $OUT/rocprof1 -name "*.db" | head -1) > $OUT/kernel_stats_1stream.md 2>&1; head -16 $OUT/kernel_stats_1stream.md
cat $OUT/bench_under_rocprof_1stream.json | head -c 400
