#!/bin/bash
# per-round evidence session: rocprofv3 kernel stats of the bench command (two capture streams + overlap / idle-gap report, and CP_STREAMS=1), per-launch tables,
# one-stream stats of res_50 B=8 (f32 and CP_SPLIT_BF16=1) and hrnet B=8, PMC traffic / MFMA utilisation.  usage: tools/gpu_round_profile.sh <tag> [pmc]
TAG=${1:-r6}; OUT=gpurun_out/${TAG}prof; mkdir -p $OUT; export TMPDIR=/tmp
# (a) the default bench: two steps in flight in one two-stream graph
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/rocprof -o $TAG -- python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-profile --no-other-configs > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
DB=$(find $OUT/rocprof -name "*.db" | head -1)
timeout 120 python tools/rocpd_summary.py $DB > $OUT/kernel_stats.md 2>&1; head -8 $OUT/kernel_stats.md
timeout 120 python tools/overlap_report.py $DB > $OUT/two_steps_in_flight_overlap.txt 2>&1; head -4 $OUT/two_steps_in_flight_overlap.txt
# (b) one step per replay, two capture streams (rounds 2-4's arrangement): the idle gaps / single-kernel time the second step fills
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/rocprof2 -o $TAG -- python bench.py --in-flight 1 --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-other-configs > $OUT/bench_under_rocprof_one_step.json 2> $OUT/rocprof2.err
DB2=$(find $OUT/rocprof2 -name "*.db" | head -1)
timeout 120 python tools/overlap_report.py $DB2 > $OUT/two_stream_overlap.txt 2>&1; head -4 $OUT/two_stream_overlap.txt
rm -rf $OUT/rocprof2
# (c) one step per replay, ONE stream: the per-kernel averages the bench line's in-sequence HIP-event figures must agree with
CP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/rocprof1 -o $TAG -- python bench.py --in-flight 1 --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-other-configs > $OUT/bench_under_rocprof_1stream.json 2> $OUT/rocprof1.err
DB1=$(find $OUT/rocprof1 -name "*.db" | head -1)
timeout 120 python tools/rocpd_summary.py $DB1 > $OUT/kernel_stats_1stream.md 2>&1; head -8 $OUT/kernel_stats_1stream.md
timeout 120 python tools/overlap_report.py $DB1 > $OUT/one_stream_gaps.txt 2>&1; head -2 $OUT/one_stream_gaps.txt
rm -rf $OUT/rocprof $OUT/rocprof1
timeout 200 python tools/layer_profile.py dla_34 16 > $OUT/layers_dla34.txt 2>&1; head -2 $OUT/layers_dla34.txt | tail -1
timeout 200 python tools/layer_profile.py res_50 8 > $OUT/layers_res50.txt 2>&1; head -2 $OUT/layers_res50.txt | tail -1
timeout 200 python tools/layer_profile.py hrnet 8 > $OUT/layers_hrnet.txt 2>&1; head -2 $OUT/layers_hrnet.txt | tail -1
CP_SPLIT_BF16=1 timeout 200 python tools/layer_profile.py res_50 8 > $OUT/layers_res50_split_bf16.txt 2>&1; head -2 $OUT/layers_res50_split_bf16.txt | tail -1
bash tools/gpu_profile_arch.sh res_50 8 $OUT/res50_kernel_stats_1stream.md > /dev/null 2>&1; tail -2 $OUT/res50_kernel_stats_1stream.md
bash tools/gpu_profile_arch.sh hrnet 8 $OUT/hrnet_kernel_stats_1stream.md > /dev/null 2>&1; tail -2 $OUT/hrnet_kernel_stats_1stream.md
CP_SPLIT_BF16=1 bash tools/gpu_profile_arch.sh res_50 8 $OUT/res50_split_bf16_kernel_stats_1stream.md > /dev/null 2>&1; head -5 $OUT/res50_split_bf16_kernel_stats_1stream.md; tail -2 $OUT/res50_split_bf16_kernel_stats_1stream.md
if [ "$2" = "pmc" ]; then
  timeout 600 python tools/pmc_traffic.py $OUT/pmc_traffic.json | tail -3
  timeout 600 python tools/pmc_mfma_util.py $OUT/mfma_util.json | tail -5
  rm -rf gpurun_out/pmc_* gpurun_out/pmc
fi
