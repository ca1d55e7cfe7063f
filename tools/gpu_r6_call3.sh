#!/bin/bash
# round 6, call 3: two go / no-go measurements BEFORE building -- the fused HRNet BasicBlock bounds and the split-bf16
# "stage C" loop with pre-split operands + LDS-DMA
OUT=gpurun_out/r6c3; mkdir -p $OUT
export TMPDIR=/tmp
( cd tools/micro && hipcc -O3 --offload-arch=gfx950 -o bf16x3_loop bf16x3_loop.hip 2>/dev/null; timeout 300 ./bf16x3_loop ) > $OUT/bf16x3_loop.txt 2>&1; echo "micro rc=$?"; cat $OUT/bf16x3_loop.txt
for args in "32 128 8" "64 64 8" "32 128 16"; do timeout 300 python tools/basicblock_probe.py $args; done > $OUT/basicblock_probe.txt 2>&1; echo "probe rc=$?"; cat $OUT/basicblock_probe.txt
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
