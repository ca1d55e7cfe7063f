#!/bin/bash
OUT=gpurun_out/r6c5; mkdir -p $OUT
export TMPDIR=/tmp
( cd tools/micro && hipcc -O3 --offload-arch=gfx950 -o bf16x3_loop bf16x3_loop.hip 2>/dev/null; timeout 300 ./bf16x3_loop ) > $OUT/bf16x3_loop.txt 2>&1; echo "micro rc=$?"; grep -E "stage C|register-fed|ds_read|f32 MFMA" $OUT/bf16x3_loop.txt
