#!/bin/bash
# round 5, GPU session 2: split-bf16 loop ceilings (micro) + the whole GPU parity suite with CP_SPLIT_BF16=1 (unchanged tolerances)
OUT=gpurun_out/r5c2; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 tools/micro/bf16x3_loop > $OUT/bf16x3_loop.txt 2>&1; echo "micro rc=$?"
cat $OUT/bf16x3_loop.txt
CP_SPLIT_BF16=1 timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x > $OUT/pytest_split_bf16.log 2>&1; echo "pytest (CP_SPLIT_BF16=1) rc=$?" | tee -a $OUT/pytest_split_bf16.log
tail -n 15 $OUT/pytest_split_bf16.log
