#!/bin/bash
OUT=gpurun_out/r6c9; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_hip.py -m gpu -q -x -k "pointwise" 2>&1 | tail -5
cd tools && timeout 600 python pointwise_probe.py 2>&1 | grep -v amdgpu | tee ../$OUT/pointwise_probe.txt
