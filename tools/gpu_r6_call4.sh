#!/bin/bash
OUT=gpurun_out/r6c4; mkdir -p $OUT
export TMPDIR=/tmp
cd tools && timeout 600 python hrnet_group_probe.py > ../$OUT/hrnet_group_probe.txt 2>&1; echo "rc=$?"; cat ../$OUT/hrnet_group_probe.txt
