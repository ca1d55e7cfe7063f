#!/bin/bash
# round 5, GPU session 1: split-bf16 go / no-go (micro loop + real kernel A/B), igemm tile / occupancy A/B, full GPU test suite, bench line
OUT=gpurun_out/r5c1; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 tools/micro/bf16x3_loop > $OUT/bf16x3_loop.txt 2>&1; echo "micro rc=$?"
timeout 900 python tools/bf16x3_ab.py > $OUT/bf16x3_ab.txt 2>&1; echo "bf16x3_ab rc=$?"
timeout 900 python tools/igemm_ab.py 64064,128064,128128,40064064,20064064,30128064,20128064,10128128 > $OUT/igemm_ab.txt 2>&1; echo "igemm_ab rc=$?"
timeout 1500 python -m pytest tests -m gpu -q -s --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -5
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -c 600 $OUT/bench.err
tail -n 30 $OUT/bf16x3_loop.txt
tail -n 30 $OUT/bf16x3_ab.txt
