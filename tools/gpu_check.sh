#!/bin/bash
# One GPU-box session: parity tests, bench line, rocprofv3 kernel stats.  usage: tools/gpu_check.sh <tag> [pytest args]
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q "$@" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json | head -c 6000
