#!/bin/bash
# usage: tools/gpu_tests.sh <tag> [pytest args...]   (GPU parity tests only)
TAG=${1:-t}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q "$@" > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -25 $OUT/pytest.log
