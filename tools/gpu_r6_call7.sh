#!/bin/bash
OUT=gpurun_out/r6c7; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_hip.py -m gpu -x -q -k "c_pipeline or c_plan_handle or pybind_ext" > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -15 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
