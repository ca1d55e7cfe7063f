#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_hip.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python bench.py --no-cpu-baseline --no-profile 2>/dev/null | head -c 200; echo
