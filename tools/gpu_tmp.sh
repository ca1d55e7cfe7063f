#!/bin/bash
timeout 300 python -m pytest tests/test_decode_hip.py -m gpu -q -x 2>&1 | tail -2
timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; l=json.loads(sys.stdin.read()); print(l['value'], l['ms_per_step'], l['decode'])"
