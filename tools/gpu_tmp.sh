#!/bin/bash
timeout 300 python bench.py --steps 2000 --warmup 20 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('2000 steps:', l['value'], l['ms_per_step'], l['step_ms'])"
