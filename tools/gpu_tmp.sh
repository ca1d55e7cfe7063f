#!/bin/bash
timeout 300 python -m pytest tests/test_conv_hip.py -m gpu -q -x -k "head3x3 or winograd" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_engine_hip.py -m gpu -q -x -k "dla_34 or process or plan" 2>&1 | tail -4
for f in 1 0; do CP_FUSE_HEADS=$f timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; l=json.loads(sys.stdin.read()); print('fuse=$f', l['value'], l['ms_per_step']); [print('   %-46s n=%2d %6.3f ms' % (k[:46], v['launches'], v['ms_per_step'])) for k,v in l['roofline']['kernels'].items() if 'vs64' in k or '256, 16' in k]"; done
