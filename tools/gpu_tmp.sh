#!/bin/bash
for d in 0 3 2; do echo "CP_DCN_DEBUG=$d"; CP_DCN_DEBUG=$d timeout 120 python tools/bench_conv.py d64_128,d128_64,d512_16 0; done
for d in 0 3 2; do CP_DCN_DEBUG=$d timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; l=json.loads(sys.stdin.read()); k=[v for n,v in l['roofline']['kernels'].items() if n.startswith('dcn')][0]; print('debug $d', l['value'], 'dcn ms', k['ms_per_step'], k['executed_tflops'])"; done
