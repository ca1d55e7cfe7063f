#!/bin/bash
for sm in 0 300 600 1100; do CP_DCN_SMALL=$sm timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; l=json.loads(sys.stdin.read()); ks=[(n,v) for n,v in l['roofline']['kernels'].items() if n.startswith('dcn')]; print('small<$sm', l['value'], [(n[16:30], v['launches'], v['ms_per_step'], v['executed_tflops']) for n,v in ks])"; done
