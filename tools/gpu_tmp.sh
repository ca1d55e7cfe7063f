#!/bin/bash
for ns in 3 4; do echo "CP_STREAMS=$ns"; CP_STREAMS=$ns timeout 200 python bench.py --arch hrnet --batch 8 --no-cpu-baseline --no-profile 2>&1 | tail -5 | cut -c1-300; done
