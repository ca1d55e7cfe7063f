#!/bin/bash
for sd in 0.3 1.0 1.5 3 6; do echo "om std $sd"; CP_OM_STD=$sd timeout 120 python tools/bench_conv.py d64_128,d128_64,d512_16 0; done
timeout 200 python tools/om_stats.py dla_34
