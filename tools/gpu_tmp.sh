#!/bin/bash
timeout 300 python tools/cplan_bench.py dla_34 16 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/cplan_bench.py hrnet 8 2>&1 | grep -v amdgpu.ids
