#!/bin/bash
timeout 300 python tools/sched_try.py hrnet 8 2>&1 | grep -v amdgpu.ids | tail -9
timeout 300 python tools/sched_try.py dla_34 16 2>&1 | grep -v amdgpu.ids | tail -9
