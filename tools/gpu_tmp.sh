#!/bin/bash
export TMPDIR=/tmp
timeout 400 python tools/pmc_mfma_util.py gpurun_out/r2p/mfma_util.json 2>&1 | tail -20
