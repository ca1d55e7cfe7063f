#!/bin/bash
# scratch: A/B of one change on the GPU box
TILES="0 8064064 9064064 0 8064064 9064064" bash tools/gpu_dcn_ab.sh tmp
