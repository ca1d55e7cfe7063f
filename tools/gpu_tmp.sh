#!/bin/bash
# scratch: the command of the moment for one gpurun call (see gpu_check.sh / gpu_profile.sh for the kept ones)
