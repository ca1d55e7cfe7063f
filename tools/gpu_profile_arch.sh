#!/bin/bash
# one-stream rocprofv3 kernel stats of another configuration: usage tools/gpu_profile_arch.sh <arch> <batch> <out.md>
ARCH=$1; B=$2; OUT=$3; export TMPDIR=/tmp; D=gpurun_out/prof_$ARCH; rm -rf $D
CP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $D -o p -- python bench.py --arch $ARCH --batch $B --in-flight 1 --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-other-configs > $D.json 2> $D.err
python tools/rocpd_summary.py $(find $D -name "*.db" | head -1) > $OUT 2>&1
python - <<PY >> $OUT
import json
l = json.loads([x for x in open("$D.json") if x.startswith("{")][0])
print("\nbench line of the same run (CP_STREAMS=1, under rocprofv3): %s B=%d  %.1f img/s, %.3f ms/step" % ("$ARCH", $B, l["value"], l["ms_per_step"]))
PY
rm -rf $D; head -8 $OUT; tail -3 $OUT
