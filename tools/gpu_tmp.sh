#!/bin/bash
# scratch
OUT=gpurun_out/tmp; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
for cfg in "dla_34 16" "res_50 8" "hrnet 8"; do
  set -- $cfg
  timeout 200 python bench.py --arch $1 --batch $2 --no-cpu-baseline > $OUT/b_$1_$2.json 2> $OUT/b_$1_$2.err || tail -5 $OUT/b_$1_$2.err
  python - <<PY
import json
l=json.load(open("$OUT/b_$1_$2.json"))
print("$1 B=$2", l["value"], "img/s", l["ms_per_step"], "ms/step", l["step_ms"]["median"], l["step_ms"]["p10"], l["step_ms"]["p90"], l["roofline"]["kernel"], l["roofline"]["frac"])
PY
done
