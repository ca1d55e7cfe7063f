#!/bin/bash
# scratch: A/B of one change on the GPU box
OUT=gpurun_out/tmp; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_hip.py tests/test_engine_hip.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/tmp/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['step_ms'])
for k,v in d['roofline']['kernels'].items(): print(k, v)
PY
timeout 200 python tools/layer_profile.py dla_34 16 | grep -i "hm\|wh\|hps\|reg\|hp_offset\|total" | head -20
