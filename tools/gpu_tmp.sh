#!/bin/bash
timeout 300 python - <<'PY'
import os, torch
from centerpose_amd import engine, synth
for arch in ("hrnet", "dla_34"):
    sd = synth.make_state_dict(arch); x = synth.make_images(2, 128, 128).cuda()
    e0 = engine.Engine(arch, sd, 2, 128, 128, use_graph=False); ref = [t.clone() for t in e0(x)]
    os.environ["CP_GRAPH"] = "dag"
    e1 = engine.Engine(arch, sd, 2, 128, 128, use_graph=True)
    for _ in range(3): out = e1(x)
    torch.cuda.synchronize()
    print(arch, "dag graph bit-identical:", all(torch.equal(a, b) for a, b in zip(ref, out)))
    os.environ["CP_GRAPH"] = ""
PY
for g in dag; do for cfg in "hrnet 8" "hrnet 16" "dla_34 16" "res_50 8"; do set -- $cfg
  CP_GRAPH=$g timeout 200 python bench.py --arch $1 --batch $2 --no-cpu-baseline --no-profile 2>&1 | tail -1 | python -c "
import json,sys
try:
    l=json.loads(sys.stdin.read()); print('graph=[$g] $1 B=$2', l['value'], 'img/s', l['ms_per_step'], 'ms')
except Exception as e: print('graph=[$g] $1 B=$2 FAILED', e)"
done; done
