"""Block-by-block comparison of the mobilenetv3 plan against the oracle (GPU box, CP_BUFFER_REUSE=0)."""
import os, re, sys
os.environ["CP_BUFFER_REUSE"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from centerpose_amd import engine, synth
from oracle import nets_torch as nt

arch = "mobilenetv3"
sd = synth.make_state_dict(arch)
x = synth.make_images(1, 128, 128, seed=7)
eng = engine.Engine(arch, sd, 1, 128, 128, sigmoid_heads=False, use_graph=False)
eng(x.cuda())
torch.cuda.synchronize()
# oracle block outputs
p = "backbone_model"
with torch.no_grad():
    out = nt._hswish(nt._bn(sd, p + ".bn1", nt._conv(sd, p + ".conv1", x, 2, 1)))
    refs = {"conv1": out}
    for si, blocks in enumerate(nt.MBV3):
        for bi, (k, ci, ce, co, act, se, st) in enumerate(blocks):
            out = nt._mb_block(sd, "%s.bneck%d.%d" % (p, si, bi), out, k, ci, co, act, se, st)
            refs["bneck%d.%d" % (si, bi)] = out
last = {}
for kind, name, _, l in eng.launches:
    m = re.search(r"(bneck\d\.\d+)", name)
    key = m.group(1) if m else ("conv1" if name.endswith("backbone_model.conv1") else None)
    if name == "se.scale":
        key = cur
    if key:
        cur = key
        last[key] = (name, l.out)
for key, r in refs.items():
    name, t = last[key]
    C = r.shape[1]
    o = t[..., :C].permute(0, 3, 1, 2).cpu()
    print("%-10s %-45s err %.3e  max %.3e  pad-max %.3e" % (key, name[-45:], (o - r).abs().max().item(), r.abs().max().item(),
                                                            t[..., C:].abs().max().item() if t.shape[3] > C else 0))
