"""Dev helper (GPU box): engine vs oracle on a small input + quick timing.  usage: check_engine.py [arch ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from centerpose_amd import synth, engine
from oracle import nets_torch

archs = sys.argv[1:] or ["dla_34", "res_50", "hrnet"]
for arch in archs:
    sd = synth.make_state_dict(arch)
    x = synth.make_images(2, 128, 128)
    ref = nets_torch.forward(arch, sd, x)
    eng = engine.Engine(arch, sd, 2, 128, 128, sigmoid_heads=False, use_graph=False)
    outs = eng(x.cuda())
    torch.cuda.synchronize()
    print(arch, "launches", len(eng.launches), "GF/img %.2f" % (eng.flops_per_image / 1e9))
    for (h, _), o, r in zip(__import__("centerpose_amd").nets.HEADS, outs, ref):
        d = (o.cpu() - r).abs().max().item()
        print("   %-10s max|diff| %.3e  (max|ref| %.2f)" % (h, d, r.abs().max().item()))
if "time" in os.environ.get("CP_CHECK", "time"):
    for arch, B in (("dla_34", 16),):
        sd = synth.make_state_dict(arch)
        eng = engine.Engine(arch, sd, B, 512, 512, use_graph=True)
        x = synth.make_images(B).cuda()
        for _ in range(3):
            eng(x)
        torch.cuda.synchronize()
        t = time.time()
        n = 10
        for _ in range(n):
            eng(x)
        torch.cuda.synchronize()
        dt = (time.time() - t) / n
        print("%s B=%d graph: %.2f ms/batch  %.1f img/s  %.1f TFLOP/s" % (arch, B, dt * 1e3, B / dt, eng.flops_per_image * B / dt / 1e12))
        recs = eng.profile()
        tot = sum(r["ms"] for r in recs)
        print("sum of per-launch times %.2f ms" % tot)
        for r in sorted(recs, key=lambda r: -r["ms"])[:25]:
            print("  %-6s %-55s %7.3f ms  %6.1f TF" % (r["kind"], r["name"][-55:], r["ms"], r["flops"] / r["ms"] / 1e9 if r["ms"] > 0 else 0))
