"""same-process A/B of the fused head launch: F(2x2) V-stationary kernel vs the F(2x4) eight-wave kernel, B = 16, 64 -> 256 @128x128.
usage: python tools/head_ab.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centerpose_amd import ops
B, H, W, hc = 16, 128, 128, 256
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, H, W, 64, device="cuda", generator=g)
w3 = torch.randn(hc, 64, 3, 3, device="cuda", generator=g) / 24.0
wp3 = ops.pack_conv_weight(w3)
sc, sh = ops.fold_bn(hc, None, torch.zeros(hc, device="cuda"))
for n2, act2 in ((1, 2), (2, 0), (17, 2), (34, 0)):
    w1 = (torch.randn(n2, hc, device="cuda", generator=g) / 16.0).contiguous()
    b1 = torch.randn(n2, device="cuda", generator=g)
    res, outs = {}, {}
    for rep in range(3):
        for w24 in (False, True):
            u = (ops.pack_wino24_weight if w24 else ops.pack_wino_weight)(wp3, 64, hc)
            out = torch.empty(B, n2, H, W, device="cuda")
            la = ops.head3x3_1x1_launch(x, u, sc, sh, w1, b1, out, hc=hc, act2=act2, wino24=w24)
            for _ in range(5):
                la.run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(15):
                la.run()
            e1.record(); e1.synchronize()
            res.setdefault(w24, []).append(e0.elapsed_time(e1) / 15)
            outs[w24] = out
    fl = 2.0 * B * H * W * (hc * 64 * 9 + n2 * hc)
    d = (outs[True] - outs[False]).abs().max().item()
    print("n2=%2d  F(2x2) %.3f ms (%.0f TF)   F(2x4) %.3f ms (%.0f TF)   max |diff| %.2e" % (n2, min(res[False]), fl / min(res[False]) / 1e9, min(res[True]), fl / min(res[True]) / 1e9, d))
