"""same-process A/B of the 16-channel 3x3 kernel (tile=16) against the kernels it replaces (tile=3 patch / tile=128032 generic).
usage: python tools/c16_ab.py [B]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centerpose_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for name, (H, Co, s, tiles) in {"level0 16->16 s1 @512": (512, 16, 1, (3, 16)), "level1 16->32 s2 @512": (512, 32, 2, (128032, 16))}.items():
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, H, H, 16, device="cuda", generator=g)
    w = torch.randn(Co, 16, 3, 3, device="cuda", generator=g) / 12
    wp = ops.pack_conv_weight(w)
    sc, sh = ops.fold_bn(Co, None, torch.zeros(Co, device="cuda"))
    Ho = (H - 1) // s + 1
    outs, t = {}, {}
    for rep in range(6):
        for tl in tiles:
            out = torch.empty(B, Ho, Ho, Co, device="cuda")
            la = ops.conv2d_launch([x], wp, sc, sh, out, kh=3, kw=3, stride=s, pad=1, cout=Co, act=1, tile=tl)
            for _ in range(5):
                la.run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(20):
                la.run()
            e1.record(); e1.synchronize()
            t.setdefault(tl, []).append(e0.elapsed_time(e1) / 20)
            outs[tl] = out
    fl = 2.0 * B * Ho * Ho * Co * 144
    err = (outs[tiles[0]] - outs[tiles[1]]).abs().max().item()
    print(name, "  ".join("tile=%s: %.4f ms (%.1f TF, %.3f of peak)" % (tl, min(r), fl / min(r) / 1e9, fl / min(r) / 1e9 / 157.3) for tl, r in t.items()),
          "| max diff %.2e" % err)
