"""Error of the 3x3 kernels against an fp64 convolution (GPU box): direct (patch / implicit-GEMM) kernel, Winograd F(2x2,3x3),
Winograd F(2x4,3x3).  usage: python tools/wino_check.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centerpose_amd import ops

g = torch.Generator().manual_seed(1)
for cin, cout, H in [(64, 256, 128), (128, 128, 64), (256, 256, 32), (512, 512, 16)]:
    x = torch.randn(2, cin, H, H, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** .5
    ref = F.conv2d(x.double(), w.double(), None, 1, 1).permute(0, 2, 3, 1)
    wp = ops.pack_conv_weight(w.cuda())
    sc, sh = torch.ones(wp.shape[0], device="cuda"), torch.zeros(wp.shape[0], device="cuda")
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    for nm, u, tile in (("direct", None, 0), ("F(2x2,3x3)", ops.pack_wino_weight(wp, cin, cout), 0),
                        ("F(2x4,3x3)", ops.pack_wino24_weight(wp, cin, cout), ops.WINO24)):
        o = torch.empty(2, H, H, cout, device="cuda")
        ops.conv2d([xn], wp, sc, sh, o, kh=3, kw=3, stride=1, pad=1, cout=cout, wino=u, tile=tile)
        e = (o.cpu().double() - ref).abs()
        print("%4d -> %4d @%3d  %-11s max %.2e rms %.2e (ref max %.2f)" % (cin, cout, H, nm, e.max(), e.pow(2).mean().sqrt(), ref.abs().max()))
