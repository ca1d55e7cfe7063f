"""Time the reference-FFI face of DCNv2 (`_ext.dcn_v2_forward`, 14 arguments, NCHW in / out) for 64 -> 64 @128x128, B = 16 (GPU box):
what the reference's own DCN module pays per call -- NCHW <-> NHWC staging of input / offset / mask around the fused kernel -- next to
the bare kernel launch of the in-plan path on the same tensors.  usage: python tools/ext_dcn_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centerpose_amd import _ext, dcn_v2_ext, ops

B, C, Co, H, W = 16, 64, 64, 128, 128
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, C, H, W, device="cuda", generator=g)
w = torch.randn(Co, C, 3, 3, device="cuda", generator=g) * 0.05
b = torch.randn(Co, device="cuda", generator=g)
off = torch.randn(B, 18, H, W, device="cuda", generator=g) * 1.5
m = torch.rand(B, 9, H, W, device="cuda", generator=g)


def timeit(fn, iters=20):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


xn = x.permute(0, 2, 3, 1).contiguous()
om = torch.zeros(B, H, W, 32, device="cuda")
om[..., :18] = off.permute(0, 2, 3, 1)
om[..., 18:27] = m.permute(0, 2, 3, 1)
wp = ops.pack_conv_weight(w)
sc, sh = ops.fold_bn(Co, None, b)
out = torch.empty(B, H, W, Co, device="cuda")
la = ops.dcn_v2_launch(xn, om, wp, sc, sh, out, cout=Co, om_sigmoid=False)
print("in-plan kernel launch (NHWC, packed weights)      %.3f ms" % timeit(la.run))
print("_ext.dcn_v2_forward (pybind, packs per call)      %.3f ms" % timeit(lambda: _ext.dcn_v2_forward(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1)))
print("dcn_v2_ext.dcn_v2_forward (python, packs per call) %.3f ms" % timeit(lambda: dcn_v2_ext.dcn_v2_forward(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1)))
