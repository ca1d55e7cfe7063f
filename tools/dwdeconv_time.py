"""time the IDAUp depthwise deconv + add launches of DLA-34 at B=16.  A/B: run under tools/with_lib.py OLD.so"""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centerpose_amd import _lib
L = _lib.lib()
tot = 0.0
for (H, C, f) in [(64, 64, 2), (32, 128, 2), (64, 64, 2), (16, 256, 2), (32, 128, 2), (64, 64, 2), (32, 64, 4), (16, 64, 8)]:
    B = 16
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, H, H, C, device="cuda", generator=g)
    w = torch.randn(4 * f * f, C, device="cuda", generator=g)
    add = torch.randn(B, H * f, H * f, C, device="cuda", generator=g)
    out = torch.empty_like(add)
    best = 1e9
    for rep in range(4):
        for _ in range(3):
            rc = L.cp_dw_deconv_add_nhwc_f32(_lib.ptr(x), C, _lib.ptr(w), _lib.ptr(add), C, _lib.ptr(out), C, B, H, H, C, f, _lib.stream())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10):
            rc = L.cp_dw_deconv_add_nhwc_f32(_lib.ptr(x), C, _lib.ptr(w), _lib.ptr(add), C, _lib.ptr(out), C, B, H, H, C, f, _lib.stream())
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    mb = (x.numel() + 2 * add.numel()) * 4 / 1e6
    tot += best
    bits = int(out.view(torch.int32).to(torch.int64).sum().item())                # same bits <=> same value here (up to permutations of equal sums)
    print("in %3dx%-3d C=%3d f=%d: %.4f ms  %.0f MB  %.2f GB/ms = TB/s  bit-sum %d" % (H, H, C, f, best, mb, mb / best / 1e3, bits))
print("sum of the eight launches: %.4f ms" % tot)
