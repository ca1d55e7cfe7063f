"""time the 7x7 stem launches (DLA 3->16 / s1 at B=16, ResNet 3->64 / s2 at B=8, 512x512).  A/B: run under tools/with_lib.py OLD.so"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centerpose_amd import ops, _lib
for name, (B, co, st) in {"dla base_layer 3->16 s1 B=16": (16, 16, 1), "resnet conv1 3->64 s2 B=8": (8, 64, 2)}.items():
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, 3, 512, 512, device="cuda", generator=g)
    w = torch.randn(co, 3, 7, 7, device="cuda", generator=g) * 0.1
    wp = ops.pack_stem7_weight(w)
    sc, sh = ops.fold_bn(co, None, torch.zeros(co, device="cuda"))
    Ho = 512 // st
    out = torch.empty(B, Ho, Ho, co, device="cuda")
    la = ops.stem7x7_launch(x, wp, sc, sh, out, st)
    best = 1e9
    for rep in range(5):
        for _ in range(5):
            la.run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            la.run()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    fl = 2.0 * B * Ho * Ho * co * 147
    print("%-30s %-34s %.4f ms  %.1f TF (%.3f of peak)  checksum %.6e" % (name, _lib.lib().cp_last_kernel().decode(), best, fl / best / 1e9, fl / best / 1e9 / 157.3, out.double().sum().item()))
# HRNet conv1: 3 -> 64, 3x3 / stride 2 (generic conv entry with the NCHW input)
B = 8
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, 3, 512, 512, device="cuda", generator=g)
w = torch.randn(64, 3, 3, 3, device="cuda", generator=g) * 0.1
wp = ops.pack_conv_weight(w, stem=True)
sc, sh = ops.fold_bn(64, None, torch.zeros(64, device="cuda"))
out = torch.empty(B, 256, 256, 64, device="cuda")
la = ops.conv2d_launch([x], wp, sc, sh, out, kh=3, kw=3, stride=2, pad=1, cout=64, act=1, in_nchw=True)
best = 1e9
for rep in range(5):
    for _ in range(5):
        la.run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        la.run()
    e1.record(); e1.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
mb = (x.numel() + out.numel()) * 4 / 1e6
print("%-30s %-34s %.4f ms  %.0f MB -> %.2f TB/s  checksum %.6e" % ("hrnet conv1 3->64 3x3 s2 B=8", _lib.lib().cp_last_kernel().decode(), best, mb, mb / best / 1e3, out.double().sum().item()))
