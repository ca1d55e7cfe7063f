"""same-process A/B of the OPT-IN split-bf16 implicit GEMM (conv_igemm_bf16x3.hip: 3 bf16 terms per fp32 operand, six bf16
MFMAs, fp32 accumulate) against the f32-MFMA kernel (igemm_conv_kernel, auto tile) on res_50 / DLA-34 / hrnet shapes, with the error of
BOTH against an fp64 convolution of image 0.  The first row is the go / no-go GEMM of the verdict (256 -> 256 1x1 @64x64, B = 8; kill
criterion: < 1.3x).  usage: python tools/bf16x3_ab.py [quick]"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centerpose_amd import _lib, ops

SHAPES = [("GO/NO-GO 1x1 256->256 @64 B8", 8, 64, 256, 256, 1, 1),
          ("r50 l1 1x1 256->64 @128 B8", 8, 128, 256, 64, 1, 1), ("r50 l1 1x1 64->256 @128 B8", 8, 128, 64, 256, 1, 1),
          ("r50 l2 1x1 512->128 @64 B8", 8, 64, 512, 128, 1, 1), ("r50 l2 1x1 128->512 @64 B8", 8, 64, 128, 512, 1, 1),
          ("r50 l3 1x1 1024->256 @32 B8", 8, 32, 1024, 256, 1, 1), ("r50 l3 1x1 256->1024 @32 B8", 8, 32, 256, 1024, 1, 1),
          ("r50 l4 1x1 2048->512 @16 B8", 8, 16, 2048, 512, 1, 1), ("r50 l4 1x1 512->2048 @16 B8", 8, 16, 512, 2048, 1, 1),
          ("r50 l2 3x3s2 128->128 @128 B8", 8, 128, 128, 128, 3, 2), ("r50 l3 3x3s2 256->256 @64 B8", 8, 64, 256, 256, 3, 2),
          ("r50 l4 3x3s2 512->512 @32 B8", 8, 32, 512, 512, 3, 2), ("r50 ds 1x1s2 256->512 @128 B8", 8, 128, 256, 512, 1, 2),
          ("dla root 1x1 128->64 @128 B16", 16, 128, 128, 64, 1, 1), ("dla root 1x1 256->128 @64 B16", 16, 64, 256, 128, 1, 1),
          ("dla root 1x1 896->256 @32 B16", 16, 32, 896, 256, 1, 1), ("dla s2 3x3 64->128 @128 B16", 16, 128, 64, 128, 3, 2),
          ("hrnet fuse 1x1 128->64 @32 B8", 8, 32, 128, 64, 1, 1), ("hrnet s2 3x3 32->64 @128 B8", 8, 128, 32, 64, 3, 2),
          ("big 1x1 1024->1024 @64 B8", 8, 64, 1024, 1024, 1, 1)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    SHAPES = SHAPES[:3]


def timeit(la, reps=5, iters=20):
    best = 1e9
    for _ in range(reps):
        for _ in range(3):
            la.run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(iters):
            la.run()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


MODES = ("f32", "128x128", "128x64", "64x128", "64x64")
print("%-34s %-30s %s   %s" % ("shape", "f32 MFMA (auto tile)", "  ".join("%-22s" % ("sb<%s>" % m) for m in MODES[1:]), "max err vs fp64 (image 0): f32 | bf16x3 | rule"))
ratios = []
for name, B, H, Ci, Co, k, s in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, H, H, Ci, device="cuda", generator=g)
    w = torch.randn(Co, Ci, k, k, device="cuda", generator=g) / (Ci * k * k) ** 0.5
    wp = ops.pack_conv_weight(w)
    sc, sh = ops.fold_bn(Co, None, torch.zeros(Co, device="cuda"))
    Ho = (H + 2 * (k // 2) - k) // s + 1
    fl = 2.0 * B * Ho * Ho * Co * Ci * k * k
    ref = F.conv2d(x[:1].permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), None, s, k // 2).clamp_min(0).permute(0, 2, 3, 1)
    cols, errs, tms = [], [], {}
    for mode in MODES:
        out = torch.empty(B, Ho, Ho, Co, device="cuda")
        if mode == "f32":
            la = ops.conv2d_launch([x], wp, sc, sh, out, kh=k, kw=k, stride=s, pad=k // 2, cout=Co, act=1, split_bf16=False)
        else:
            if wp.shape[0] % int(mode.split("x")[1]):
                cols.append("%-22s" % "-"); continue
            os.environ["CP_SPLIT_BF16_TILE"] = mode
            la = ops.conv2d_launch([x], wp, sc, sh, out, kh=k, kw=k, stride=s, pad=k // 2, cout=Co, act=1, split_bf16=True)
            del os.environ["CP_SPLIT_BF16_TILE"]
        t = timeit(la)
        tms[mode] = t
        kern = la.kernel.replace("igemm_conv_kernel", "ig").replace(", 2, 2, 32, false", "")
        cols.append("%.4f ms %6.1f TF %s" % (t, fl / t / 1e9, kern) if mode == "f32" else "%-22s" % ("%.4f ms %5.1f TF" % (t, fl / t / 1e9)))
        if mode == "f32" or len(errs) < 2:
            errs.append((out[:1].double().cpu() - ref).abs().max().item())
        if mode != "f32":
            ratios.append((name, mode, tms["f32"] / t))
    best = max([r for n, m, r in ratios if n == name] or [0.0])
    errs += [float("nan")] * (2 - len(errs))
    code = ops.split_bf16_tile(B * Ho * Ho, wp.shape[0], K=wp.shape[1])
    rule = "f32" if code is None else "%dx%d" % ((code - 3000000) // 1000, code % 1000)
    rr = 1.0 if rule == "f32" else tms["f32"] / tms.get(rule, tms["f32"])
    print("%-34s %-30s %s   %.2e | %.2e   best x%.2f   rule -> %s x%.2f" % (name, cols[0], "  ".join(cols[1:]), errs[0], errs[1], best, rule, rr))
go = max(r for n, m, r in ratios if n.startswith("GO/NO-GO")) if ratios else 0.0
print("\nGO/NO-GO GEMM: split-bf16 / f32 MFMA speed ratio %.2f (kill criterion < 1.3): %s" % (go, "GO" if go >= 1.3 else "NO-GO"))
