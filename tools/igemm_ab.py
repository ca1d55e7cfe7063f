"""same-process A/B of generic implicit-GEMM tile codes on DLA-34 / ResNet-50 / HRNet shapes.
usage: python tools/igemm_ab.py tileA,tileB[,..]      tile = BM*1000+BN, + 10000000*L = at most L blocks per CU (LDS padding)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centerpose_amd import ops
TILES = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "64064,128064,128128").split(",")]
#          name                                   B   H    Ci    Co   k  s  residual
SHAPES = {"dla s2 3x3 32->64 @256": (16, 256, 32, 64, 3, 2, 0), "dla s2 3x3 64->128 @128": (16, 128, 64, 128, 3, 2, 0), "dla s2 3x3 128->256 @64": (16, 64, 128, 256, 3, 2, 0),
          "dla s2 3x3 256->512 @32": (16, 32, 256, 512, 3, 2, 0), "dla root 1x1 128->64 @128": (16, 128, 128, 64, 1, 1, 0), "dla root 1x1 256->128 @64": (16, 64, 256, 128, 1, 1, 0),
          "dla root 1x1 448->128 @64": (16, 64, 448, 128, 1, 1, 0), "dla root 1x1 512->256 @32": (16, 32, 512, 256, 1, 1, 0), "dla root 1x1 896->256 @32": (16, 32, 896, 256, 1, 1, 0),
          "r50 1x1 256->64 @128": (8, 128, 256, 64, 1, 1, 0), "r50 1x1 64->256 @128 +res": (8, 128, 64, 256, 1, 1, 1), "r50 1x1 512->128 @64": (8, 64, 512, 128, 1, 1, 0),
          "r50 1x1 128->512 @64 +res": (8, 64, 128, 512, 1, 1, 1), "r50 1x1 1024->256 @32": (8, 32, 1024, 256, 1, 1, 0), "r50 1x1 256->1024 @32 +res": (8, 32, 256, 1024, 1, 1, 1),
          "r50 1x1 512->2048 @16 +res": (8, 16, 512, 2048, 1, 1, 1), "r50 3x3s2 128->128 @128": (8, 128, 128, 128, 3, 2, 0), "r50 3x3s2 256->256 @64": (8, 64, 256, 256, 3, 2, 0),
          "r50 ds 1x1s2 256->512 @128": (8, 128, 256, 512, 1, 2, 0), "hrnet 1x1 64->256 @128": (8, 128, 64, 256, 1, 1, 1), "hrnet s2 3x3 32->64 @128": (8, 128, 32, 64, 3, 2, 0)}
tot = {tl: 0.0 for tl in TILES}
bestsum = 0.0
for name, (B, H, Ci, Co, k, s, r) in SHAPES.items():
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, H, H, Ci, device="cuda", generator=g)
    w = torch.randn(Co, Ci, k, k, device="cuda", generator=g) / (Ci * k * k) ** 0.5
    wp = ops.pack_conv_weight(w)
    sc, sh = ops.fold_bn(Co, None, torch.zeros(Co, device="cuda"))
    Ho = (H + 2 * (k // 2) - k) // s + 1
    res = torch.randn(B, Ho, Ho, Co, device="cuda", generator=g) if r else None
    outs, t = {}, {}
    for rep in range(5):
        for tl in TILES:
            if wp.shape[0] % (tl % 1000):
                continue
            out = torch.empty(B, Ho, Ho, Co, device="cuda")
            la = ops.conv2d_launch([x], wp, sc, sh, out, kh=k, kw=k, stride=s, pad=k // 2, cout=Co, act=1, tile=tl, res=res, split_bf16=False)
            for _ in range(5):
                la.run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(20):
                la.run()
            e1.record(); e1.synchronize()
            t.setdefault(tl, []).append(e0.elapsed_time(e1) / 20)
            outs[tl] = out
    fl = 2.0 * B * Ho * Ho * Co * Ci * k * k
    ks = list(t)
    same = all(torch.equal(outs[ks[0]], outs[tl]) for tl in ks)
    for tl in ks:
        tot[tl] += min(t[tl])
    bestsum += min(min(r_) for r_ in t.values())
    print("%-30s" % name, "  ".join("%d: %.4f ms (%5.1f TF)" % (tl, min(r_), fl / min(r_) / 1e9) for tl, r_ in t.items()), "| identical" if same else "| max diff %.2e" % max((outs[ks[0]] - outs[tl]).abs().max().item() for tl in ks))
print("sum over shapes:", "  ".join("%d: %.4f ms" % (tl, v) for tl, v in tot.items()), " best-per-shape: %.4f ms" % bestsum)
