"""same-process A/B of generic implicit-GEMM tile codes on DLA-34 / ResNet-50 shapes.  usage: python tools/igemm_ab.py tileA,tileB[,..]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centerpose_amd import ops
TILES = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "64064,2064064").split(",")]
SHAPES = {"dla s2 3x3 32->64 @256": (16, 256, 32, 64, 3, 2), "dla s2 3x3 64->128 @128": (16, 128, 64, 128, 3, 2), "dla s2 3x3 128->256 @64": (16, 64, 128, 256, 3, 2),
          "dla s2 3x3 256->512 @32": (16, 32, 256, 512, 3, 2), "dla root 1x1 128->64 @128": (16, 128, 128, 64, 1, 1), "dla root 1x1 256->128 @64": (16, 64, 256, 128, 1, 1),
          "dla root 1x1 448->128 @64": (16, 64, 448, 128, 1, 1), "dla root 1x1 512->256 @32": (16, 32, 512, 256, 1, 1), "dla root 1x1 896->256 @32": (16, 32, 896, 256, 1, 1),
          "r50 1x1 256->64 @128": (8, 128, 256, 64, 1, 1), "r50 1x1 64->256 @128": (8, 128, 64, 256, 1, 1), "r50 1x1 512->128 @64": (8, 64, 512, 128, 1, 1),
          "r50 1x1 128->512 @64": (8, 64, 128, 512, 1, 1), "r50 1x1 1024->256 @32": (8, 32, 1024, 256, 1, 1), "r50 1x1 256->1024 @32": (8, 32, 256, 1024, 1, 1)}
for name, (B, H, Ci, Co, k, s) in SHAPES.items():
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, H, H, Ci, device="cuda", generator=g)
    w = torch.randn(Co, Ci, k, k, device="cuda", generator=g) / (Ci * k * k) ** 0.5
    wp = ops.pack_conv_weight(w)
    sc, sh = ops.fold_bn(Co, None, torch.zeros(Co, device="cuda"))
    Ho = (H + 2 * (k // 2) - k) // s + 1
    outs, t = {}, {}
    for rep in range(5):
        for tl in TILES:
            out = torch.empty(B, Ho, Ho, Co, device="cuda")
            la = ops.conv2d_launch([x], wp, sc, sh, out, kh=k, kw=k, stride=s, pad=k // 2, cout=Co, act=1, tile=tl)
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
    same = all(torch.equal(outs[TILES[0]], outs[tl]) for tl in TILES)
    print("%-28s" % name, "  ".join("%d: %.4f ms (%.1f TF)" % (tl, min(r), fl / min(r) / 1e9) for tl, r in t.items()), "| identical" if same else "| max diff %.2e" % max((outs[TILES[0]] - outs[tl]).abs().max().item() for tl in TILES))
