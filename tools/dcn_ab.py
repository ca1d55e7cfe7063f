"""same-process A/B of DCNv2 tile codes (cp_dcn_desc.tile) on DLA-34's DCN shapes at B=16, outputs compared.
usage: python tools/dcn_ab.py [tileA,tileB,...]   (0 = the library's own choice)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centerpose_amd import ops
TILES = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "0,8064064").split(",")]
SHAPES = {"64->64 @128": (128, 64, 64), "128->64 @128": (128, 128, 64), "128->128 @64": (64, 128, 128), "256->128 @64": (64, 256, 128), "256->256 @32": (32, 256, 256)}
for name, (H, Ci, Co) in SHAPES.items():
    B = 16
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, H, H, Ci, device="cuda", generator=g)
    om = torch.randn(B, H, H, 32, device="cuda", generator=g) * float(os.environ.get("CP_OM_STD", "2.0"))
    w = torch.randn(Co, Ci, 3, 3, device="cuda", generator=g) / (Ci * 9) ** 0.5
    wp = ops.pack_conv_weight(w)
    sc, sh = ops.fold_bn(Co, None, torch.zeros(Co, device="cuda"))
    outs, t = {}, {}
    for rep in range(5):
        for tl in TILES:
            out = torch.empty(B, H, H, Co, device="cuda")
            fn = lambda: ops.dcn_v2(x, om, wp, sc, sh, out, cout=Co, act=1, tile=tl)
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10):
                fn()
            e1.record(); e1.synchronize()
            t.setdefault(tl, []).append(e0.elapsed_time(e1) / 10)
            outs[tl] = out
    fl = 2.0 * B * H * H * Co * Ci * 9
    d = max((outs[TILES[0]] - outs[tl]).abs().max().item() for tl in TILES)
    print("%-16s" % name, "  ".join("%d: %.4f ms (%.1f TF, %.3f)" % (tl, min(r), fl / min(r) / 1e9, fl / min(r) / 1e9 / 157.3) for tl, r in t.items()), "| max diff %.2e" % d)
