"""Micro-benchmark of single conv / dcn launches (GPU box).
usage: bench_conv.py name[,name...] [tile,...]   names: see CASES; tile -(MT*10+NT) = Winograd kernel variant (-11, -12, -21), -24 = F(2x4,3x3)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from centerpose_amd import ops

CASES = {  # name: (B, H, W, Cin, Cout, k, stride, pad, kind)
    "c64_128": (16, 128, 128, 64, 64, 3, 1, 1, "conv"),
    "c32_128": (8, 128, 128, 32, 32, 3, 1, 1, "conv"),          # HRNet-W32 high-resolution branch at its per-GPU batch
    "c64_64b8": (8, 64, 64, 64, 64, 3, 1, 1, "conv"),
    "c128_32b8": (8, 32, 32, 128, 128, 3, 1, 1, "conv"),
    "c256_16b8": (8, 16, 16, 256, 256, 3, 1, 1, "conv"),
    "c128_64": (16, 64, 64, 128, 128, 3, 1, 1, "conv"),
    "c256_32": (16, 32, 32, 256, 256, 3, 1, 1, "conv"),
    "c512_16": (16, 16, 16, 512, 512, 3, 1, 1, "conv"),
    "head": (16, 128, 128, 64, 1536, 3, 1, 1, "conv"),
    "om64_128": (16, 128, 128, 64, 27, 3, 1, 1, "conv"),
    "om512_16": (16, 16, 16, 512, 27, 3, 1, 1, "conv"),
    "om128_64": (16, 64, 64, 128, 27, 3, 1, 1, "conv"),
    "om256_32": (16, 32, 32, 256, 27, 3, 1, 1, "conv"),
    "l0": (16, 512, 512, 16, 16, 3, 1, 1, "conv"),
    "l1": (16, 512, 512, 16, 32, 3, 2, 1, "conv"),
    "stem": (16, 512, 512, 3, 16, 7, 1, 3, "stem"),
    "p64_256": (16, 128, 128, 64, 256, 1, 1, 0, "conv"),
    "p256_64": (16, 128, 128, 256, 64, 1, 1, 0, "conv"),
    "p512_128": (16, 64, 64, 512, 128, 1, 1, 0, "conv"),
    "p1024_256": (16, 32, 32, 1024, 256, 1, 1, 0, "conv"),
    "p2048_512": (16, 16, 16, 2048, 512, 1, 1, 0, "conv"),
    "s2_128": (16, 128, 128, 128, 128, 3, 2, 1, "conv"),
    "d64_128": (16, 128, 128, 64, 64, 3, 1, 1, "dcn"),
    "d128_64": (16, 64, 64, 128, 128, 3, 1, 1, "dcn"),
    "d512_16": (16, 16, 16, 512, 256, 3, 1, 1, "dcn"),
}


def run(name, tile=0, iters=20):
    B, H, W, Ci, Co, k, s, p, kind = CASES[name]
    g = torch.Generator(device="cuda").manual_seed(0)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    w = torch.randn(Co, Ci, k, k, device="cuda", generator=g) * 0.05
    wp = ops.pack_conv_weight(w, stem=(kind == "stem"))
    sc, sh = ops.fold_bn(Co, None, torch.zeros(Co, device="cuda"))
    co_store = wp.shape[0] if Co == 27 else Co
    out = torch.empty(B, Ho, Wo, co_store, device="cuda")
    if kind == "stem":
        x = torch.randn(B, Ci, H, W, device="cuda", generator=g)
        fn = lambda: ops.conv2d([x], wp, sc, sh, out, kh=k, kw=k, stride=s, pad=p, cout=Co, act=1, in_nchw=True, tile=tile)
    elif kind == "conv":
        x = torch.randn(B, H, W, Ci, device="cuda", generator=g)
        u = (ops.pack_wino24_weight if tile == -24 else ops.pack_wino_weight)(wp, Ci, co_store) if tile < 0 else None      # -24: F(2x4,3x3)
        fn = lambda: ops.conv2d([x], wp, sc, sh, out, kh=k, kw=k, stride=s, pad=p, cout=co_store, act=1 + int(os.environ.get('CP_ABL', '0')), tile=abs(tile) if u is not None else tile, wino=u)
    else:
        x = torch.randn(B, H, W, Ci, device="cuda", generator=g)
        om = torch.randn(B, H, W, 32, device="cuda", generator=g) * float(os.environ.get("CP_OM_STD", "1.5"))
        fn = lambda: ops.dcn_v2(x, om, wp, sc, sh, out, cout=Co, act=1, tile=tile)
    for _ in range(15):      # the first launches of a process run ~10 % slow (clock ramp): warm up well before timing
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * B * Ho * Wo * Co * Ci * k * k
    print("%-10s tile=%-7d %8.3f ms  %6.1f TF" % (name, tile, ms, fl / ms / 1e9), flush=True)


if __name__ == "__main__":
    names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(CASES)
    tiles = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
    for n in names:
        for t in tiles:
            try:
                run(n, t)
            except Exception as e:
                print(n, t, "ERR", str(e)[:100])
