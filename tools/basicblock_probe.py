"""VERDICT r5 #2, measured before building: what a fused HRNet BasicBlock kernel (conv3x3 + BN + ReLU -> conv3x3 + BN + residual + ReLU in
ONE launch, the intermediate tile with its 1-pixel halo recomputed and kept in LDS) can cost at best, from the kernels that exist.

A fused F(2x4) block owns one MFMA M-tile = 8 x 4 Winograd tiles = 16 x 16 pixels of the INTERMEDIATE, i.e. 14 x 14 output pixels:
ceil(128 / 14)^2 = 100 blocks per 128 x 128 image instead of 64 (1.56 x the blocks, each doing conv1 AND conv2 on a full M-tile).
  pair512  : the two launches it would replace, as they run today (B = 8, 32 -> 32 @128x128: 512 blocks each), inside one hipGraph
  one800   : ONE conv launch with the fused kernel's block count and per-block conv1 work (B = 8, 160 x 160 map = 800 blocks of 16 x 16):
             patch load + 2 stages + epilogue -- the fused kernel does all of that (its epilogue goes to LDS instead of HBM) and then
             a second main loop + epilogue on top
  pair800  : two of those = an upper bound of the fused kernel's time (it saves one HBM round trip and one launch boundary of this)
go / no-go: fused >= 1.2 x faster than pair512 needs fused <= pair512 / 1.2; the fused time lies between one800 and pair800.
usage: basicblock_probe.py [C] [map] [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from centerpose_amd import ops


def graph_time(launches, reps=20, iters=30):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for l in launches:
            l.run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                for l in launches:
                    l.run()
        for _ in range(5):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            g.replay()
        e1.record()
        e1.synchronize()
    return e0.elapsed_time(e1) / iters / reps * 1e3          # us per pass of `launches`


def convs(B, H, W, C, n):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, H, W, C, device="cuda", generator=g)
    outs = [torch.empty(B, H, W, C, device="cuda") for _ in range(2)]
    ls = []
    for i in range(n):
        w = torch.randn(C, C, 3, 3, device="cuda", generator=g) * 0.05
        wp = ops.pack_conv_weight(w)
        u = ops.pack_wino24_weight(wp, C, C)
        sc, sh = ops.fold_bn(C, None, torch.zeros(C, device="cuda"))
        src = x if i == 0 else outs[(i - 1) % 2]
        ls.append(ops.conv2d_launch([src], wp, sc, sh, outs[i % 2], kh=3, kw=3, stride=1, pad=1, cout=C, act=ops.ACT_RELU,
                                    res=x if i % 2 == 1 else None, wino=u, tile=ops.WINO24))
    return ls


def main():
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    HW = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    t14 = -(-HW // 14)
    big = 16 * t14                                             # a map with the fused kernel's block count at 16 x 16 pixels per block
    nt = (C + 31) // 32
    print("C=%d map %dx%d B=%d: unfused %d blocks per launch, fused %d spatial tiles (x %d channel tiles of work each)" %
          (C, HW, HW, B, B * (-(-HW // 16)) ** 2 * nt, B * t14 * t14, nt))
    for rep in range(3):
        pair = graph_time(convs(B, HW, HW, C, 2))
        one = graph_time(convs(B, HW, HW, C, 1))
        one_big = graph_time(convs(B, big, big, C, 1))
        pair_big = graph_time(convs(B, big, big, C, 2))
        print("  run %d: pair512 %.1f us (single %.1f)   one800 %.1f us   pair800 %.1f us   -> fused must be <= %.1f us for 1.2x; "
              "bounds [%.1f, %.1f]" % (rep, pair, one, one_big, pair_big, pair / 1.2, one_big, pair_big))


if __name__ == "__main__":
    main()
