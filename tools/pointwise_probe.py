"""How far are the HBM-bound 1x1 convolutions of res_50 / hrnet layer1 (msra_resnet.py Bottleneck conv3 64 -> 256 + residual, conv1 256 -> 64,
downsample 64 -> 256 at 128 x 128, B = 8) from what a streaming kernel reaches on this chip?  In-graph times of the launches as they run
today next to an elementwise kernel with the same (or more) HBM traffic, and of conv_pointwise.hip's kernel (tile code 1) that came out of
it.  profiles/r6_pointwise_probe.txt was written by the version of this script that also had the 128-row / K = 128 variants and the
no-MFMA / no-store ablations of the kernel (template switches since removed).  usage: pointwise_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from centerpose_amd import ops
from basicblock_probe import graph_time

B, H, W = 8, 128, 128
g = torch.Generator(device="cuda").manual_seed(0)


def conv1x1(ci, co, res, tile=0, hw=None, b=None):
    hh = ww = hw or H
    bb = b or B
    x = torch.randn(bb, hh, ww, ci, device="cuda", generator=g)
    w = torch.randn(co, ci, 1, 1, device="cuda", generator=g) * 0.05
    sc, sh = ops.fold_bn(co, None, torch.zeros(co, device="cuda"))
    out = torch.empty(bb, hh, ww, co, device="cuda")
    r = torch.randn(bb, hh, ww, co, device="cuda", generator=g) if res else None
    mb = 4 * (x.numel() + out.numel() + (r.numel() if res else 0)) / 1e6
    return ops.conv2d_launch([x], ops.pack_conv_weight(w), sc, sh, out, kh=1, kw=1, cout=co, act=ops.ACT_RELU, res=r, tile=tile), mb


def stream(c, add):
    x = torch.randn(B, H, W, c, device="cuda", generator=g)
    se = torch.ones(B, 1, 1, c, device="cuda")
    a = torch.randn(B, H, W, c, device="cuda", generator=g) if add else None
    out = torch.empty(B, H, W, c, device="cuda")
    return ops.scale_add_launch(x, se, a, out), 4 * (x.numel() * (3 if add else 2)) / 1e6


GEN = 128064
for name, (l, mb) in (("conv3 64 -> 256 + residual + ReLU   generic", conv1x1(64, 256, True, GEN)), ("conv3 64 -> 256 + residual + ReLU   pointwise", conv1x1(64, 256, True, 1)), 
                      ("downsample 64 -> 256               generic", conv1x1(64, 256, False, GEN)), ("downsample 64 -> 256               pointwise", conv1x1(64, 256, False, 1)), 
                      ("conv1 64 -> 64                     generic", conv1x1(64, 64, False, GEN)), ("conv1 64 -> 64                     pointwise", conv1x1(64, 64, False, 1)),
                      ("conv3 128 -> 512 + res @64x64      generic", conv1x1(128, 512, True, 64064, 64)), ("conv3 128 -> 512 + res @64x64      generic 128", conv1x1(128, 512, True, GEN, 64)),
                      
                      ("128 -> 128 @64x64                  generic", conv1x1(128, 128, False, 64064, 64)), 
                      ("conv3 64 -> 256 + res, B = 16      generic", conv1x1(64, 256, True, GEN, None, 16)), ("conv3 64 -> 256 + res, B = 16      pointwise", conv1x1(64, 256, True, 1, None, 16)),
                      ("64 -> 256 + res @32x32 (256 blocks) generic", conv1x1(64, 256, True, 64064, 32)), ("64 -> 256 + res @32x32 (256 blocks) pointwise", conv1x1(64, 256, True, 1, 32)),
                      ("conv1 256 -> 64", conv1x1(256, 64, False)),
                      ("elementwise x * s + a -> out, 256 ch (402 MB)", stream(256, True)), ("elementwise x * s -> out, 256 ch (268 MB)", stream(256, False))):
    for rep in range(2):
        t = graph_time([l])
        print("%-52s %6.1f us  %6.1f MB  %5.2f TB/s  kernel %s" % (name, t, mb, mb / t, l.kernel), flush=True)
