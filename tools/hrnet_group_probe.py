"""HRNet-W32 B = 8 branch convolutions (pose_higher_hrnet.py:217-222), in-graph times of the grouped F(2x4) launch and of what a
split-C of its LONG members would cost, EMULATED with existing launches before anything is built: a member split S ways over its input
channels is S x the blocks with 1/S of the stages each and S x the output writes -- the same block population as an unsplit member
with C / S input channels at S x the batch (the last-arriver reduction of a real split is not in the emulation).
usage: hrnet_group_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from centerpose_amd import ops
from basicblock_probe import graph_time


def member(B, HW, Ci, Co, g):
    x = torch.randn(B, HW, HW, Ci, device="cuda", generator=g)
    w = torch.randn(Co, Ci, 3, 3, device="cuda", generator=g) * 0.05
    wp = ops.pack_conv_weight(w)
    sc, sh = ops.fold_bn(Co, None, torch.zeros(Co, device="cuda"))
    return dict(x=x, wp=wp, u24=ops.pack_wino24_weight(wp, Ci, Co), scale=sc, shift=sh, cout=Co, act=ops.ACT_RELU, res=None, n=B * HW * HW * Co)


def group(ms):
    whole = torch.empty(sum(m["n"] for m in ms), device="cuda")
    off, recs = 0, []
    for m in ms:
        B, H, W, _ = m["x"].shape
        r = dict(m)
        r["out"] = whole[off:off + m["n"]].view(B, H, W, m["cout"])
        off += m["n"]
        r.pop("n")
        recs.append(r)
    return ops.conv3x3_group_launch(recs, whole)


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    m32, m64, m128, m256 = member(8, 128, 32, 32, g), member(8, 64, 64, 64, g), member(8, 32, 128, 128, g), member(8, 16, 256, 256, g)
    s128_2 = member(16, 32, 64, 128, g)          # 128 ch split 2: 256 blocks x 4 stages
    s256_4 = member(32, 16, 64, 256, g)          # 256 ch split 4: 256 blocks x 4 stages
    s256_2 = member(16, 16, 128, 256, g)         # 256 ch split 2: 128 blocks x 8 stages
    cases = [("single 32@128", [m32]), ("single 64@64", [m64]), ("single 128@32", [m128]), ("single 256@16", [m256]),
             ("stage2 group (32, 64)", [m32, m64]),
             ("stage3 group (32, 64, 128)", [m32, m64, m128]),
             ("stage3 group, 128 split 2 (emulated)", [m32, m64, s128_2]),
             ("stage4 group (32, 64, 128, 256)", [m32, m64, m128, m256]),
             ("stage4 group, 256 split 2 (emulated)", [m32, m64, m128, s256_2]),
             ("stage4 group, 256 split 4 (emulated)", [m32, m64, m128, s256_4]),
             ("stage4 group, 256 split 4 + 128 split 2 (emulated)", [m32, m64, s128_2, s256_4]),
             ("pair (128, 256)", [m128, m256]), ("pair (128 split 2, 256 split 4)", [s128_2, s256_4])]
    for rep in range(2):
        for name, ms in cases:
            print("run %d  %-52s %6.1f us" % (rep, name, graph_time([group(ms)])), flush=True)


main()
