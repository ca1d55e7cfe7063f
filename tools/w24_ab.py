"""same-process A/B of a conv3x3_wino24_kernel switch (env variable read per launch).  usage: python tools/w24_ab.py ENVVAR v0,v1,..."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centerpose_amd import ops
VAR, VALS = sys.argv[1], sys.argv[2].split(",")
for name, (B, H, W, Ci, Co) in {"c64_128": (16, 128, 128, 64, 64), "c128_64": (16, 64, 64, 128, 128), "c256_32": (16, 32, 32, 256, 256),
                                "om64_128": (16, 128, 128, 64, 32), "c32_128b8": (8, 128, 128, 32, 32)}.items():
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, H, W, Ci, device="cuda", generator=g)
    w = torch.randn(Co, Ci, 3, 3, device="cuda", generator=g) * 0.05
    wp = ops.pack_conv_weight(w)
    u = ops.pack_wino24_weight(wp, Ci, Co)
    sc, sh = ops.fold_bn(Co, None, torch.zeros(Co, device="cuda"))
    res = torch.randn(B, H, W, Co, device="cuda", generator=g)
    outs, t = {}, {}
    for rep in range(4):
        for v in VALS:
            os.environ[VAR] = v
            out = torch.empty(B, H, W, Co, device="cuda")
            la = ops.conv2d_launch([x], wp, sc, sh, out, kh=3, kw=3, stride=1, pad=1, cout=Co, act=1, res=res, tile=ops.WINO24, wino=u)
            for _ in range(10):
                la.run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(30):
                la.run()
            e1.record(); e1.synchronize()
            t.setdefault(v, []).append(e0.elapsed_time(e1) / 30)
            outs[v] = out
    fl = 2.0 * B * H * W * Co * Ci * 9
    same = all(torch.equal(outs[VALS[0]], outs[v]) for v in VALS)
    print(name, "  ".join("%s=%s: %.4f ms (%.1f TF)" % (VAR, v, min(r), fl / min(r) / 1e9) for v, r in t.items()), "| bit-identical" if same else "| DIFFERENT")
