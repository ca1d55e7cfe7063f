"""What would ONE launch for the four n <= 2 heads buy?  The existing F(2x4) head kernel with 1024 mid channels (four heads' 3x3
weights back to back, one head's 1x1 rows: the OUTPUT is meaningless, the work is that of a merged launch) against four launches."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centerpose_amd import ops
B, H, W = 16, 128, 128
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, H, W, 64, device="cuda", generator=g)
def mk(hc):
    w3 = torch.randn(hc, 64, 3, 3, device="cuda", generator=g) / 24.0
    wp3 = ops.pack_conv_weight(w3)
    sc, sh = ops.fold_bn(hc, None, torch.zeros(hc, device="cuda"))
    u = ops.pack_wino24_weight(wp3, 64, hc)
    w1 = (torch.randn(2, hc, device="cuda", generator=g) / 16.0).contiguous()
    b1 = torch.randn(2, device="cuda", generator=g)
    out = torch.empty(B, 2, H, W, device="cuda")
    return ops.head3x3_1x1_launch(x, u, sc, sh, w1, b1, out, hc=hc, act2=0, wino24=True)
one = [mk(256) for _ in range(4)]
merged = mk(1024)
def t(fn):
    best = 1e9
    for rep in range(4):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    return best
a = t(lambda: [l.run() for l in one]); b = t(merged.run); a2 = t(lambda: [l.run() for l in one]); b2 = t(merged.run)
print("four launches of 256 mid channels: %.4f / %.4f ms   one launch of 1024: %.4f / %.4f ms   (%s)" % (a, a2, b, b2, merged.kernel))
