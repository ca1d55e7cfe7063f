"""time multi_pose_decode (nms_topk + pose_assign) on a random batch; optional second library for an A/B.
usage: python tools/decode_time.py [B]   (run under tools/with_lib.py OLD.so for the other side)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centerpose_amd.decode import multi_pose_decode
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda c: torch.rand(B, c, 128, 128, device="cuda", generator=g)
heat, hm_hp = torch.sigmoid(4 * r(1) - 3), torch.sigmoid(4 * r(17) - 3)
wh, kps, reg, hpo = r(2) * 20, (r(34) - 0.5) * 30, r(2), r(2)
best = 1e9
for rep in range(5):
    for _ in range(5):
        d = multi_pose_decode(heat, wh, kps, reg, hm_hp, hpo, 100)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        d = multi_pose_decode(heat, wh, kps, reg, hm_hp, hpo, 100)
    e1.record(); e1.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
print("multi_pose_decode B=%d: %.4f ms per call (incl. 3 allocations); checksum %.6f" % (B, best, d.double().sum().item()))
