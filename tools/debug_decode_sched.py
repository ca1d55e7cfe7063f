import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from centerpose_amd import engine, synth
from centerpose_amd.decode import multi_pose_decode
arch, B, hw = "res_50", 1, 128
sd = synth.make_state_dict(arch)
x = synth.make_images(B, hw, hw, seed=3).cuda()
for trial in range(4):
    eng = engine.Engine(arch, sd, B, hw, hw, decode_k=100, sigmoid_heads=("hm", "hm_hp"))
    for rep in range(3):
        outs, dets = eng.process(x)
        torch.cuda.synchronize()
        want = multi_pose_decode(outs[0], outs[1], outs[2], reg=outs[3], hm_hp=outs[4], hp_offset=outs[5], K=100)
        ok = torch.equal(dets, want)
        print("trial", trial, "rep", rep, "dets equal:", ok, "max diff", float((dets - want).abs().max()))
    deps = eng.dependencies()
    n = len(eng.launches)
    for i in range(n):
        k, name, _, l = eng.launches[i]
        if i >= n - 16:
            print("  %3d s%d %-40s deps %s" % (i, eng.stream_plan[i], name[-40:], [(j, eng.stream_plan[j]) for j in deps[i]]))
    eng.run_eager(); torch.cuda.synchronize()
    print("  eager dets equal:", torch.equal(eng.dets, want))
