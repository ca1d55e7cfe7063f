"""Per-launch table of the engine schedule (GPU box), every launch timed inside the step (HIP events between consecutive
launches; CP_PROFILE_HOT=1: each launch repeated back to back, cache-hot).  usage: layer_profile.py [arch] [B] > table"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from centerpose_amd import synth, engine
arch = sys.argv[1] if len(sys.argv) > 1 else "dla_34"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
eng = engine.Engine(arch, synth.make_state_dict(arch), B, 512, 512, use_graph=False)
eng.input.copy_(synth.make_images(B).cuda())
recs = eng.profile_in_sequence(iters=10) if os.environ.get('CP_PROFILE_HOT', '0') == '0' else eng.profile(iters=10)
tot = sum(r["ms"] for r in recs)
print("%s B=%d: %d launches, %.3f ms total, %.1f img/s, %.1f TF overall" % (arch, B, len(recs), tot, B / tot * 1e3, sum(r["flops"] for r in recs) / tot / 1e9))
for r in recs:
    print("%-5s %-62s %8.3f ms %7.1f GF %6.1f TF  %4.1f%%  %7.1f MB %5.2f TB/s" % (
        r["kind"], r["name"][-62:], r["ms"], r["flops"] / 1e9, r["flops"] / r["ms"] / 1e9 if r["ms"] else 0, 100 * r["ms"] / tot,
        r["bytes"] / 1e6, r["bytes"] / r["ms"] / 1e9 if r["ms"] else 0))
