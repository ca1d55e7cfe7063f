"""Throughput of the C plan runtime (cp_plan_process through ctypes: no engine.py in the loop) next to the Python engine's
one-replay step, same plan file.  usage: cplan_bench.py [arch] [B]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from centerpose_amd import cplan, engine, synth

arch = sys.argv[1] if len(sys.argv) > 1 else "dla_34"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
x = synth.make_images(B).cuda()
eng = engine.Engine(arch, synth.make_state_dict(arch), B, 512, 512, decode_k=100)
for _ in range(5):
    eng.process(x)
want = eng.dets.clone()
path = os.path.join(tempfile.mkdtemp(), "p.cpplan")
eng.save_plan(path)


def rate(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return B * n / (time.perf_counter() - t0)


print("%s B=%d python engine, one replay per step: %.1f img/s" % (arch, B, rate(lambda: eng.process(x))))
del eng
torch.cuda.empty_cache()
cp = cplan.CPlan(path, use_graph=1)
d = cp.process(x, K=100)
torch.cuda.synchronize()
print("C plan runtime bit-identical dets:", bool(torch.equal(d, want)))
print("%s B=%d C plan runtime (cp_plan_process):      %.1f img/s" % (arch, B, rate(lambda: cp.process(x, K=100))))
cp.close()
