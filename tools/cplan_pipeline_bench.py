"""Throughput of the C plan handle (cp_plan_process) and of the C pipeline (cp_pipeline_process: `depth` plan instances in one hipGraph,
captured by csrc/plan_runtime.cpp with its fixed per-instance stream placement) next to the Python entry points of the same plan.
usage: cplan_pipeline_bench.py [arch] [B] [depth] [steps]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from centerpose_amd import cplan

arch = sys.argv[1] if len(sys.argv) > 1 else "dla_34"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
D = int(sys.argv[3]) if len(sys.argv) > 3 else 2
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 60
det = bench.make_detector(arch)
eng = bench.make_engine(arch, B, det=det)
batches = [x.cuda() for x in bench.make_batches(B, D)]
py1 = bench.timed_stream(det, batches, steps, 10, 1)
pyD = bench.timed_stream(det, batches, steps, 10, D)
path = os.path.join(tempfile.mkdtemp(), "p.cpplan")
eng.save_plan(path)
plan = cplan.CPlan(path)
pipe = cplan.CPipeline(plan, depth=D)


def timed(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


c1 = timed(lambda: plan.process(batches[0], 100), steps)
cD = timed(lambda: pipe.process(batches, 100), steps // D)
a = eng.process(batches[0])[1].clone()
b = pipe.process(batches, 100)[0]
torch.cuda.synchronize()
print("%s B=%d: python process %.1f img/s, python process_stream(depth=%d) %.1f | C cp_plan_process %.1f, C cp_pipeline_process(depth=%d) %.1f img/s; "
      "C pipeline == python engine bits: %s" % (arch, B, B * steps / py1, D, B * steps / pyD, B * steps / c1, D, B * (steps // D) * D / cD, torch.equal(a, b)))
