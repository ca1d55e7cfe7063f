"""Experiment (GPU box): N batches in flight.  N engines with their own activation buffers (weights shared through const_cache)
replay their step graphs on N launch streams round-robin, so the tail / stalls of one step's kernels are filled by the next
step's.  usage: inflight_try.py [arch] [B] [N ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from centerpose_amd import engine, synth

arch = sys.argv[1] if len(sys.argv) > 1 else "dla_34"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
Ns = [int(x) for x in sys.argv[3:]] or [1, 2, 3]
sd = synth.make_state_dict(arch)
cc, sc = {}, {}
x = synth.make_images(B, seed=317).cuda()
for N in Ns:
    engs = [engine.Engine(arch, sd, B, 512, 512, decode_k=100, const_cache=cc, sched_cache=sc) for _ in range(N)]
    streams = [torch.cuda.Stream() for _ in range(N)]
    for e in engs:
        e.input.copy_(x)
        e.process(e.input)
    torch.cuda.synchronize()
    ref = engs[0].dets.clone()
    def run(steps):
        for i in range(steps):
            k = i % N
            with torch.cuda.stream(streams[k]):
                engs[k].process(engs[k].input)
    run(20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(120)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = all(torch.equal(e.dets, ref) for e in engs)
    print("%s B=%d  %d in flight: %.3f ms/step  %.1f img/s  dets identical %s" % (arch, B, N, dt / 120 * 1e3, B * 120 / dt, same), flush=True)
    del engs
