"""Experiment (round 5): how much of the step is idle / under-filled GPU that a SECOND batch in flight can fill?
rocprofv3 shows ~0.85 ms per 7.3 ms step with no kernel running (92 gaps between dependent graph nodes) and 4.3 ms with exactly one.
  A  one engine, B images per replay, replays back to back on one stream (what bench.py times)
  B  two engines (own activations, own captured graph), B images each, replays alternating on two streams: two steps in flight
  C  two engines of B/2 images each, replayed side by side on two streams (the same step split in two)
usage: python tools/pipeline_try.py [arch] [B] [iters]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

arch = sys.argv[1] if len(sys.argv) > 1 else "dla_34"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200


def run(engs, streams, n):
    for e in engs:
        e.process(e.input)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        e, s = engs[i % len(engs)], streams[i % len(streams)]
        with torch.cuda.stream(s):
            e.process(e.input)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


from centerpose_amd import synth
e1 = bench.make_engine(arch, B)
e1.input.copy_(synth.make_images(B).cuda())
ref = [t.clone() for t in e1.process(e1.input)[0]] + [e1.dets.clone()]
cur = torch.cuda.current_stream()
tA = min(run([e1], [cur], iters) for _ in range(3))
print("A  one engine   B=%d           : %.1f img/s  %.3f ms per batch" % (B, B * iters / tA, tA / iters * 1e3))
e2 = bench.make_engine(arch, B)
e2.input.copy_(e1.input)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
tB = min(run([e1, e2], [sa, sb], iters) for _ in range(3))
torch.cuda.synchronize()
same = all(torch.equal(a, b) for a, b in zip(ref, list(e2.outputs) + [e2.dets]))
print("B  two engines  B=%d in flight : %.1f img/s  %.3f ms per batch  (x%.3f)  second engine's outputs %s" % (
    B, B * iters / tB, tB / iters * 1e3, tA / tB, "bit-identical" if same else "DIFFER"))
del e2
torch.cuda.empty_cache()
if B % 2 == 0:
    h1, h2 = bench.make_engine(arch, B // 2), bench.make_engine(arch, B // 2)
    tC = min(run([h1, h2], [sa, sb], 2 * iters) for _ in range(3))
    print("C  two engines  B=%d side by side: %.1f img/s  %.3f ms per %d images  (x%.3f)" % (B // 2, B * iters / tC, tC / iters * 1e3, B, tA / tC))
