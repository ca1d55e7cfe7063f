"""Which PAIRS of streams let two captured plan instances overlap?  usage: python tools/pipeline_try3.py [arch] [B]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
arch = sys.argv[1] if len(sys.argv) > 1 else "hrnet"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cc, sc = {}, {}
e1, e2 = bench.make_engine(arch, B, const_cache=cc, sched_cache=sc), bench.make_engine(arch, B, const_cache=cc, sched_cache=sc)
e1.process(e1.input); e2.process(e2.input); torch.cuda.synchronize()
S = [torch.cuda.Stream() for _ in range(10)]
print("streams:", [hex(s.cuda_stream)[-6:] for s in S])


def run(engs, streams, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        with torch.cuda.stream(streams[i % len(streams)]):
            engs[i % len(engs)].process(engs[i % len(engs)].input)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


run([e1, e2], [S[0], S[1]], 10)
t1 = run([e1], [torch.cuda.current_stream()], 100)
print("one engine, default stream: %.1f img/s" % (B * 100 / t1))
for i in range(len(S)):
    row = []
    for j in range(len(S)):
        if j <= i:
            row.append("   .  ")
            continue
        t = run([e1, e2], [S[i], S[j]], 60)
        row.append("%6.0f" % (B * 60 / t))
    print("S%d " % i + " ".join(row), flush=True)
# the default stream + one other
for j in range(4):
    t = run([e1, e2], [torch.cuda.current_stream(), S[j]], 60)
    print("default + S%d: %.0f img/s" % (j, B * 60 / t))
