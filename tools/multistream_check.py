"""A/B of the multi-stream hipGraph capture (GPU box): usage multistream_check.py arch batch [streams,...]"""
import os, sys, time, faulthandler, torch
faulthandler.enable()
sys.path.insert(0, "/root/repo")
from centerpose_amd import engine, synth
arch, B = sys.argv[1], int(sys.argv[2])
sd = synth.make_state_dict(arch)
HW = int(os.environ.get('CP_HW', '512'))
x = synth.make_images(B, HW, HW).cuda()
def run(ns):
    e = engine.Engine(arch, sd, B, HW, HW)
    e.nstreams = ns
    out = [t.clone() for t in e(x)]
    for _ in range(5): e(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): e(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    return out, dt, e
ref, t1, e1 = run(1)
print('single stream ok %.3f ms' % (t1 * 1e3), flush=True)
for ns in [int(v) for v in (sys.argv[3].split(',') if len(sys.argv) > 3 else ['2'])]:
    print('streams', ns, flush=True)
    o, t, e = run(ns)
    same = all(torch.equal(a, b) for a, b in zip(ref, o))
    from collections import Counter
    print(arch, B, "streams", ns, "%.3f ms (1 stream %.3f ms)  bit-identical %s  launches per stream %s" % (t * 1e3, t1 * 1e3, same, sorted(Counter(e.stream_of_launch).items())))
