"""Which construction order keeps two plan instances concurrent?  (bench.py's first pipelined version showed no overlap where
tools/pipeline_try.py showed +26 % on hrnet B=8.)  usage: python tools/pipeline_try2.py [arch] [B]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
arch = sys.argv[1] if len(sys.argv) > 1 else "hrnet"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
iters = 200


def run(engs, streams, n):
    for i in range(4):
        with torch.cuda.stream(streams[i % len(streams)]):
            engs[i % len(engs)].process(engs[i % len(engs)].input)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        e, s = engs[i % len(engs)], streams[i % len(streams)]
        with torch.cuda.stream(s):
            e.process(e.input)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def report(tag, engs, streams):
    t1 = min(run(engs[:1], [torch.cuda.current_stream()], iters) for _ in range(2))
    t2 = min(run(engs, streams, iters) for _ in range(2))
    print("%-70s one %.1f img/s   two in flight %.1f img/s  (x%.3f)" % (tag, B * iters / t1, B * iters / t2, t1 / t2), flush=True)


# V1: the order of tools/pipeline_try.py: e1 built + captured on the default stream, then e2 built, then the streams, e2 captured on its stream
e1 = bench.make_engine(arch, B); e1.process(e1.input); torch.cuda.synchronize()
e2 = bench.make_engine(arch, B)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
report("V1 separate caches, e1 captured first, streams made before e2's capture", [e1, e2], [sa, sb])
del e1, e2; torch.cuda.empty_cache()
# V2: bench.py's first version: both engines built (shared caches), streams made, captures happen lazily on the step streams
cc, sc = {}, {}
e1, e2 = bench.make_engine(arch, B, const_cache=cc, sched_cache=sc), bench.make_engine(arch, B, const_cache=cc, sched_cache=sc)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
report("V2 shared caches, both built, then streams, lazy capture on the step streams", [e1, e2], [sa, sb])
del e1, e2; torch.cuda.empty_cache()
# V3: shared caches, but every engine captured on the default stream BEFORE the step streams exist
cc, sc = {}, {}
e1, e2 = bench.make_engine(arch, B, const_cache=cc, sched_cache=sc), bench.make_engine(arch, B, const_cache=cc, sched_cache=sc)
e1.process(e1.input); e2.process(e2.input); torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
report("V3 shared caches, both captured on the default stream, then the step streams", [e1, e2], [sa, sb])
del e1, e2; torch.cuda.empty_cache()
# V4: separate caches, bench order
e1, e2 = bench.make_engine(arch, B), bench.make_engine(arch, B)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
report("V4 separate caches, both built, then streams, lazy capture on the step streams", [e1, e2], [sa, sb])
del e1, e2; torch.cuda.empty_cache()
# V5: as V3 with high-priority step streams
cc, sc = {}, {}
e1, e2 = bench.make_engine(arch, B, const_cache=cc, sched_cache=sc), bench.make_engine(arch, B, const_cache=cc, sched_cache=sc)
e1.process(e1.input); e2.process(e2.input); torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=-1)
report("V5 as V3, step streams with priority -1", [e1, e2], [sa, sb])
