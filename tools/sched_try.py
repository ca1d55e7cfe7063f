"""Try launch orders / stream placements for the two-stream hipGraph capture (GPU box).  usage: sched_try.py [arch] [B]
The data-dependency DAG (Engine.dependencies() of the emission order) is fixed; every candidate is a topological order of
it plus a stream per launch; outputs must stay bit-identical."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from centerpose_amd import synth, engine

arch = sys.argv[1] if len(sys.argv) > 1 else "dla_34"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
eng = engine.Engine(arch, synth.make_state_dict(arch), B, 512, 512, use_graph=False)
eng.input.copy_(synth.make_images(B).cuda())
recs = eng.profile_in_sequence(iters=5)
dur = [r["ms"] for r in recs]
orig = list(eng.launches)
deps0 = eng.dependencies()
n = len(orig)
eng.run_eager(); torch.cuda.synchronize()
ref = [o.clone() for o in eng.outputs]
print("%s B=%d: %d launches, %.3f ms in sequence" % (arch, B, n, sum(dur)))

children = [[] for _ in range(n)]
for i, d in enumerate(deps0):
    for j in d:
        children[j].append(i)
blevel = [0.0] * n
for i in reversed(range(n)):
    blevel[i] = dur[i] + max([blevel[c] for c in children[i]], default=0.0)
print("critical path %.3f ms" % max(blevel))


def list_schedule(prio, nproc=2, side_ok=lambda i: True):
    """classic list scheduling; returns (order, assign)"""
    indeg = [len(d) for d in deps0]
    finish = [0.0] * n
    ready = [i for i in range(n) if indeg[i] == 0]
    free = [0.0] * nproc
    order, assign, start = [], [0] * n, [0.0] * n
    done = 0
    while done < n:
        # earliest time any (proc, ready task) pair can start
        best = None
        for i in ready:
            est = max([finish[j] for j in deps0[i]], default=0.0)
            for p in range(nproc):
                if p > 0 and not side_ok(i):
                    continue
                t = max(est, free[p])
                key = (t, -prio[i], p)
                if best is None or key < best[0]:
                    best = (key, i, p, t)
        _, i, p, t = best
        ready.remove(i)
        start[i], finish[i], free[p], assign[i] = t, t + dur[i], t + dur[i], p
        order.append(i)
        done += 1
        for c in children[i]:
            indeg[c] -= 1
            if indeg[c] == 0:
                ready.append(c)
    order.sort(key=lambda i: (start[i], i))
    # a topological order is required: sort by start keeps it (a child never starts before its parents finish)
    return order, assign, max(finish)


def capture_and_time(order, assign, tag):
    pos = {i: k for k, i in enumerate(order)}
    eng.launches = [orig[i] for i in order]
    deps = [sorted(pos[j] for j in deps0[i]) for i in order]
    for k, d in enumerate(deps):
        assert all(j < k for j in d), "not a topological order"
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        eng._run_branches(s, 2, deps, None if assign is None else [assign[i] for i in order])
    torch.cuda.synchronize()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(eng.outputs, ref))
    ts = []
    for _ in range(40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    side = sum(1 for x in eng.stream_of_launch if x)
    print("%-34s median %.3f ms  p10 %.3f  (%d launches on the side stream, bit-identical %s)" % (tag, ts[len(ts) // 2], ts[4], side, same))
    return g


keep = []
NP = int(os.environ.get("NPROC", "2"))
keep.append(capture_and_time(list(range(n)), None, "emission order, greedy placement (2 streams)"))
o, a, mk = list_schedule(blevel)
keep.append(capture_and_time(o, a, "list schedule (b-level), 2 streams, sim %.3f" % mk))
def time_dag(tag, g):
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    same = all(torch.equal(x, y) for x, y in zip(eng.outputs, ref))
    ts = []
    for _ in range(40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort()
    print("%-60s median %.3f ms p10 %.3f bit-identical %s" % (tag, ts[20], ts[4], same), flush=True)

for npr in (2, 3, 4, 6):
    o, a, mk = list_schedule(blevel, nproc=npr)
    pos = {i: k for k, i in enumerate(o)}
    eng.launches = [orig[i] for i in o]
    deps = [sorted(pos[j] for j in deps0[i]) for i in o]
    asg = [a[i] for i in o]
    # explicit graph whose edges are what an npr-stream capture would record: the stream predecessor + the cross-stream dependencies
    last = {}
    edges = []
    for k in range(n):
        e = set(j for j in deps[k] if asg[j] != asg[k])
        if asg[k] in last:
            e.add(last[asg[k]])
        last[asg[k]] = k
        edges.append(sorted(e))
    keep.append(eng.build_dag_graph(edges))
    time_dag("explicit graph, chains of a %d-stream schedule, sim %.3f" % (npr, mk), keep[-1])
keep.append(eng.build_dag_graph())
time_dag("explicit graph, full DAG (transitively reduced), last order", keep[-1])
if NP > 200:

    o, a, mk = list_schedule(blevel, nproc=NP)
    pos = {i: k for k, i in enumerate(o)}
    eng.launches = [orig[i] for i in o]
    deps = [sorted(pos[j] for j in deps0[i]) for i in o]
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    print("capturing on %d streams ..." % NP, flush=True)
    with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
        eng._run_branches(st, NP, deps, [a[i] for i in o])
    torch.cuda.synchronize()
    print("captured", flush=True)
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    same = all(torch.equal(x, y) for x, y in zip(eng.outputs, ref))
    ts = []
    for _ in range(40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort()
    print("list schedule, %d streams, sim %.3f: median %.3f ms p10 %.3f bit-identical %s" % (NP, mk, ts[20], ts[4], same))
