"""From a rocprofv3 kernel trace (rocpd SQLite) of the two-stream replay: per kernel name, the time it runs ALONE on the GPU and the
time it shares it with another kernel, per step.  usage: overlap_report.py results.db"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else [x for x in cols if "name" in x][0]
ev = c.execute("select %s, start, end from kernels order by start" % name).fetchall()
steps = max(1, sum(1 for n, _, _ in ev if "pose_assign" in n))
pts = []
for i, (n, s, e) in enumerate(ev):
    pts.append((s, 1, i)); pts.append((e, 0, i))
pts.sort()
active, alone, shared = set(), collections.Counter(), collections.Counter()
last = None
idle, ngaps, two = 0, 0, 0
for t, kind, i in pts:
    if last is not None and not active and t - last < 200000:      # < 0.2 ms with nothing running: a gap inside a step (launch latency), not the host between steps
        idle += t - last
        ngaps += 1
    if last is not None and len(active) >= 2:
        two += t - last
    if last is not None and active:
        dt = t - last
        if len(active) == 1:
            alone[ev[next(iter(active))][0]] += dt
        else:
            for j in active:
                shared[ev[j][0]] += dt
    if kind: active.add(i)
    else: active.discard(i)
    last = t
short = lambda n: n.replace("void ", "").split("(")[0][:60]
tot_alone = sum(alone.values())
print("%d steps; GPU time with exactly one kernel running: %.3f ms per step; two or more: %.3f ms; NO kernel running inside a step (gaps < 0.2 ms): "
      "%.3f ms per step in %.1f gaps (%.2f us each)" % (steps, tot_alone / steps / 1e6, two / steps / 1e6, idle / steps / 1e6, ngaps / steps,
                                                       idle / max(1, ngaps) / 1e3))
for n in sorted(set(alone) | set(shared), key=lambda n: -alone[n]):
    if (alone[n] + shared[n]) / steps < 5000: continue
    print("  %-60s alone %7.3f ms  shared %7.3f ms per step" % (short(n), alone[n] / steps / 1e6, shared[n] / steps / 1e6))
