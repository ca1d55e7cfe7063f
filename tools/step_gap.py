"""idle time between consecutive graph replays: end of the step's last kernel (pose_assign_kernel) -> start of the next step's first
kernel (the stem), from a rocprofv3 kernel trace (rocpd SQLite).  usage: step_gap.py results.db"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else [x for x in cols if "name" in x][0]
ev = c.execute("select %s, start, end from kernels order by start" % name).fetchall()
gaps, tails = [], []
for i, (n, s, e) in enumerate(ev):
    if "pose_assign" in n:
        nxt = [x for x in ev[i + 1:i + 4] if "stem" in x[0]]
        if nxt and nxt[0][1] - e < 200000:
            gaps.append(nxt[0][1] - e)
        prev = [x for x in ev[max(0, i - 3):i] if "nms_topk" in x[0]]
        if prev:
            tails.append(e - prev[0][1])
if gaps:
    gaps.sort(); tails.sort()
    print("graph replay -> next replay: %d gaps, median %.2f us, p10 %.2f, p90 %.2f" % (len(gaps), gaps[len(gaps) // 2] / 1e3, gaps[len(gaps) // 10] / 1e3, gaps[len(gaps) * 9 // 10] / 1e3))
    print("decode tail (nms_topk start -> pose_assign end): median %.2f us" % (tails[len(tails) // 2] / 1e3))
