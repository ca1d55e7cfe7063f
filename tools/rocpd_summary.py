"""Summarise a rocprofv3 (rocpd SQLite) kernel trace: per-kernel calls / total / average duration.
usage: rocpd_summary.py results.db [out.md]"""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                     "group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    tot = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, cnt, s, a, mn, mx in rows:
        n = n if len(n) < 110 else n[:107] + "..."
        lines.append("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f |" % (n, cnt, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    lines.append("")
    lines.append("total kernel time: %.3f ms over %d dispatches" % (tot / 1e6, sum(r[1] for r in rows)))
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(*sys.argv[1:3])
