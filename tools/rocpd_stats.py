#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average / min / max duration) from a rocprofv3 rocpd sqlite database.
    python tools/rocpd_stats.py gpurun_out/profNN/r_results.db [out.csv]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = con.execute(
        "select %s, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
        "from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    tot = float(sum(r[2] for r in rows)) or 1.0
    lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
    for n, c, t, a, mn, mx in rows:
        lines.append('"%s",%d,%d,%.1f,%.2f,%d,%d' % (n, c, t, a, 100.0 * t / tot, mn, mx))
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
