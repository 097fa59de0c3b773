#!/usr/bin/env python3
"""Step periods of a captured-step run in a rocprofv3 rocpd database: the start-to-start distance of a marker kernel (default
k_pack_stage1), averaged over blocks of steps, with the kernel time and the largest gap inside a step.
    python tools/step_periods.py r_results.db [marker] [block]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "k_pack_stage1"
block = int(sys.argv[3]) if len(sys.argv) > 3 else 100
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t.startswith("kernels")][0] if any(t.startswith("kernels") for t in tabs) else "kernels"
cols = [r[1] for r in con.execute("pragma table_info(%s)" % kt)]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = con.execute("select %s, start, end from %s order by start" % (name_col, kt)).fetchall()
steps, cur = [], None
for nm, s, e in rows:
    if nm.startswith(marker) or (" " + marker) in nm:
        if cur is not None:
            steps.append(cur)
        cur = {"t0": s, "busy": 0, "gap": 0, "gapname": "", "prev": None}
    if cur is None:
        continue
    cur["busy"] += e - s
    if cur["prev"] is not None and s - cur["prev"] > cur["gap"]:
        cur["gap"], cur["gapname"] = s - cur["prev"], nm[:28]
    cur["prev"] = e
print("%d steps" % len(steps))
for i in range(0, len(steps) - 1, block):
    blk = steps[i:i + block + 1]
    if len(blk) < 2:
        break
    per = (blk[-1]["t0"] - blk[0]["t0"]) / (len(blk) - 1) / 1e3
    busy = sum(b["busy"] for b in blk[:-1]) / (len(blk) - 1) / 1e3
    gap = sum(b["gap"] for b in blk[:-1]) / (len(blk) - 1) / 1e3
    names = {}
    for b in blk[:-1]:
        names[b["gapname"]] = names.get(b["gapname"], 0) + 1
    print("steps %5d..%5d: period %7.1f us, kernels %7.1f us, largest in-step gap %6.1f us before %s"
          % (i, i + len(blk) - 2, per, busy, gap, max(names, key=names.get)))
