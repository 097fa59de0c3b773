#!/usr/bin/env python3
"""Largest gaps between consecutive kernels in a rocprofv3 rocpd database, with the kernels around them.
    python tools/find_stall.py r_results.db [n]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t.startswith("kernels")][0]
cols = [r[1] for r in con.execute("pragma table_info(%s)" % kt)]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = con.execute("select %s, start, end from %s order by start" % (name_col, kt)).fetchall()
gaps = sorted(((rows[i][1] - rows[i - 1][2], i) for i in range(1, len(rows))), reverse=True)[:n]
t0 = rows[0][1]
for g, i in sorted(gaps, key=lambda x: x[1]):
    print("gap %9.1f us at kernel #%d (t = %.1f ms): %s -> %s" % (g / 1e3, i, (rows[i][1] - t0) / 1e6, rows[i - 1][0][:40], rows[i][0][:40]))
print("tables:", [t for t in tabs if not t.startswith("rocpd_info")][:40])
