#!/usr/bin/env python3
"""Kernel timeline of the LAST step in a rocprofv3 rocpd database: name, start offset, duration, gap to the previous
kernel (all us).    python tools/timeline.py r_results.db [n_kernels]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = con.execute("select %s, start, end from kernels order by start desc limit %d" % (name_col, n)).fetchall()[::-1]
t0, prev = rows[0][1], None
for nm, s, e in rows:
    print("%-48s start %9.1f  dur %8.1f  gap %7.1f" % (nm[:48], (s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3))
    prev = e
