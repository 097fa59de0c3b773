#!/bin/bash
# The graphics clock a kernel ran at: GRBM_GUI_ACTIVE (cycles, summed over the 8 XCDs) against the traced duration.
# usage (GPU box): bash tools/clk_probe.sh <match> <program> [args...]     (CLK_JSON=<file>: also as JSON, kernels >= 0.1 ms)
# Under --pmc the kernels of a step run one after the other (no side-stream overlap); kernels of a few microseconds read
# nonsense (the counter's granularity) and are left out of the JSON.
M=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/clk
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d /tmp/clk -o r -- "$@" > /dev/null 2>&1
python3 - "$M" "${CLK_JSON:-}" <<'PY'
import json, sqlite3, sys
con = sqlite3.connect("/tmp/clk/r_results.db")
cyc = {}
for kn, v, nd in con.execute("select kernel_name, sum(value), count(distinct dispatch_id) from counters_collection where counter_name='GRBM_GUI_ACTIVE' group by kernel_name"):
    cyc[kn] = v / max(nd, 1)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
nc = "name" if "name" in cols else [c for c in cols if "name" in c][0]
out = {}
for kn, d, n in con.execute("select %s, avg(end - start), count(*) from kernels group by %s" % (nc, nc)):
    if sys.argv[1] in kn and kn in cyc:
        print("%-40s %8.3f ms  %6.1f Mcycles/XCD  -> %.2f GHz  (%d launches)" % (kn[:40], d / 1e6, cyc[kn] / 8e6, cyc[kn] / 8 / d, n))
        if d >= 1e5:
            out[kn.split("(")[0].replace("void ", "")] = {"ms": d / 1e6, "mcycles_per_xcd": cyc[kn] / 8e6, "ghz": cyc[kn] / 8 / d, "launches": n}
if sys.argv[2]:
    json.dump({"note": "GRBM_GUI_ACTIVE / 8 XCDs / traced duration, kernels serialised by --pmc; peak graphics clock 2.4 GHz",
               "kernels": out}, open(sys.argv[2], "w"), indent=1, sort_keys=True)
PY
