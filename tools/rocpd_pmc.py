#!/usr/bin/env python3
"""Per-kernel PMC counter averages (per dispatch) from one or more rocprofv3 --pmc rocpd databases.
    python tools/rocpd_pmc.py gpurun_out/pmcA/r_results.db [more.db ...] [--match substr] [--json out.json]"""
import json
import sqlite3
import sys


def main():
    args = sys.argv[1:]
    match, out = None, None
    if "--match" in args:
        i = args.index("--match"); match = args[i + 1]; del args[i:i + 2]
    if "--json" in args:
        i = args.index("--json"); out = args[i + 1]; del args[i:i + 2]
    res = {}
    for db in args:
        con = sqlite3.connect(db)
        q = ("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection "
             "group by kernel_name, counter_name")
        for kn, cn, v, nd in con.execute(q):
            if match and match not in kn:
                continue
            short = kn.split("(")[0].replace("void ", "")
            res.setdefault(short, {})[cn] = v / max(nd, 1)
    text = json.dumps(res, indent=1, sort_keys=True)
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
