#!/usr/bin/env python3
"""Per-kernel average durations (us) of every gpurun_out/pv_<variant>/r_results.db side by side, for kernels matching the patterns given."""
import glob
import os
import sqlite3
import sys

pats = sys.argv[1:] or ["k_"]
dbs = sorted(glob.glob("gpurun_out/pv_*/r_results.db"))
table = {}
for path in dbs:
    var = os.path.basename(os.path.dirname(path))[3:]
    db = sqlite3.connect(path)
    for name, cnt, avg in db.execute("select name, count(*), avg(end - start) from kernels group by name"):
        if any(p in name for p in pats):
            short = name.replace("(anonymous namespace)::", "").replace("void ", "")[:46]
            table.setdefault(short, {})[var] = avg / 1e3
names = [os.path.basename(os.path.dirname(p))[3:] for p in dbs]
print(f"{'kernel':46s} " + " ".join(f"{n:>9s}" for n in names))
for k, row in sorted(table.items()):
    print(f"{k:46s} " + " ".join(f"{row.get(n, float('nan')):9.1f}" for n in names))
for n in names:
    log = f"gpurun_out/pv_{n}.log"
    if os.path.exists(log):
        import json
        for line in open(log):
            if line.startswith("{"):
                print(n, "ms_per_step", round(json.loads(line)["ms_per_step"], 4))
