#!/usr/bin/env python3
"""Per-kernel averages of the counters in a rocprofv3 counter_collection csv: tools/pmc_summary.py <dir>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"][:70]
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[name].add(r["Dispatch_Id"])
names = sorted({c for v in acc.values() for c in v})
print("kernel".ljust(70), "calls", *[n[-14:].rjust(14) for n in names])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get(names[0], 0)):
    n = len(cnt[k])
    print(k.ljust(70), f"{n:5d}", *[f"{v.get(c, 0) / n:14.4g}" for c in names])
