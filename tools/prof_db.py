#!/usr/bin/env python3
"""Per-kernel summary (calls, average / total duration) of a rocprofv3 results database."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else cols[0]
rows = db.execute(f"select {name}, count(*), avg(end - start), sum(end - start) from kernels group by {name} order by 4 desc").fetchall()
tot = sum(r[3] for r in rows)
print(f"{'kernel':90s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>9s} {'%':>6s}")
for n, c, a, t in rows[:top]:
    print(f"{n[:90]:90s} {c:6d} {a/1e3:10.1f} {t/1e6:9.2f} {100*t/tot:6.2f}")
