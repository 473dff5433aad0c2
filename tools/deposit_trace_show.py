#!/usr/bin/env python3
"""Reads gpurun_out/deposit_trace_<workload>.npz (tools/deposit_trace.py): how long the waves of the deposit kernels
live, against the particles they walk, and how many are alive over the kernel's duration."""
import sys
import numpy as np

d = np.load(sys.argv[1])
for name in ("rows", "units"):
    a = d[name]
    live = a[:, 1] != 0
    a = a[live]
    if len(a) == 0:
        continue
    t0 = a[:, 0].astype(np.int64); t1 = a[:, 1].astype(np.int64)
    recent = t0 > t0.max() - 100000                              # records of earlier launches (rows that wrote none this time) are dropped
    a = a[recent]; t0 = t0[recent]; t1 = t1[recent]
    base = t0.min() if name == "rows" else base
    t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0          # microseconds (100 MHz clock)
    work = a[:, 2] & 0x7fffffff
    dur = t1 - t0
    print(f"== {name}: {len(a)} waves, kernel span {t1.max():.1f} us, wave-time sum {dur.sum() / 1e3:.1f} ms, mean {dur.mean():.2f} us, max {dur.max():.1f} us")
    edges = [0, 1, 65, 129, 257, 513, 1025, 2049, 4097, 8193, 1 << 30]
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (work >= lo) & (work < hi)
        if m.any():
            print(f"   particles [{lo:5d},{hi:10d}): waves {m.sum():7d}  particles {work[m].sum():10d}  mean life {dur[m].mean():8.2f} us  p99 {np.percentile(dur[m], 99):8.2f}  sum {dur[m].sum() / 1e3:8.2f} ms")
    # alive waves over time
    span = t1.max()
    bins = np.linspace(min(0.0, t0.min()), span, 21)
    mids = 0.5 * (bins[:-1] + bins[1:])
    alive = [((t0 <= m) & (t1 > m)).sum() for m in mids]
    started = np.histogram(t0, bins)[0]
    print("   time (us):  ", " ".join(f"{b:6.0f}" for b in bins[:-1]))
    print("   alive waves:", " ".join(f"{v:6d}" for v in alive))
    print("   started:    ", " ".join(f"{v:6d}" for v in started))
    xcc = (a[:, 3] >> 0) & 0xf
