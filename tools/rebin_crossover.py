#!/usr/bin/env python3
"""Bin stage of the uniform set of the headline size as a function of the share of particles that change their cell per
step (set by the velocity dispersion; light particles, so that it stays put): for choosing the share up to which a
binning starts from the order before.  Run it with two builds (PM_LIBRARY): the default one and one that always takes
that way (-DPM_REBIN_SHARE=1).

    python tools/rebin_crossover.py --sigmas 10,100,200,400,800
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nbody-simulation_amd")):
    sys.path.insert(0, p)

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 24)
ap.add_argument("--grid", type=int, default=512)
ap.add_argument("--sigmas", default="10,100,200,400,800")
ap.add_argument("--full-sort", action="store_true")
args = ap.parse_args()

import pm  # noqa: E402

L, DT = 10000.0, 0.005
rng = np.random.default_rng(5)
pos = (rng.random((args.n, 3)) * L).astype(np.float32)
unit = rng.normal(0, 1, (args.n, 3)).astype(np.float32)
mass = np.full(args.n, 1e-6, np.float32)
for sigma in [float(x) for x in args.sigmas.split(",")]:
    with pm.Simulation(args.grid, L, pos, unit * np.float32(sigma), mass, fused_gradient=True, timing=True, full_sort=args.full_sort) as sim:
        sim.step(DT, 1)
        sim.synchronize()
        sim.step(DT, 3)
        sim.synchronize()
        sim.stage_times()
        sim.step(DT, 10)
        sim.synchronize()
        st = sim.stage_times()
        inc, movers = sim.last_binning()
        print(json.dumps({"sigma": sigma, "from_order_before": inc, "movers_share": round(movers / args.n, 4),
                          "bin_ms": round(st["bin"][0] / st["bin"][1], 4), "gather_ms": round(st["gather"][0] / st["gather"][1], 4)}), flush=True)
