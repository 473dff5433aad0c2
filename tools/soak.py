#!/usr/bin/env python3
"""Long run of the headline workload: ms/step per chunk of steps as the two galaxies fall through each other, with the
state checked after every chunk (finite, inside the box, total mass on the mesh)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nbody-simulation_amd")]
import bench
import pm

# soak.py [chunks [steps_per_chunk [workload [n [G]]]]]
arg = sys.argv[1:] + [None] * 5
chunks, per = int(arg[0] or 8), int(arg[1] or 500)
workload, n, G, L, dt = arg[2] or "two_galaxy", int(arg[3] or 16_777_216), int(arg[4] or 512), 10000, 0.005
pos, vel, mass = bench.make_workload(workload, n, L, 1052)
total = float(mass.astype(np.float64).sum())
with pm.Simulation(G, L, pos, vel, mass, fused_gradient=True) as sim:
    sim.step(dt, 3)
    for c in range(chunks):
        sim.synchronize()
        t = time.perf_counter()
        sim.step(dt, per)
        sim.synchronize()
        ms = (time.perf_counter() - t) * 1e3 / per
        x = sim.positions()
        sim.deposit()
        d = float(sim.density().astype(np.float64).sum())
        ke, pe = sim.energies()
        ok = bool(np.isfinite(x).all() and x.min() >= 0.0 and x.max() <= L and abs(d - total) <= 1e-6 * total)
        print(json.dumps({"steps_done": 3 + (c + 1) * per, "ms_per_step": round(ms, 4), "ok": ok, "mass_on_mesh": d,
                          "x_spread": [float(x[:, 0].std()), float(x[:, 1].std())], "KE": ke, "PE": pe}), flush=True)
        assert ok
