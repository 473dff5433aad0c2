#!/usr/bin/env python3
"""Rate of the main loop WITH its per-step trajectory copies (main.py:61) on the headline workload: Simulation.run
(snapshots copied beside the next step, page-locked or pageable host array) against stepping and downloading by hand.
Not the headline number: that one keeps the state in HBM."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nbody-simulation_amd")]
import bench
import pm

n, G, L, dt, K = 16_777_216, 512, 10000, 0.005, 16
pos, vel, mass = bench.make_workload("two_galaxy", n, L, 1052)
out = {"n": n, "G": G, "steps": K}
with pm.Simulation(G, L, pos, vel, mass, fused_gradient=True) as sim:
    sim.step(dt, 2)
    sim.synchronize()
    t = time.perf_counter(); sim.step(dt, K); sim.synchronize(); out["resident_ms_per_step"] = (time.perf_counter() - t) * 1e3 / K
    t = time.perf_counter()
    for _ in range(4):
        sim.step(dt)
        x = sim.positions()
    out["step_then_download_ms_per_step"] = (time.perf_counter() - t) * 1e3 / 4
    for pinned in (True, False):
        t = time.perf_counter()
        traj, en = sim.run(K, dt, trajectory=True, energy_every=10, pinned=pinned)
        total = (time.perf_counter() - t) * 1e3
        # the allocation of the host array is part of the call; time a second run into the same kind of memory too
        out[f"run_{'pinned' if pinned else 'pageable'}_ms_per_step_with_alloc"] = total / K
        del traj
    print(json.dumps(out), flush=True)
    # allocation apart: the C entry point on an array that exists already
    from pm import _lib as Lb
    traj = Lb.pinned_empty((K, n, 3), np.float32)
    ke, pe = np.zeros(2), np.zeros(2)
    t = time.perf_counter()
    sim._ck(sim.lib.pm_run(sim.ctx, dt, K, Lb.ptr(traj), 10, Lb.ptr(ke), Lb.ptr(pe)))
    out["run_pinned_ms_per_step"] = (time.perf_counter() - t) * 1e3 / K
    out["updates_per_s_run_pinned"] = n / (out["run_pinned_ms_per_step"] * 1e-3)
print(json.dumps(out))
