#!/usr/bin/env python3
"""Rates that include the host <-> device copies of the boundary (not the headline number): upload of the particle
arrays, one resident step, download of positions and velocities, and the granular (host array in / host array out)
calls of pm.functions on the headline workload."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nbody-simulation_amd")]
import bench
import pm
from pm import functions as F

n, G, L, dt = 16_777_216, 512, 10000, 0.005
pos, vel, mass = bench.make_workload("two_galaxy", n, L, 1052)
sim = pm.Simulation(G, L, pos, vel, mass, fused_gradient=True)
sim.step(dt, 2)
sim.synchronize()
out = {}
t = time.perf_counter(); sim.set_particles(pos, vel, mass); sim.synchronize(); out["upload_ms"] = (time.perf_counter() - t) * 1e3
t = time.perf_counter(); sim.step(dt, 1); sim.synchronize(); out["step_ms"] = (time.perf_counter() - t) * 1e3
t = time.perf_counter(); x = sim.positions(); v = sim.velocities(); out["download_ms"] = (time.perf_counter() - t) * 1e3
out["updates_per_s_upload_step_download"] = n / ((out["upload_ms"] + out["step_ms"] + out["download_ms"]) * 1e-3)
sim.close()
t = time.perf_counter()
d = F.CIC(pos, G, L, mass)
p = F.compute_potential(d, G)
g = F.gradient(p)
f = F.force_CIC(g, pos, G, L)
out["granular_step_ms"] = (time.perf_counter() - t) * 1e3
out["updates_per_s_granular"] = n / (out["granular_step_ms"] * 1e-3)
print(json.dumps(out))
