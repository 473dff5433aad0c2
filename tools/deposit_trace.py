#!/usr/bin/env python3
"""Wave timeline of the deposit kernels (a tuning build with -DPM_TRACE, named by PM_LIBRARY): the headline workload is
stepped a few times, then one more step is run and the per-wave (start, end, particles, hardware id) records of its
deposit are written to gpurun_out/deposit_trace_<workload>.npz.  tools/deposit_trace_show.py reads them."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nbody-simulation_amd")):
    sys.path.insert(0, p)
import bench  # noqa: E402
import pm  # noqa: E402
from pm import _lib  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "two_galaxy"
n, G = 16777216, 512
pos, vel, mass = bench.make_workload(workload, n, bench.BOX, 1052)
sim = pm.Simulation(G, bench.BOX, pos, vel, mass, fused_gradient=True)
sim.step(bench.DT, 5)
sim.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
rows = 1 << 19
buf = np.zeros((2, rows, 4), np.uint32)
lib.pm_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = lib.pm_debug_trace(buf.ctypes.data, buf.nbytes)
assert rc == 0, rc
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"deposit_trace_{os.environ.get('PM_TAG', '')}{workload}.npz"), rows=buf[0, :G * G], units=buf[1, :8192])
print("trace written")
