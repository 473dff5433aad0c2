import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nbody-simulation_amd"), os.path.join(ROOT, "tests")]
import torch
import pm
from pm.distributed import SlabSimulation, HipSlabEngine
import test_gpu_slab as T
case = sys.argv[1] if len(sys.argv) > 1 else "g32"
c = T.CASES[case]
G, L = c["G"], c["L"]
pos, vel, mass = T.make(c["n"], L, 3, c["crowd"])
sim = pm.Simulation(G, L, pos, vel, mass, fused_gradient=True)
sim.deposit(); d0 = sim.density().copy()
sim.solve_potential(); p0 = sim.potential().copy()
ss = SlabSimulation(G, L, pos, vel, mass, np.arange(len(pos), dtype=np.uint32), capacity=len(pos))
e = ss.engine
e.deposit()
ghost = e.planes_out(0, G, 1)
e.planes_in(0, 0, ghost, add=True)
d1 = e.planes_out(0, 0, 3).cpu().numpy()
print("density planes 0..2 max err", np.abs(d1 - d0[:3]).max(), np.abs(d0).max())
dall = np.concatenate([e.planes_out(0, i, 1).cpu().numpy() for i in range(G)])
print("density all max err", np.abs(dall - d0).max(), "argmax plane", np.unravel_index(np.abs(dall - d0).argmax(), d0.shape))
ss2 = SlabSimulation(G, L, pos, vel, mass, np.arange(len(pos), dtype=np.uint32), capacity=len(pos))
ss2.step(0.0)
pall = np.concatenate([ss2.engine.planes_out(1, 1 + i, 1).cpu().numpy() for i in range(G)])
print("potential max err", np.abs(pall - p0).max(), np.abs(p0).max(), np.unravel_index(np.abs(pall - p0).argmax(), p0.shape))
lo = ss2.engine.planes_out(1, 0, 1).cpu().numpy()[0]; up = ss2.engine.planes_out(1, G + 1, 2).cpu().numpy()
print("halo err", np.abs(lo - pall[G - 1]).max(), np.abs(up - pall[:2]).max())
# one real step
sim2 = pm.Simulation(G, L, pos, vel, mass, fused_gradient=True); sim2.step(c["dt"], 1)
ss3 = SlabSimulation(G, L, pos, vel, mass, np.arange(len(pos), dtype=np.uint32), capacity=len(pos)); ss3.step(c["dt"], 1)
i, x, v = ss3.particles()
xr, vr = sim2.positions(), sim2.velocities()
dv = np.abs(v - vr[i]); bad = np.argsort(dv.max(1))[-5:]
print("v err max", dv.max(), "of", np.abs(vr).max())
for b in bad: print(i[b], pos[i[b]], dv[b], (pos[i[b]] * G / L))
