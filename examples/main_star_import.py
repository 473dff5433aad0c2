#!/usr/bin/env python3
"""A main.py-shaped script whose only link to the engine is its first statement: everything below the star
import uses the names the reference's main.py takes from `from functions import *` (main.py:1) -- `time`, `np`,
`import_galaxy`, `CIC` / `NGP`, `compute_potential`, `force_CIC` / `force_NGP`, the two energy functions and
`elapsed_time` -- in the order main.py:20-69 uses them, with np.gradient between the solve and the gather as at
main.py:54.  It shows (and tests/test_gpu_parity.py checks) that swapping that one import is all a main.py needs.

    python examples/main_star_import.py [galaxy1.npz galaxy2.npz]       # STEPS=5 python ... for a short run

Without files it uses the two 5000-particle galaxies under tests/golden/ (written by the reference's own
generators).  Plotting and the viewer (main.py:71-86) are not part of the path and are left out."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nbody-simulation_amd"))

from pm.functions import *      # noqa: F401,F403,E402 -- the one line that differs from main.py

grid_size = 100
box_size = 10000
dt = 0.005
steps = int(os.environ.get("STEPS", "500"))
CIC_interpolation = os.environ.get("NGP", "") == ""

_golden = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
galaxy_files = sys.argv[1:3] if len(sys.argv) >= 3 else [
    os.path.join(_golden, "galaxy_ellipse_importance_1052_n5000.npz"),
    os.path.join(_golden, "galaxy_diffuse_sphere_importance_1052_n5000.npz")]

start_time = time.time()        # noqa: F405

positions, velocities, masses, labels = import_galaxy(      # noqa: F405
    galaxy_files[0], galaxy_files[1], separation=400.0, direction=(0.67, 0.33, 0), velocity=(-3000, 300, 0))
positions += 0.5 * np.asarray(box_size, dtype=float) - positions.mean(axis=0)       # noqa: F405

trajectory = []
energies = []

density_func = CIC if CIC_interpolation else NGP                # noqa: F405
force_func = force_CIC if CIC_interpolation else force_NGP      # noqa: F405

for step in range(steps):
    density = density_func(positions, grid_size, box_size, masses)
    potential = compute_potential(density, grid_size)           # noqa: F405
    potential_gradient = np.gradient(potential)                 # noqa: F405 -- NumPy's own, as at main.py:54
    forces = force_func(potential_gradient, positions, grid_size, box_size)
    velocities += forces * dt
    positions += velocities * dt
    positions = positions.astype(np.float32) % box_size         # noqa: F405
    trajectory.append(positions.copy())
    if step % 10 == 0:
        KE = compute_kinetic_energy(velocities, masses)         # noqa: F405
        PE = compute_potential_energy(positions, masses, potential, grid_size, box_size)    # noqa: F405
        energies.append([KE, PE, KE + PE])

print(f"Main loop completed in {elapsed_time(start_time):.2f} seconds.")     # noqa: F405
release()                                                       # noqa: F405 -- frees the cached engines (not in main.py)
