"""Drop-in for the force / integration functions of the reference's functions.py.

Same names, argument order and return conventions as the reference (host ndarray in, freshly
allocated host ndarray out), so that `from pm.functions import *` makes a main.py-shaped step
loop run unchanged with every stage on the GPU:

    density   = CIC(positions, grid_size, box_size, masses)          # functions.py:19-70
    potential = compute_potential(density, grid_size)                # functions.py:73-89
    potential_gradient = gradient(potential)                         # np.gradient, main.py:54
    forces    = force_CIC(potential_gradient, positions, grid_size, box_size)   # functions.py:99-180

This granular mode copies every array over PCIe on every call; it exists for parity tests and
plumbing.  The fast path is pm.Simulation, which keeps all state resident.
"""
import time

import numpy as np

from . import _lib as L
from .simulation import Simulation

# `np` and `time` are part of what `from functions import *` hands to the reference's main.py (functions.py:1-2 import
# them at module level and main.py:20, 33, 54, 60 use them through the star import), and so is elapsed_time
# (functions.py:363-364, main.py:69, 86): they are exported here for the same reason.
__all__ = ["np", "time", "CIC", "NGP", "compute_potential", "gradient", "force_CIC", "force_NGP", "kick_drift",
           "compute_kinetic_energy", "compute_potential_energy", "import_galaxy", "elapsed_time", "release"]

_engines = {}


def _engine(grid_size, box_size, n, ngp=False):
    """One cached engine per (grid_size, box_size, branch); regrown when more particles arrive."""
    key = (int(grid_size), float(box_size), bool(ngp))
    sim = _engines.get(key)
    if sim is None or sim.capacity < n:
        if sim is not None:
            sim.close()
        sim = _engines[key] = Simulation(grid_size, box_size, capacity=max(n, 1024), interpolation="NGP" if ngp else "CIC")
    return sim


def release():
    """Free every cached engine (device memory, FFT plans)."""
    for sim in _engines.values():
        sim.close()
    _engines.clear()


def CIC(positions: np.ndarray, grid_size: int, box_size: float, masses: np.ndarray) -> np.ndarray:
    """Cloud-in-cell mass deposition -> (G,G,G) float32, no division by the cell volume."""
    sim = _engine(grid_size, box_size, len(positions))
    sim.set_particles(positions, None, masses)
    sim.deposit()
    return sim.density()


def compute_potential(density: np.ndarray, grid_size: int) -> np.ndarray:
    """FFT Poisson solve: +density_k / k^2 with k in cycles per sample, DC zeroed.  float32 for a float32 density
    (CIC); float64 for a float64 one (NGP), as SciPy's transforms and NumPy's promotion make it in the reference."""
    density = np.asarray(density)
    if density.shape != (grid_size,) * 3:
        raise ValueError(f"density must have shape {(grid_size,) * 3}")
    sim = _engine(grid_size, 1.0, 0)       # the solve does not depend on the box size
    if density.dtype == np.float64:
        sim.set_mesh_f64(L.MESH_DENSITY, density)
        sim.solve_potential_f64()
        return sim.mesh_f64(L.MESH_POTENTIAL)
    sim.set_mesh(L.MESH_DENSITY, density)
    sim.solve_potential()
    return sim.potential()


def gradient(potential: np.ndarray):
    """np.gradient(potential): tuple of three meshes of the potential's dtype, one-sided on the faces."""
    potential = np.asarray(potential)
    g = potential.shape[0]
    sim = _engine(g, 1.0, 0)
    if potential.dtype == np.float64:
        sim.set_mesh_f64(L.MESH_POTENTIAL, potential)
        sim.gradient_f64()
        return tuple(sim.mesh_f64(w) for w in (L.MESH_GX, L.MESH_GY, L.MESH_GZ))
    sim.set_mesh(L.MESH_POTENTIAL, potential)
    sim.gradient()
    return sim.potential_gradient()


def force_CIC(potential_gradient, positions: np.ndarray, grid_size: int, box_size: float) -> np.ndarray:
    """Cloud-in-cell gather of the three gradient meshes -> (N,3) float32."""
    sim = _engine(grid_size, box_size, len(positions))
    for which, a in zip((L.MESH_GX, L.MESH_GY, L.MESH_GZ), potential_gradient):
        sim.set_mesh(which, a)
    sim.set_particles(positions, None, None)
    sim.compute_forces()
    return sim.forces()


def kick_drift(potential_gradient, positions, velocities, grid_size, box_size, dt):
    """force_CIC fused with main.py:58-60; returns (new float32 positions, velocities)."""
    sim = _engine(grid_size, box_size, len(positions))
    for which, a in zip((L.MESH_GX, L.MESH_GY, L.MESH_GZ), potential_gradient):
        sim.set_mesh(which, a)
    sim.set_particles(positions, velocities, None)
    sim.gather_kick_drift(dt)
    return sim.positions(), sim.velocities()


def NGP(positions: np.ndarray, grid_size: int, box_size: float, masses: np.ndarray) -> np.ndarray:
    """Nearest-grid-point mass histogram -> (G,G,G) float64 (functions.py:11-16).  Raises IndexError
    where the reference does: a float32 position that wraps to exactly box_size."""
    sim = _engine(grid_size, box_size, len(positions))
    sim.set_particles(positions, None, masses)
    sim.deposit_ngp()
    return sim.density_f64()


def force_NGP(potential_gradient, positions: np.ndarray, grid_size: int, box_size: float) -> np.ndarray:
    """The gradient meshes at each particle's nearest-grid-point cell -> (N,3) in the meshes' dtype
    (functions.py:92-97): float64 in the NGP branch proper, float32 for float32 meshes."""
    grads = [np.asarray(a) for a in potential_gradient]
    sim = _engine(grid_size, box_size, len(positions))
    if all(a.dtype == np.float64 for a in grads):
        for which, a in zip((L.MESH_GX, L.MESH_GY, L.MESH_GZ), grads):
            sim.set_mesh_f64(which, a)
        sim.set_particles(positions, None, None)
        sim.compute_forces_ngp_f64()
        return sim.forces_f64()
    if any(a.dtype != np.float32 for a in grads):
        raise TypeError("force_NGP takes three float32 or three float64 gradient meshes")
    for which, a in zip((L.MESH_GX, L.MESH_GY, L.MESH_GZ), grads):
        sim.set_mesh(which, a)
    sim.set_particles(positions, None, None)
    sim.compute_forces_ngp()
    return sim.forces()


def compute_kinetic_energy(velocities, masses):
    """0.5 * sum(m * |v|^2) (functions.py:352-354), summed in float64 on the GPU."""
    velocities = np.asarray(velocities)
    sim = _engine(2, 1.0, len(velocities))
    sim.set_particles(np.zeros((len(velocities), 3), np.float32), velocities, masses)
    sim.set_mesh(L.MESH_POTENTIAL, np.zeros((2, 2, 2), np.float32))     # (pm_energies sums both energies and wants a potential for the other)
    return sim.energies()[0]


def compute_potential_energy(positions, masses, potential, grid_size, box_size):
    """0.5 * sum(m * potential[nearest cell]) (functions.py:356-361), summed in float64 on the GPU; the products
    are float64 too when the potential is (the NGP branch)."""
    potential = np.asarray(potential)
    if potential.dtype == np.float64:
        sim = _engine(grid_size, box_size, len(positions), ngp=True)
        sim.set_mesh_f64(L.MESH_POTENTIAL, potential)
        sim.set_particles(positions, None, masses)
        return sim.energies()[1]
    sim = _engine(grid_size, box_size, len(positions))
    sim.set_mesh(L.MESH_POTENTIAL, potential)
    sim.set_particles(positions, None, masses)
    return sim.energies()[1]


def import_galaxy(path1, path2=None, separation=1.0, direction=(1, 0, 0), velocity=(0, 0, 0)):
    """Load one or two galaxy .npz files (keys positions, velocities, masses, labels; float64 on disk)
    as the reference does (functions.py:303-348): float32 on load; the second galaxy is shifted so that
    its centre of mass sits `separation` along `direction` from the first one's, and the relative
    `velocity` is split between the two for zero total momentum.  Host-side set-up: plain NumPy.

    Returns (positions, velocities, masses) for one file, (positions, velocities, masses, labels) for two;
    with two files positions and velocities come out float64, as in the reference."""
    def read(path):
        with np.load(path) as z:
            return [z[k].astype(np.float32) for k in ("positions", "velocities", "masses")] + [z["labels"]]

    xa, va, ma, la = read(path1)
    if path2 is None:
        return xa, va, ma
    xb, vb, mb, lb = read(path2)
    centre = lambda x, m: (m[:, None] * x).sum(0) / m.sum()      # float32 centre of mass
    axis = np.asarray(direction, float)
    axis /= np.linalg.norm(axis)
    xb = xb + ((centre(xa, ma) + axis * float(separation)) - centre(xb, mb))
    total_a, total_b = np.sum(ma), np.sum(mb)
    rel = np.asarray(velocity)
    va = va + np.asarray(-(total_b / (total_a + total_b)) * rel)
    vb = vb + np.asarray((total_a / (total_a + total_b)) * rel)
    return np.vstack((xa, xb)), np.vstack((va, vb)), np.concatenate((ma, mb)), np.concatenate((la, lb))


def elapsed_time(start_time):
    """Seconds since `start_time`, a time.time() stamp (functions.py:363-364)."""
    return time.time() - start_time
