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
import numpy as np

from . import _lib as L
from .simulation import Simulation

__all__ = ["CIC", "compute_potential", "gradient", "force_CIC", "kick_drift", "release"]

_engines = {}


def _engine(grid_size, box_size, n):
    """One cached engine per (grid_size, box_size); regrown when more particles arrive."""
    key = (int(grid_size), float(box_size))
    sim = _engines.get(key)
    if sim is None or sim.capacity < n:
        if sim is not None:
            sim.close()
        sim = _engines[key] = Simulation(grid_size, box_size, capacity=max(n, 1024))
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
    """FFT Poisson solve: +density_k / k^2 with k in cycles per sample, DC zeroed -> float32."""
    density = np.asarray(density)
    if density.shape != (grid_size,) * 3:
        raise ValueError(f"density must have shape {(grid_size,) * 3}")
    sim = _engine(grid_size, 1.0, 0)       # the solve does not depend on the box size
    sim.set_mesh(L.MESH_DENSITY, density)
    sim.solve_potential()
    return sim.potential()


def gradient(potential: np.ndarray):
    """np.gradient(potential): tuple of three float32 meshes, one-sided on the faces."""
    potential = np.asarray(potential)
    g = potential.shape[0]
    sim = _engine(g, 1.0, 0)
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
