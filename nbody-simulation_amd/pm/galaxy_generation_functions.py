"""The reference's importance-sampled galaxy generators with its names and signatures
(galaxy_generation_functions.py:181-244, 267-383, 386-394, 454-491), computed on the GPU by libpm.so
(pm_generate_galaxy): host ndarrays out, float64 like the reference's.

`stream="numpy"` (the default) draws the standard normal deviates on the host from np.random.default_rng(seed) in the
order the reference consumes them, so a seed names the same galaxy as in the reference (positions bit for bit; masses
and the diffuse sphere's velocities to the rounding of sums taken in another order).  `stream="device"` draws them on
the GPU from a counter-based generator instead: the same distributions, another stream, no host arrays -- for sets of
10^7 particles and more.

Not mirrored: view_configuration (matplotlib), the equal-mass / importance disk samplers (plain NumPy one-liners with no
reduction in them), and the reference's `generate_galaxy("disk", ...)` branch, which passes `seed` where `Rmax` is
expected and saves an undefined `labels` (galaxy_generation_functions.py:479-481, 490).
"""
import ctypes
import os

import numpy as np

from . import _lib

STAR, DARK = "1", "2"                     # Type.Star.value, Type.Dark.value (galaxy_generation_functions.py:6-8)
ELLIPSE, DIFFUSE_SPHERE = 0, 1


def _generate(kind, n, params, seed, normals, device):
    lib = _lib.load()
    n = int(n)
    if n < 1:
        raise ValueError("N_particles must be at least 1")
    pos, vel, mass = np.empty((n, 3)), np.empty((n, 3)), np.empty(n)
    par = np.asarray(params, dtype=np.float64)
    nrm = None if normals is None else np.ascontiguousarray(normals, dtype=np.float64)
    rc = lib.pm_generate_galaxy(int(device), kind, n, _lib.ptr(par), len(par), ctypes.c_uint64((0 if seed is None else int(seed)) & (2 ** 64 - 1)),
                                None if nrm is None else _lib.ptr(nrm), _lib.ptr(pos), _lib.ptr(vel), _lib.ptr(mass))
    _lib.check(None, rc)
    return pos, vel, mass


def generate_ellipse_importance(N_particles, a, b, c, M_tot=1.0, seed=None, *, disp=0.1, virial_seed=None, stream="numpy", device=0):
    """galaxy_generation_functions.py:181-244.  Returns (positions, velocities, masses, labels).  The reference draws
    the in-plane velocity dispersion from an unseeded generator (:242 calls virial_ellipsoid without a seed);
    `virial_seed` makes that part repeatable."""
    n = int(N_particles)
    normals = None
    if stream == "numpy":
        rng = np.random.default_rng(seed)
        z = rng.standard_normal((n, 3))
        rng2 = np.random.default_rng(virial_seed)
        normals = np.concatenate((z.ravel(), rng2.standard_normal(n), rng2.standard_normal(n)))
    elif stream != "device":
        raise ValueError(f"Unknown stream: {stream!r}")
    pos, vel, mass = _generate(ELLIPSE, n, [a, b, c, M_tot, disp], seed, normals, device)
    return pos, vel, mass, np.array([STAR] * n, dtype="U5")


def generate_diffuse_sphere_importance(N_particles, R, M_tot, seed=None, *, stellar_fraction=0.25, R_dm=390.0, stream="numpy", device=0):
    """galaxy_generation_functions.py:267-383.  Returns (positions, velocities, masses, labels): the first
    int(0.5 N) particles are stars (Plummer weights), the rest dark matter (NFW weights)."""
    n = int(N_particles)
    n_star = int(0.5 * n)
    normals = None
    if stream == "numpy":
        rng = np.random.default_rng(seed)
        star = rng.standard_normal((n_star, 3))
        dark = rng.standard_normal((n - n_star, 3))
        q = rng.standard_normal((n, 3))
        normals = np.concatenate((star.ravel(), dark.ravel(), q.ravel()))
    elif stream != "device":
        raise ValueError(f"Unknown stream: {stream!r}")
    pos, vel, mass = _generate(DIFFUSE_SPHERE, n, [R, M_tot, stellar_fraction, R_dm], seed, normals, device)
    return pos, vel, mass, np.array([STAR] * n_star + [DARK] * (n - n_star), dtype="U5")


def save_galaxy_npz(path, positions, masses, labels, velocities=None):
    """galaxy_generation_functions.py:386-394: the on-disk format import_galaxy reads."""
    if velocities is None:
        velocities = np.zeros_like(positions)
    np.savez(file=path, positions=positions, velocities=velocities, masses=masses, labels=labels)


def generate_galaxy(morphology, sampling="importance", N_particles=10000, Rd=None, R=None, z0=None, a=None, b=None, c=None,
                    M_tot=1.0, dm_factor=0.0, seed=None, *, out_dir="Outputs", **kw):
    """galaxy_generation_functions.py:454-491 for the two morphologies that have a working branch there: builds the
    galaxy and saves it as `{out_dir}/{morphology}_{sampling}_{seed}.npz` (the name main.py:26-27 loads).  Returns the
    path (the reference returns nothing and opens a plot)."""
    if morphology == "ellipse":
        positions, velocities, masses, labels = generate_ellipse_importance(N_particles, a, b, c, M_tot, seed, **kw)
    elif morphology == "diffuse_sphere":
        positions, velocities, masses, labels = generate_diffuse_sphere_importance(N_particles, R, M_tot, seed, **kw)
    elif morphology == "disk":
        raise ValueError("the reference's disk branch does not run (it passes seed as Rmax and saves undefined labels); "
                         "use Simulation.generate('disk', ...) for an exponential disk")
    else:
        raise ValueError(f"Unknown morphology: {morphology!r}")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, f"{morphology}_{sampling}_{seed}.npz")
    save_galaxy_npz(path, positions, masses, labels, velocities)
    return path
