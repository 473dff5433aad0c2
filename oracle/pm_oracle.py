"""TEST INFRASTRUCTURE -- CPU oracle for the PM step.  Not part of the product.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may
import this module; the product path (nbody-simulation_amd/) never does and
fails loudly when its HIP library is missing.

Parity status: PINNED.  Every function is checked against tests/golden/*.npz
(outputs of the reference's own functions, written by tools/make_golden.py)
in tests/test_oracle_golden.py: bit for bit for the particle loops, the
gradient, the Poisson solve and the integrator.

The two particle loops (numba-compiled in the reference) are restated in C
(pm_oracle.c -> liboracle.so); the vectorised stages are restated with
numpy / scipy.fft, the same third-party arithmetic the reference calls
(no version is pinned by the reference; here numpy 2.2 / scipy 1.15).
"""
import ctypes
import os
import subprocess

import numpy as np
from scipy import fft as _sfft

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


_i64 = ctypes.c_int64
_f32 = ctypes.c_float
_f64 = ctypes.c_double


def _pos64(positions):
    """Positions widened to float64 (value preserving): numba's typing of the loops."""
    return np.ascontiguousarray(positions, dtype=np.float64)


def cell_index(positions, grid_size, box_size):
    """(N,3) int32 cell of every particle -- functions.py:27-39."""
    p = _pos64(positions)
    out = np.empty((len(p), 3), dtype=np.int32)
    _lib().oracle_cell_index(_p(p, _f64), _i64(len(p)), _i64(grid_size), _i64(int(box_size)), _p(out, ctypes.c_int32))
    return out


def cell_keys(positions, grid_size, box_size):
    """Row-major cell key (i*G + j)*G + k of every particle, int64."""
    c = cell_index(positions, grid_size, box_size).astype(np.int64)
    return (c[:, 0] * grid_size + c[:, 1]) * grid_size + c[:, 2]


def sort_order(keys):
    """Cell order of the particles: stable sort by key of the current storage order."""
    return np.argsort(keys, kind="stable")


def CIC(positions, grid_size, box_size, masses):
    """functions.py:19-70.  Sequential float32 accumulation in particle order."""
    p = _pos64(positions)
    m = np.ascontiguousarray(masses, dtype=np.float32)
    d = np.empty((grid_size,) * 3, dtype=np.float32)
    _lib().oracle_cic(_p(p, _f64), _p(m, _f32), _i64(len(p)), _i64(grid_size), _i64(int(box_size)), _p(d, _f32))
    return d


def CIC_f64(positions, grid_size, box_size, masses):
    """The same deposit accumulated in float64 (accuracy yardstick, not a reference output)."""
    p = _pos64(positions)
    m = np.ascontiguousarray(masses, dtype=np.float32)
    d = np.empty((grid_size,) * 3, dtype=np.float64)
    _lib().oracle_cic_f64(_p(p, _f64), _p(m, _f32), _i64(len(p)), _i64(grid_size), _i64(int(box_size)), _p(d, _f64))
    return d


def softening_coefficient(length, grid_size, box_size):
    """Coefficient c of exp(-c k2) for a Gaussian softening length in box units (pm_set_softening; not in the reference):
    exp(-(length k_phys)^2 / 2), k_phys = 2 pi k G / L, rounded to float32 as the library holds it."""
    a = 2.0 * np.pi * float(length) * grid_size / float(box_size)
    return np.float32(0.5 * a * a)


def compute_potential(density, grid_size, soft=0.0):
    """functions.py:73-89.  k in cycles/sample, +delta_k/k^2, DC zeroed, full complex64 FFT.  soft != 0 (an extension,
    see softening_coefficient) multiplies the spectrum by exp(-soft k2) as well."""
    n = grid_size
    j = np.arange(n)
    j = np.where(j < (n + 1) // 2, j, j - n)
    k = (j * (1.0 / n)).astype(np.float32)                  # np.fft.fftfreq(n) as float32 (:75)
    kk = k * k
    k2 = (kk[:, None, None] + kk[None, :, None]) + kk[None, None, :]   # :76-77, same add order
    k2[0, 0, 0] = 1                                         # :78
    mean = np.mean(density)                                 # :81
    dk = _sfft.fftn(density - mean)                         # :82
    pk = dk / k2                                            # :85
    if soft:
        if pk.dtype == np.complex64:
            pk = pk * np.exp(-np.float32(soft) * k2)
        else:
            pk = pk * np.exp(-np.float64(np.float32(soft)) * k2.astype(np.float64))
    pk[0, 0, 0] = 0.0                                       # :86
    return np.real(_sfft.ifftn(pk))                         # :89


def gradient(potential):
    """np.gradient(potential) as called at main.py:54 -- tuple of three float32 meshes."""
    f = np.ascontiguousarray(potential, dtype=np.float32)
    g = f.shape[0]
    out = [np.empty_like(f) for _ in range(3)]
    _lib().oracle_gradient(_p(f, _f32), _i64(g), _p(out[0], _f32), _p(out[1], _f32), _p(out[2], _f32))
    return tuple(out)


def force_CIC(potential_gradient, positions, grid_size, box_size):
    """functions.py:99-180."""
    p = _pos64(positions)
    g = [np.ascontiguousarray(a, dtype=np.float32) for a in potential_gradient]
    f = np.empty((len(p), 3), dtype=np.float32)
    _lib().oracle_force_cic(_p(g[0], _f32), _p(g[1], _f32), _p(g[2], _f32), _p(p, _f64), _i64(len(p)),
                            _i64(grid_size), _i64(int(box_size)), _p(f, _f32))
    return f


def kick_drift(positions, velocities, forces, dt, box_size):
    """main.py:58-60.  Returns (new float32 positions, velocities in their own dtype)."""
    f = np.ascontiguousarray(forces, dtype=np.float32)
    n3 = f.size
    if velocities.dtype == np.float32 and positions.dtype == np.float32:
        x = np.ascontiguousarray(positions).copy()
        v = np.ascontiguousarray(velocities).copy()
        _lib().oracle_kick_drift_f32(_p(x, _f32), _p(v, _f32), _p(f, _f32), _i64(n3), _f64(dt), _i64(int(box_size)))
        return x, v
    xin = _pos64(positions)
    v = np.ascontiguousarray(velocities, dtype=np.float64).copy()
    x = np.empty(xin.shape, dtype=np.float32)
    _lib().oracle_kick_drift_f64v(_p(xin, _f64), _p(x, _f32), _p(v, _f64), _p(f, _f32), _i64(n3), _f64(dt),
                                  _i64(int(box_size)))
    return x, v


def pm_step(positions, velocities, masses, grid_size, box_size, dt, keep=False):
    """One pass of main.py:50-60.  With keep=True also returns the intermediates."""
    density = CIC(positions, grid_size, box_size, masses)
    potential = compute_potential(density, grid_size)
    grad = gradient(potential)
    forces = force_CIC(grad, positions, grid_size, box_size)
    x, v = kick_drift(positions, velocities, forces, dt, box_size)
    if keep:
        return x, v, dict(density=density, potential=potential, grad=grad, forces=forces)
    return x, v


# ---- the float64 build mode: main.py:50-60 with float64 in place of float32 everywhere ------------
# (the reference has no such switch; this is the float64 CPU restatement that the library's PM_FLAG_F64 mode is held
# to at 1e-12)

def cell_and_fraction_f64(positions, grid_size, box_size):
    """functions.py:27-43 in float64: cell indices (n,3) int64 and fractions (n,3)."""
    x = np.remainder(np.asarray(positions, np.float64), float(box_size))
    s = x * (grid_size / float(box_size))
    i = s.astype(np.int64)
    d = s - i
    i[i >= grid_size] -= grid_size
    return i, d


def force_CIC_f64(potential_gradient, positions, grid_size, box_size):
    """functions.py:99-180 with float64 gradient meshes and a float64 forces array: the eight corners in the
    reference's order, w = wx * wy * wz, forces += grad * w."""
    G = grid_size
    c, d = cell_and_fraction_f64(positions, G, box_size)
    c1 = (c + 1) % G
    w = np.stack((1.0 - d, d))                          # [corner bit, particle, axis]
    idx = np.stack((c, c1))
    f = np.zeros((len(c), 3), np.float64)
    for di, dj, dk in ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1)):
        wt = w[di, :, 0] * w[dj, :, 1] * w[dk, :, 2]
        node = (idx[di, :, 0], idx[dj, :, 1], idx[dk, :, 2])
        for a in range(3):
            f[:, a] = f[:, a] + potential_gradient[a][node] * wt
    return f


def pm_step_f64(positions, velocities, masses, grid_size, box_size, dt, keep=False):
    """One pass of main.py:50-60 with a float64 mesh (functions.py:22), float64 forces (:102) and no cast at main.py:60."""
    x = np.asarray(positions, np.float64)
    v = np.asarray(velocities, np.float64)
    density = CIC_f64(x, grid_size, box_size, masses)
    potential = compute_potential(density, grid_size)
    grad = np.gradient(potential)
    forces = force_CIC_f64(grad, x, grid_size, box_size)
    v = v + forces * dt
    x = np.remainder(x + v * dt, float(box_size))
    if keep:
        return x, v, dict(density=density, potential=potential, grad=grad, forces=forces)
    return x, v


# ---- secondary variants and diagnostics (numpy restatements) -------------------------------

def NGP(positions, grid_size, box_size, masses):
    """functions.py:11-16: nearest-grid-point histogram, float64 mesh, index in the positions' dtype."""
    density = np.zeros((grid_size,) * 3, dtype=np.float64)
    w = positions % box_size
    idx = np.floor(w / box_size * grid_size).astype(np.int64)
    np.add.at(density, (idx[:, 0], idx[:, 1], idx[:, 2]), masses)
    return density


def ngp_index(positions, grid_size, box_size):
    """functions.py:94 and :358: (pos / L * G).astype(int) % G."""
    return (positions / box_size * grid_size).astype(np.int64) % grid_size


def force_NGP(potential_gradient, positions, grid_size, box_size):
    """functions.py:92-97."""
    i = ngp_index(positions, grid_size, box_size)
    return np.stack([g[i[:, 0], i[:, 1], i[:, 2]] for g in potential_gradient], axis=1)


def pm_step_ngp(positions, velocities, masses, grid_size, box_size, dt, keep=False):
    """One pass of main.py:50-60 with CIC_interpolation = False (main.py:9, 40-41): NGP's float64 mesh makes the
    potential, its gradient and the forces float64.  The kick / drift / wrap lines are main.py:58-60 verbatim in
    NumPy, so their dtype rules are NumPy's own."""
    density = NGP(positions, grid_size, box_size, masses)
    potential = compute_potential(density, grid_size)
    grad = np.gradient(potential)
    forces = force_NGP(grad, positions, grid_size, box_size)
    velocities = velocities.copy()
    positions = positions.copy()
    velocities += forces * dt
    positions += velocities * dt
    positions = positions.astype(np.float32) % box_size
    if keep:
        return positions, velocities, dict(density=density, potential=potential, grad=grad, forces=forces)
    return positions, velocities


def compute_kinetic_energy(velocities, masses):
    """functions.py:352-354."""
    return 0.5 * np.sum(masses * np.sum(velocities ** 2, axis=1))


def compute_potential_energy(positions, masses, potential, grid_size, box_size):
    """functions.py:356-361."""
    i = ngp_index(positions, grid_size, box_size)
    return np.sum(masses * potential[i[:, 0], i[:, 1], i[:, 2]]) * 0.5


def import_galaxy(path1, path2=None, separation=1.0, direction=(1, 0, 0), velocity=(0, 0, 0)):
    """functions.py:303-348 (without its stray print)."""
    def load(path):
        with np.load(path) as z:
            return (z["positions"].astype(np.float32), z["velocities"].astype(np.float32),
                    z["masses"].astype(np.float32), z["labels"])
    x1, v1, m1, l1 = load(path1)
    if path2 is None:
        return x1, v1, m1
    x2, v2, m2, l2 = load(path2)
    com1 = (m1[:, None] * x1).sum(0) / m1.sum()
    com2 = (m2[:, None] * x2).sum(0) / m2.sum()
    d = np.asarray(direction, float)
    d /= np.linalg.norm(d)
    x2 = x2 + ((com1 + d * float(separation)) - com2)
    M1, M2 = np.sum(m1), np.sum(m2)
    V1 = -(M2 / (M1 + M2)) * np.asarray(velocity)
    V2 = (M1 / (M1 + M2)) * np.asarray(velocity)
    v1 = v1 + np.asarray(V1)
    v2 = v2 + np.asarray(V2)
    return np.vstack((x1, x2)), np.vstack((v1, v2)), np.concatenate((m1, m2)), np.concatenate((l1, l2))
