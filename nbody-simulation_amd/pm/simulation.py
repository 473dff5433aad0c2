"""Resident-state PM engine: the step loop of the reference (main.py:44-67) with all state on
the GPU.  Thin object layer over the C ABI of libpm.so."""
import numpy as np

from . import _lib as L


def ic_parameters(kind, box_size, **over):
    """Parameter list of pm_generate for a kind, defaults as fractions of the box size, overridden by name."""
    c = 0.5 * box_size
    if kind == "uniform":
        d = {"sigma_v": 10.0}
    elif kind == "two_galaxy":
        u = np.array([0.67, 0.33, 0.0])
        u /= np.linalg.norm(u)
        d = {"a1": 0.02 * box_size, "a2": 0.05 * box_size, "rmax": 0.45 * box_size, "centre1": c - 0.1 * box_size * u,
             "centre2": c + 0.1 * box_size * u, "relative_velocity": (-3000.0, 300.0, 0.0), "sigma_v": 10.0}
    elif kind == "disk":
        d = {"Rd": 0.03 * box_size, "z0": 0.003 * box_size, "Rmax": None, "zmax": None, "centre": (c, c, c), "sigma_v": 10.0}
    else:
        raise ValueError(f"kind must be 'uniform', 'two_galaxy' or 'disk', got {kind!r}")
    unknown = set(over) - set(d)
    if unknown:
        raise TypeError(f"{kind}: no parameter named {sorted(unknown)}; it takes {list(d)}")
    d.update(over)
    if kind == "disk":          # galaxy_generation_functions.py:107-108
        if d["Rmax"] is None:
            d["Rmax"] = 5.0 * d["Rd"]
        if d["zmax"] is None:
            d["zmax"] = 5.0 * d["z0"]
    out = []
    for v in d.values():
        out.extend(np.ravel(np.asarray(v, dtype=np.float64)).tolist())
    return out


class Simulation:
    """grid_size / box_size / dt are the module constants of main.py:5-7; positions, velocities
    and masses are the bare arrays of main.py:25-33, in their dtypes (float32 or float64)."""

    def __init__(self, grid_size, box_size, positions=None, velocities=None, masses=None,
                 interpolation="CIC", capacity=None, device=0, timing=False, fused_gradient=False, rocfft_only=False,
                 float64=False, full_sort=False):
        """full_sort=True (PM_FLAG_FULL_SORT) sorts the whole set into cell order every step instead of merging the few
        particles that changed their cell into the order of the step before (same order; for comparisons).
        float64=True (cloud-in-cell only) makes step / run the float64 build mode (PM_FLAG_F64): mesh, forces,
        positions and velocities all float64 -- the reference's loop with float64 in place of float32 everywhere."""
        if interpolation not in ("CIC", "NGP"):
            raise ValueError(f"interpolation must be 'CIC' or 'NGP' (main.py:9, 40-41), got {interpolation!r}")
        if float64 and interpolation != "CIC":
            raise ValueError("float64=True is the cloud-in-cell branch in float64; the NGP branch is float64 as NumPy runs it")
        self.interpolation = interpolation
        self.float64 = bool(float64)
        self.lib = L.load()
        self.grid_size, self.box_size = int(grid_size), box_size
        n = 0 if positions is None else len(positions)
        self.capacity = int(capacity if capacity is not None else max(n, 1))
        flags = ((L.FLAG_TIMING if timing else 0) | (L.FLAG_FUSED_GRADIENT if fused_gradient else 0)
                 | (L.FLAG_ROCFFT_ONLY if rocfft_only else 0) | (L.FLAG_NGP if interpolation == "NGP" else 0)
                 | (L.FLAG_F64 if float64 else 0) | (L.FLAG_FULL_SORT if full_sort else 0))
        self.full_sort = bool(full_sort)
        self.ctx = self.lib.pm_create(self.grid_size, float(box_size), self.capacity, int(device), flags)
        if not self.ctx:
            raise L.PMError("pm_create failed: " + self.lib.pm_last_error(None).decode())
        self._vel_dtype = np.float32
        if positions is not None:
            self.set_particles(positions, velocities, masses)

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.pm_destroy(self.ctx)
            self.ctx = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ck(self, rc):
        L.check(self.ctx, rc)

    @property
    def n(self):
        return int(self.lib.pm_num_particles(self.ctx))

    def set_particles(self, positions, velocities=None, masses=None):
        x = L.as_particle_array(positions, "positions")
        v = None if velocities is None else L.as_particle_array(velocities, "velocities")
        if self.float64:                # the float64 build mode holds float64 particles
            x = np.ascontiguousarray(x, np.float64)
            v = np.zeros_like(x) if v is None else np.ascontiguousarray(v, np.float64)
        m = None if masses is None else np.ascontiguousarray(masses, dtype=np.float32)
        if v is not None and len(v) != len(x) or m is not None and len(m) != len(x):
            raise ValueError("positions, velocities and masses differ in length")
        self._vel_dtype = np.float32 if v is None else v.dtype
        self._ck(self.lib.pm_set_particles(self.ctx, L.ptr(x), L.PM_F64 if x.dtype == np.float64 else L.PM_F32,
                                           L.ptr(v), L.PM_F64 if self._vel_dtype == np.float64 else L.PM_F32,
                                           L.ptr(m), len(x)))

    def generate(self, kind, n, seed=0, **params):
        """Draw a synthetic particle set on the device in place of set_particles (pm_generate): "uniform",
        "two_galaxy" (two truncated Plummer spheres on a collision course about the box centre, as main.py:28-33
        places its two galaxies) or "disk" (exponential disk).  Lengths default to fixed fractions of the box;
        keyword arguments override them:
          uniform:    sigma_v
          two_galaxy: a1, a2, rmax, centre1, centre2, relative_velocity, sigma_v
          disk:       Rd, z0, Rmax, zmax, centre, sigma_v"""
        q = ic_parameters(kind, float(self.box_size), **params)
        arr = np.asarray(q, dtype=np.float64)
        self._vel_dtype = np.float32
        self._ck(self.lib.pm_generate(self.ctx, L.IC_KINDS[kind], int(n), int(seed) & (2 ** 64 - 1), L.ptr(arr), len(arr)))

    # ---- stages ------------------------------------------------------------------------
    def deposit(self):
        self._ck(self.lib.pm_deposit(self.ctx))

    def solve_potential(self):
        self._ck(self.lib.pm_potential(self.ctx))

    def set_softening(self, length):
        """Gaussian softening of the Poisson solve (pm_set_softening): the density is smoothed over `length` (units of
        box_size) before the potential is taken.  0, the default, is the reference's bare 1/k^2 (functions.py:85)."""
        self._ck(self.lib.pm_set_softening(self.ctx, float(length)))

    def gradient(self):
        self._ck(self.lib.pm_gradient(self.ctx))

    def compute_forces(self):
        self._ck(self.lib.pm_forces(self.ctx))

    def gather_kick_drift(self, dt):
        self._ck(self.lib.pm_gather_kick_drift(self.ctx, float(dt)))

    def step(self, dt, n=1):
        self._ck(self.lib.pm_step(self.ctx, float(dt), int(n)))

    def run(self, steps, dt, trajectory=True, energy_every=10, pinned=True):
        """The main loop of the reference (main.py:44-67): `steps` steps; returns (trajectory, energies) where
        trajectory[s] is `positions.copy()` after step s (main.py:61; an array (steps, N, 3) float32 instead of a
        list, None if trajectory is False) and energies is the array of [KE, PE, KE + PE] rows of main.py:64-67, 72
        (one per energy_every-th step; empty if energy_every is 0).  The snapshot of a step is copied to the host
        while the next step runs; pinned=False keeps the trajectory in ordinary pageable memory."""
        steps, energy_every = int(steps), int(energy_every)
        n = self.n
        traj = None
        if trajectory:
            traj = L.pinned_empty((steps, n, 3), np.float32) if pinned else np.empty((steps, n, 3), np.float32)
        ns = (steps - 1) // energy_every + 1 if energy_every > 0 and steps > 0 else 0
        ke, pe = np.zeros(ns, np.float64), np.zeros(ns, np.float64)
        self._ck(self.lib.pm_run(self.ctx, float(dt), steps, L.ptr(traj), energy_every,
                                 L.ptr(ke) if ns else None, L.ptr(pe) if ns else None))
        return traj, np.stack((ke, pe, ke + pe), axis=1)

    def energies(self):
        """(KE, PE) of functions.py:352-361 on the live state and the held potential mesh."""
        ke, pe = L._dbl(), L._dbl()
        self._ck(self.lib.pm_energies(self.ctx, ke, pe))
        return ke.value, pe.value

    def energies_noalloc(self, device_address):
        """pm_energies_noalloc: KE and PE into two float64 at `device_address` (DEVICE memory), queued on the
        engine's stream: no allocation, no copy, no wait -- synchronize() before the result is read."""
        self._ck(self.lib.pm_energies_noalloc(self.ctx, int(device_address)))

    def deposit_ngp(self):
        self._ck(self.lib.pm_deposit_ngp(self.ctx))

    def mesh_f64(self, which):
        out = np.empty((self.grid_size,) * 3, np.float64)
        self._ck(self.lib.pm_get_mesh_f64(self.ctx, which, L.ptr(out)))
        return out

    def set_mesh_f64(self, which, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.shape != (self.grid_size,) * 3:
            raise ValueError(f"mesh must have shape {(self.grid_size,) * 3}, got {a.shape}")
        self._ck(self.lib.pm_set_mesh_f64(self.ctx, which, L.ptr(a)))

    def density_f64(self):
        return self.mesh_f64(L.MESH_DENSITY)

    # the NGP branch in float64 (functions.py:73-89, main.py:54, functions.py:92-97 on NGP's float64 mesh)
    def solve_potential_f64(self):
        self._ck(self.lib.pm_potential_f64(self.ctx))

    def gradient_f64(self):
        self._ck(self.lib.pm_gradient_f64(self.ctx))

    def compute_forces_ngp_f64(self):
        self._ck(self.lib.pm_forces_ngp_f64(self.ctx))

    # the cloud-in-cell calls of the float64 build mode (functions.py:19-70 and :99-180 with float64 mesh and forces)
    def deposit_f64(self):
        self._ck(self.lib.pm_deposit_f64(self.ctx))

    def compute_forces_f64(self):
        self._ck(self.lib.pm_forces_f64(self.ctx))

    def forces_f64(self):
        f = np.empty((self.n, 3), np.float64)
        self._ck(self.lib.pm_get_forces_f64(self.ctx, L.ptr(f)))
        return f

    def compute_forces_ngp(self):
        self._ck(self.lib.pm_forces_ngp(self.ctx))

    def synchronize(self):
        self._ck(self.lib.pm_synchronize(self.ctx))

    def set_flags(self, timing=False, fused_gradient=False):        # the interpolation branch stays as created
        flags = ((L.FLAG_TIMING if timing else 0) | (L.FLAG_FUSED_GRADIENT if fused_gradient else 0)
                 | (L.FLAG_FULL_SORT if self.full_sort else 0))
        self._ck(self.lib.pm_set_flags(self.ctx, flags))

    # ---- state -------------------------------------------------------------------------
    def _particles(self, want_pos, want_vel):
        n = self.n
        pd, vd = L._int(), L._int()
        self._ck(self.lib.pm_get_particles(self.ctx, None, pd, None, vd))
        x = np.empty((n, 3), np.float64 if pd.value == L.PM_F64 else np.float32) if want_pos else None
        v = np.empty((n, 3), np.float64 if vd.value == L.PM_F64 else np.float32) if want_vel else None
        self._ck(self.lib.pm_get_particles(self.ctx, L.ptr(x), pd, L.ptr(v), vd))
        return x, v

    def positions(self):
        return self._particles(True, False)[0]

    def velocities(self):
        return self._particles(False, True)[1]

    def forces(self):
        f = np.empty((self.n, 3), np.float32)
        self._ck(self.lib.pm_get_forces(self.ctx, L.ptr(f)))
        return f

    def mesh(self, which):
        out = np.empty((self.grid_size,) * 3, np.float32)
        self._ck(self.lib.pm_get_mesh(self.ctx, which, L.ptr(out)))
        return out

    def set_mesh(self, which, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        if a.shape != (self.grid_size,) * 3:
            raise ValueError(f"mesh must have shape {(self.grid_size,) * 3}, got {a.shape}")
        self._ck(self.lib.pm_set_mesh(self.ctx, which, L.ptr(a)))

    def density(self):
        return self.mesh(L.MESH_DENSITY)

    def potential(self):
        return self.mesh(L.MESH_POTENTIAL)

    def potential_gradient(self):
        return tuple(self.mesh(w) for w in (L.MESH_GX, L.MESH_GY, L.MESH_GZ))

    def cell_keys(self):
        """(keys, perm): cell key of the r-th particle in cell order and its original index."""
        n = self.n
        keys, perm = np.empty(n, np.int64), np.empty(n, np.int64)
        self._ck(self.lib.pm_get_cell_keys(self.ctx, L.ptr(keys), L.ptr(perm)))
        return keys, perm

    def last_binning(self):
        """(incremental, movers) of the last binning (pm_last_binning)."""
        import ctypes
        movers = ctypes.c_int64(0)
        rc = self.lib.pm_last_binning(self.ctx, ctypes.byref(movers))
        if rc < 0:
            self._ck(-rc)
        return bool(rc), int(movers.value)

    def stage_times(self):
        """{stage: (milliseconds, calls)} since the last call (needs timing=True)."""
        ms = np.zeros(len(L.STAGES), np.float32)
        calls = np.zeros(len(L.STAGES), np.int32)
        self._ck(self.lib.pm_stage_times(self.ctx, L.ptr(ms), L.ptr(calls)))
        return {s: (float(ms[i]), int(calls[i])) for i, s in enumerate(L.STAGES)}

    def kernel_times(self):
        """{kernel: (milliseconds, launches)} of the single launches pm_kernel_times reports, since the last call
        (needs timing=True): HIP events around each launch on the engine's stream."""
        ms = np.zeros(len(L.KERNELS), np.float32)
        calls = np.zeros(len(L.KERNELS), np.int32)
        self._ck(self.lib.pm_kernel_times(self.ctx, L.ptr(ms), L.ptr(calls)))
        return {s: (float(ms[i]), int(calls[i])) for i, s in enumerate(L.KERNELS)}
