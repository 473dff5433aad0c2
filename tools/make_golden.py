#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference.

This script is the only place that imports /root/reference.  It is run once, in
the build container (the reference does not exist on the GPU box), and its
outputs -- plain .npz arrays: inputs and the reference's outputs on them -- are
committed.  Nothing here is imported by the product or by the tests.

How the reference is imported
-----------------------------
The reference's functions.py imports numba, pyvista and pyvistaqt, and its
galaxy_generation_functions.py needs enum.StrEnum (Python >= 3.11).  None of
these is installed here, and none takes part in the arithmetic of the PM path,
so they are replaced by inert in-memory stand-ins before the import:
  * numba.njit  -> a decorator that returns the function unchanged, so CIC and
    force_CIC (functions.py:19-70, 99-180) run as plain Python loops;
  * pyvista / pyvistaqt -> empty modules (visualisation only);
  * enum.StrEnum -> class(str, Enum).
main.py, galaxy_generation.py and galaxy_classes.py are scripts with import-time
side effects and are NOT imported; the step loop of main.py:44-67 is driven
from here, calling the reference's own functions in the reference's order.

Numba typing
------------
Under numba, `positions[p,0] % box_size` with an int64 box_size and
`grid_size / box_size` are float64, so all index / weight arithmetic of CIC and
force_CIC is float64 even for float32 positions, while the accumulators are
float32 arrays rounded at every add.  As plain Python on NumPy 2 the same
expressions would stay float32 for float32 positions.  Passing
positions.astype(float64) (value preserving) makes the plain-Python run follow
numba's typing exactly; that is what `ref_CIC` / `ref_force_CIC` below do.
"""
import enum
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    numba = types.ModuleType("numba")

    def njit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    numba.njit = njit
    sys.modules["numba"] = numba
    sys.modules["pyvista"] = types.ModuleType("pyvista")
    pvqt = types.ModuleType("pyvistaqt")
    pvqt.BackgroundPlotter = type("BackgroundPlotter", (), {})
    sys.modules["pyvistaqt"] = pvqt
    if not hasattr(enum, "StrEnum"):
        class StrEnum(str, enum.Enum):
            def __str__(self):
                return str(self.value)
        enum.StrEnum = StrEnum
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, REF)
    import functions as F
    import galaxy_generation_functions as GGF
    return F, GGF


F, GGF = import_reference()


def ref_CIC(pos, G, L, m):
    return F.CIC(np.asarray(pos, dtype=np.float64), G, L, m)


def ref_force_CIC(grad, pos, G, L):
    return F.force_CIC(grad, np.asarray(pos, dtype=np.float64), G, L)


def cell_index(pos, G, L):
    """(i,j,k) as CIC computes them (functions.py:27-39) in float64."""
    p = np.asarray(pos, dtype=np.float64)
    s = (p % L) * (G / L)
    return s.astype(np.int64)


def run_steps(pos, vel, mass, G, L, dt, K, snaps, full_at=(0,)):
    """main.py:44-67 driven step by step; returns a dict of arrays.

    `pos`, `vel` are used with the dtypes given (the shipped default run has
    float64 velocities and float64 positions during step 0 only)."""
    out = {}
    positions = pos.copy()
    velocities = vel.copy()
    for step in range(K):
        cell = cell_index(positions, G, L)
        density = ref_CIC(positions, G, L, mass)
        potential = F.compute_potential(density, G)
        grad = np.gradient(potential)
        forces = ref_force_CIC(grad, positions, G, L)
        if step in full_at:
            t = f"s{step}_"
            out[t + "cell"] = cell.astype(np.int32)
            out[t + "density"] = density
            out[t + "potential"] = potential
            out[t + "gx"], out[t + "gy"], out[t + "gz"] = grad
            out[t + "forces"] = forces
        velocities += forces * dt                       # main.py:58
        positions += velocities * dt                    # main.py:59
        positions = positions.astype(np.float32) % L    # main.py:60
        if step % 10 == 0:                              # main.py:64-67
            out[f"s{step}_KE"] = np.float64(F.compute_kinetic_energy(velocities, mass))
            out[f"s{step}_PE"] = np.float64(F.compute_potential_energy(positions, mass, potential, G, L))
        if (step + 1) in snaps:
            out[f"x{step + 1}"] = positions.copy()
            out[f"v{step + 1}"] = velocities.copy()
        print(f"  step {step} done", flush=True)
    return out


def make_uniform():
    """Uniform random box, float32 state: the correctness config scaled down."""
    N, G, L, dt, K = 10000, 32, 10000, 0.005, 20
    rng = np.random.default_rng(0)
    pos = (rng.random((N, 3)) * L).astype(np.float32)
    vel = rng.normal(0.0, 3.0e4, (N, 3)).astype(np.float32)
    mass = (rng.random(N) + 0.5).astype(np.float32) * np.float32(3.0e4)
    d = dict(pos0=pos, vel0=vel, mass=mass, G=G, L=L, dt=dt, K=K)
    d.update(run_steps(pos, vel, mass, G, L, dt, K, snaps=(1, 5, 10, 20), full_at=(0, 19)))
    # secondary variants on the same inputs: NGP deposit / force (functions.py:11-16, 92-97)
    d["ngp_density"] = F.NGP(pos, G, L, mass)
    grad0 = (d["s0_gx"], d["s0_gy"], d["s0_gz"])
    d["ngp_forces"] = F.force_NGP(grad0, pos, G, L)
    np.savez_compressed(os.path.join(OUT, "uniform_n10000_g32.npz"), **d)
    print("uniform: max|F0|", np.abs(d["s0_forces"]).max(), "max dens", d["s0_density"].max())


def make_ngp():
    """The nearest-grid-point branch (CIC_interpolation = False, main.py:9, 40-41) over K steps: float64 meshes.
    float32 particles, and the same set with float64 velocities and float64 positions during step 0 (the dtypes of
    the shipped two-galaxy run)."""
    N, G, L, dt, K = 10000, 32, 10000, 0.005, 12
    rng = np.random.default_rng(5)
    pos = (rng.random((N, 3)) * L).astype(np.float32)
    pos[: N // 4] = (0.3 * L + rng.normal(0, 0.02 * L, (N // 4, 3))).astype(np.float32)     # a clump
    vel = rng.normal(0.0, 3.0e4, (N, 3)).astype(np.float32)
    mass = (rng.random(N) + 0.5).astype(np.float32) * np.float32(3.0e4)
    d = dict(pos0=pos, vel0=vel, mass=mass, G=G, L=L, dt=dt, K=K)
    for tag, p0, v0 in (("", pos, vel), ("d_", pos.astype(np.float64), vel.astype(np.float64))):
        positions, velocities = p0.copy(), v0.copy()
        for step in range(K):
            density = F.NGP(positions, G, L, mass)                      # main.py:50
            potential = F.compute_potential(density, G)                 # main.py:51
            grad = np.gradient(potential)                               # main.py:54
            forces = F.force_NGP(grad, positions, G, L)                 # main.py:55
            if step == 0:
                d[tag + "s0_density"], d[tag + "s0_potential"], d[tag + "s0_forces"] = density, potential, forces
                d[tag + "s0_gx"], d[tag + "s0_gy"], d[tag + "s0_gz"] = grad
            velocities += forces * dt                                   # main.py:58
            positions += velocities * dt                                # main.py:59
            positions = positions.astype(np.float32) % L                # main.py:60
            if step % 10 == 0:                                          # main.py:64-67
                d[f"{tag}s{step}_KE"] = np.float64(F.compute_kinetic_energy(velocities, mass))
                d[f"{tag}s{step}_PE"] = np.float64(F.compute_potential_energy(positions, mass, potential, G, L))
            if step + 1 in (1, 5, K):
                d[f"{tag}x{step + 1}"], d[f"{tag}v{step + 1}"] = positions.copy(), velocities.copy()
        assert d[tag + "s0_potential"].dtype == np.float64 and d[tag + "s0_forces"].dtype == np.float64
    np.savez_compressed(os.path.join(OUT, "ngp_n10000_g32.npz"), **d)
    print("ngp: max|F0|", np.abs(d["s0_forces"]).max(), "dtypes", d["x1"].dtype, d["v1"].dtype, d["d_v1"].dtype)


def make_c1mini():
    """The shipped two-galaxy run (main.py:25-33) at 5000 + 5000 particles.

    The two galaxy files are produced by the reference's own generators with
    the recipes of galaxy_generation.py:4,7 at reduced N and are committed as
    fixtures, so the importer test starts from files the reference wrote."""
    n_each = 5000
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        os.mkdir("Outputs")
        GGF.generate_galaxy("ellipse", "importance", n_each, a=3, b=1.8, c=1.5, M_tot=100000, seed=1052)
        GGF.generate_galaxy("diffuse_sphere", "importance", N_particles=n_each, R=1.25, M_tot=1.55, seed=1052)
        p1 = "Outputs/ellipse_importance_1052.npz"
        p2 = "Outputs/diffuse_sphere_importance_1052.npz"
        for p, name in ((p1, "galaxy_ellipse_importance_1052_n5000.npz"),
                        (p2, "galaxy_diffuse_sphere_importance_1052_n5000.npz")):
            with np.load(p) as z:
                np.savez_compressed(os.path.join(OUT, name), **{k: z[k] for k in z.files})
        positions, velocities, masses, labels = F.import_galaxy(
            p1, p2, separation=400.0, direction=(0.67, 0.33, 0), velocity=(-3000, 300, 0))
        one = F.import_galaxy(p1)
        os.chdir(cwd)
    L, dt = 10000, 0.005
    positions += 0.5 * np.asarray(L, dtype=float) - positions.mean(axis=0)   # main.py:33
    print("c1mini dtypes", positions.dtype, velocities.dtype, masses.dtype, labels.dtype)
    base = dict(pos0=positions.copy(), vel0=velocities.copy(), mass=masses, labels=labels,
                single_pos=one[0], single_vel=one[1], single_mass=one[2], L=L, dt=dt)
    # multi-step at G=50 with every intermediate of step 0
    d = dict(base, G=50, K=10)
    d.update(run_steps(positions, velocities, masses, 50, L, dt, 10, snaps=(1, 5, 10), full_at=(0,)))
    np.savez_compressed(os.path.join(OUT, "c1mini_g50.npz"), **d)
    # one step at the shipped grid size G=100 (meshes other than density left out for size)
    r = run_steps(positions, velocities, masses, 100, L, dt, 1, snaps=(1,), full_at=(0,))
    e = dict(G=100, K=1, s0_cell=r["s0_cell"], s0_density=r["s0_density"], s0_forces=r["s0_forces"],
             x1=r["x1"], v1=r["v1"], s0_KE=r["s0_KE"], s0_PE=r["s0_PE"],
             s0_potential_sample=r["s0_potential"][::7, ::7, ::7].copy())
    np.savez_compressed(os.path.join(OUT, "c1mini_g100_step1.npz"), **e)
    print("c1mini: occupied cells", int((r["s0_density"] != 0).sum()), "max|F|", np.abs(r["s0_forces"]).max())


def make_edges():
    """Wrap / cell-boundary edge cases on a tiny mesh."""
    G, L = 8, 10000
    h = L / G
    f32 = np.float32
    xs = [0.0, -0.0, -1e-9, 1e-9, float(L), float(np.nextafter(f32(L), f32(0))), float(np.nextafter(float(L), 0.0)),
          h, float(np.nextafter(h, 0.0)), float(np.nextafter(h, 1e9)), 3 * h, 7 * h, 7.5 * h,
          float(np.nextafter(f32(7 * h), f32(0))), -h, -0.5 * h, -float(L), 2.5 * L, -1.5 * L + 3.0,
          L - 1e-12, 1234.5678, 9999.999, 5000.0, float(f32(9999.9995)), -3.0e-4]
    rng = np.random.default_rng(7)
    n = len(xs)
    pos = np.empty((3 * n, 3), dtype=np.float64)
    for ax in range(3):                     # each edge value on each axis, other coords random
        blk = rng.random((n, 3)) * L
        blk[:, ax] = xs
        pos[ax * n:(ax + 1) * n] = blk
    # drop values whose float64 scaled coordinate reaches G: the reference would index out of bounds
    ok = np.all(cell_index(pos, G, L) < G, axis=1)
    pos = pos[ok]
    mass = (rng.random(len(pos)) + 0.5).astype(np.float32)
    pos32 = pos.astype(np.float32)
    ok32 = np.all(cell_index(pos32, G, L) < G, axis=1)
    pos32 = pos32[ok32]
    mass32 = mass[ok32]
    grad = tuple(rng.normal(size=(G, G, G)).astype(np.float32) for _ in range(3))
    d = dict(G=G, L=L, pos64=pos, mass64=mass, pos32=pos32, mass32=mass32,
             gx=grad[0], gy=grad[1], gz=grad[2],
             cell64=cell_index(pos, G, L).astype(np.int32), cell32=cell_index(pos32, G, L).astype(np.int32),
             density64=ref_CIC(pos, G, L, mass), density32=ref_CIC(pos32, G, L, mass32),
             forces64=ref_force_CIC(grad, pos, G, L), forces32=ref_force_CIC(grad, pos32, G, L))
    # main.py:60 on the raw edge values: float32 remainder, may return exactly L
    raw = np.array(xs + [-1e-3, -1e-5, 1e4 + 1e-3, 19999.999, -9999.999], dtype=np.float64)
    d["wrap_in"] = raw
    d["wrap_out"] = raw.astype(np.float32) % L
    np.savez_compressed(os.path.join(OUT, "edge_cases_g8.npz"), **d)
    print("edges:", len(pos), "f64 rows,", len(pos32), "f32 rows; wrap hits L:",
          int((d["wrap_out"] == np.float32(L)).sum()))


def make_virial():
    """virial_ellipsoid (galaxy_generation_functions.py:9-94) called with a seed on the positions and masses of the
    committed ellipse file: inside generate_ellipse_importance it is called without one, so that file's velocities
    cannot be reproduced; this one pins the restatement of the function itself.  Also the equal-mass sphere's
    recipe is not needed: the diffuse sphere file is reproducible from its seed."""
    e = np.load(os.path.join(OUT, "galaxy_ellipse_importance_1052_n5000.npz"))
    v = GGF.virial_ellipsoid(e["positions"], e["masses"], a=3, b=1.8, c=1.5, G=1.0, seed=7)
    np.savez_compressed(os.path.join(OUT, "galaxy_virial_seed7_n5000.npz"), velocities=v, seed=7, a=3.0, b=1.8, c=1.5)
    print("virial:", v.shape, float(np.abs(v).max()))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["edges", "uniform", "c1mini"]
    if "edges" in which:
        make_edges()
    if "uniform" in which:
        make_uniform()
    if "c1mini" in which:
        make_c1mini()
    if "ngp" in which:
        make_ngp()
    if "virial" in which:
        make_virial()
