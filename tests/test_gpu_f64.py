"""The float64 build mode (PM_FLAG_F64: cloud-in-cell with float64 mesh, forces, positions and velocities) through the C
ABI against the float64 CPU restatement oracle.pm_step_f64: 1e-12 of the largest value for density, potential, gradient
and forces, 1e-12 L and 1e-12 max|v| for positions and velocities after K steps -- where the float32 path is held to 1e-5.
The reference itself has no float64 switch (its mesh is float32, functions.py:22): parity of this mode is with the
restatement, which tests/test_host_cpu.py ties to the reference-pinned float32 step."""
import numpy as np
import pytest

from oracle import pm_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-12


@pytest.fixture(scope="module")
def pm():
    import pm as _pm
    _pm.load()
    return _pm


def particles(n, L, seed, clump=True):
    rng = np.random.default_rng(seed)
    pos = rng.random((n, 3)) * L
    if clump:
        pos[: n // 3] = 0.4 * L + rng.normal(0, 0.01 * L, (n // 3, 3))
    pos[-3:] = [[0.0, 0.0, 0.0], [np.nextafter(L, 0), 0.5 * L, np.nextafter(L, 0)], [L - 1e-9, 1e-9, 0.5 * L]]     # faces and corners
    vel = rng.normal(0, 0.02 * L, (n, 3))
    mass = (rng.random(n) + 0.5).astype(np.float32)
    return pos, vel, mass


def close(a, b, scale=None):
    scale = np.abs(b).max() if scale is None else scale
    return np.abs(a - b).max() <= TOL * scale


@pytest.mark.parametrize("G", [16, 50, 64])
def test_stages_in_float64(pm, G):
    from pm import _lib
    n, L, dt = 30_000, 100.0, 0.01
    pos, vel, mass = particles(n, L, G)
    xo, vo, k = O.pm_step_f64(pos, vel, mass, G, L, dt, keep=True)
    with pm.Simulation(G, L, pos, vel, mass, float64=True) as sim:
        sim.deposit_f64()
        assert close(sim.mesh_f64(_lib.MESH_DENSITY), k["density"])
        sim.solve_potential_f64()
        assert close(sim.mesh_f64(_lib.MESH_POTENTIAL), k["potential"])
        sim.gradient_f64()
        for w, g in zip((_lib.MESH_GX, _lib.MESH_GY, _lib.MESH_GZ), k["grad"]):
            assert close(sim.mesh_f64(w), g, np.abs(np.asarray(k["grad"])).max())
        sim.compute_forces_f64()
        assert close(sim.forces_f64(), k["forces"])
        sim.step(dt)
        x, v = sim.positions(), sim.velocities()
    assert x.dtype == np.float64 and v.dtype == np.float64
    dx = np.abs(x - xo)
    assert np.minimum(dx, L - dx).max() <= TOL * L and close(v, vo)


def test_hundred_steps_in_float64(pm):
    """K = 100 steps: 1e-12 L and 1e-12 max|v| -- the bar the float32 path meets at 1e-5."""
    n, G, L, dt, K = 100_000, 64, 10000.0, 0.005, 100
    rng = np.random.default_rng(1)
    pos = rng.random((n, 3)) * L
    vel = rng.normal(0, 0.1 * L / G / dt, (n, 3))
    mass = (rng.random(n) + 0.5).astype(np.float32)
    x, v = pos, vel
    for _ in range(K):
        x, v = O.pm_step_f64(x, v, mass, G, L, dt)
    with pm.Simulation(G, L, pos, vel, mass, float64=True) as sim:
        sim.step(dt, K)
        xg, vg = sim.positions(), sim.velocities()
        ke, pe = sim.energies()
    dx = np.abs(xg - x)
    assert np.minimum(dx, L - dx).max() <= TOL * L
    assert np.abs(vg - v).max() <= TOL * np.abs(v).max()
    assert np.abs(xg - pos).max() > 1e-3 * L            # they did move
    assert abs(ke - O.compute_kinetic_energy(v, mass)) <= 1e-10 * ke


def test_correctness_configuration_in_float64(pm):
    """N = 10^6 on a 128^3 mesh (the correctness configuration), five steps."""
    n, G, L, dt = 1_000_000, 128, 10000.0, 0.005
    rng = np.random.default_rng(2)
    pos = rng.random((n, 3)) * L
    vel = rng.normal(0, 0.3 * L / G / dt, (n, 3))
    mass = np.ones(n, np.float32)
    x, v = pos, vel
    for _ in range(5):
        x, v = O.pm_step_f64(x, v, mass, G, L, dt)
    with pm.Simulation(G, L, pos, vel, mass, float64=True) as sim:
        sim.step(dt, 5)
        xg, vg = sim.positions(), sim.velocities()
    dx = np.abs(xg - x)
    assert np.minimum(dx, L - dx).max() <= TOL * L and np.abs(vg - v).max() <= TOL * np.abs(v).max()


def test_float64_mode_wants_float64_particles(pm):
    from pm import _lib
    lib = _lib.load()
    ctx = lib.pm_create(16, 100.0, 8, 0, _lib.FLAG_F64)
    assert ctx
    x = np.zeros((4, 3), np.float32)
    rc = lib.pm_set_particles(ctx, x.ctypes.data, _lib.PM_F32, x.ctypes.data, _lib.PM_F32, None, 4)
    assert rc != 0 and b"float64" in lib.pm_last_error(ctx)
    lib.pm_destroy(ctx)
    assert not lib.pm_create(16, 100.0, 8, 0, _lib.FLAG_F64 | _lib.FLAG_NGP) and b"PM_FLAG_F64" in lib.pm_last_error(None)
    with pm.Simulation(16, 100.0, x.astype(np.float64), None, None, float64=True) as sim:
        with pytest.raises(_lib.PMError):
            sim.run(2, 0.01, trajectory=True, energy_every=0)      # float32 snapshots: not in this mode
        traj, en = sim.run(2, 0.01, trajectory=False, energy_every=1)
        assert traj is None and en.shape == (2, 3)
    # and on an ordinary context the float64 cloud-in-cell calls refuse float32 particles
    with pm.Simulation(16, 100.0, x, x, None) as sim:
        with pytest.raises(_lib.PMError):
            sim.deposit_f64()
