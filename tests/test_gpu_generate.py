"""GPU: particle sets drawn on the device (pm_generate through the C ABI) against their numpy restatement.

Bar: the device and numpy evaluate the same float64 expressions on the same Philox draws and differ only in the last
bit of log / pow / sin / cos, so after the cast to float32 nearly every coordinate is identical and none is further
than 2 float32 ulps of its magnitude scale away (1e-6 relative to the box size for positions, to |v| + sigma for
velocities).  Against the reference's own generators the sets agree in distribution only (see oracle/pm_ic.py).
"""
import numpy as np
import pytest

from oracle import pm_ic as I
from oracle import pm_oracle as O

pytestmark = pytest.mark.gpu

L = 10000.0
KINDS = (("uniform", I.UNIFORM), ("two_galaxy", I.TWO_GALAXY), ("disk", I.DISK))


@pytest.fixture(scope="module")
def pm():
    import pm as _pm
    _pm.load()
    return _pm


def periodic_diff(a, b):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    return np.minimum(d, L - d)


@pytest.mark.parametrize("name,kind", KINDS)
@pytest.mark.parametrize("n,seed", [(100_003, 1052), (4096, 2 ** 40 + 5)])
def test_generated_set_matches_restatement(pm, name, kind, n, seed):
    with pm.Simulation(32, L, capacity=n) as sim:
        sim.generate(name, n, seed)
        assert sim.n == n
        x, v = sim.positions(), sim.velocities()
    rx, rv, _ = I.generate(kind, n, seed, L)
    assert x.dtype == np.float32 and v.dtype == np.float32
    assert np.mean(x == rx) > 0.999 and np.mean(v == rv) > 0.999
    assert periodic_diff(x, rx).max() <= 1e-6 * L
    assert np.abs(v.astype(np.float64) - rv).max() <= 1e-6 * (np.abs(rv).max() + 10.0)


def test_overridden_parameters_and_rejection_fallback(pm):
    n = 20_000
    over = dict(a1=50.0, a2=400.0, rmax=300.0, centre1=(100.0, 200.0, 300.0), centre2=(9000.0, 8000.0, 50.0),
                relative_velocity=(10.0, -20.0, 30.0), sigma_v=0.0)
    from pm.simulation import ic_parameters
    with pm.Simulation(16, L, capacity=n) as sim:
        sim.generate("two_galaxy", n, 9, **over)
        x, v = sim.positions(), sim.velocities()
    rx, rv, _ = I.generate(I.TWO_GALAXY, n, 9, L, ic_parameters("two_galaxy", L, **over))
    assert np.mean(x == rx) > 0.999 and periodic_diff(x, rx).max() <= 1e-6 * L
    # no dispersion: every particle carries exactly half of the relative velocity
    assert np.array_equal(v, rv) and np.array_equal(v[0], np.float32([-5.0, 10.0, -15.0])) and np.array_equal(v[-1], np.float32([5.0, -10.0, 15.0]))
    # second galaxy: rmax < a2, most radii are rejected a few times, none ends beyond rmax
    sep = (x[n // 2:].astype(np.float64) - np.array(over["centre2"]) + L / 2) % L - L / 2
    assert np.linalg.norm(sep, axis=1).max() <= 300.0 * (1 + 1e-5)


def test_generated_set_steps_like_the_same_arrays_set_from_the_host(pm):
    # the records pm_generate writes are the ones pm_set_particles would write for the same arrays: identical bits after steps
    n, G = 50_000, 32
    with pm.Simulation(G, L, capacity=n) as a:
        a.generate("two_galaxy", n, 5)
        x0, v0 = a.positions(), a.velocities()
        a.step(0.005, 3)
        xa, va = a.positions(), a.velocities()
        keys_a = a.cell_keys()
    with pm.Simulation(G, L, x0, v0, np.ones(n, np.float32)) as b:
        b.step(0.005, 3)
        xb, vb = b.positions(), b.velocities()
        keys_b = b.cell_keys()
    assert np.array_equal(xa, xb) and np.array_equal(va, vb)
    assert np.array_equal(keys_a[0], keys_b[0]) and np.array_equal(keys_a[1], keys_b[1])
    # and one step of it against the CPU restatement of the step
    rx, rv = O.pm_step(x0, v0, np.ones(n, np.float32), G, int(L), 0.005)
    with pm.Simulation(G, L, capacity=n) as c:
        c.generate("two_galaxy", n, 5)
        c.step(0.005, 1)
        assert periodic_diff(c.positions(), rx).max() <= 1e-5 * L
        assert np.abs(c.velocities() - rv).max() <= 1e-5 * np.abs(rv).max()


def test_generate_argument_errors(pm):
    lib = pm.load()
    with pm.Simulation(16, L, capacity=100) as sim:
        with pytest.raises(pm.PMError, match="n outside"):
            sim.generate("uniform", 101)
        with pytest.raises(pm.PMError, match="positive"):
            sim.generate("disk", 10, Rd=0.0)
        with pytest.raises(pm.PMError, match="sigma_v"):
            sim.generate("uniform", 10, sigma_v=-1.0)
        with pytest.raises(pm.PMError, match="finite"):
            sim.generate("uniform", 10, sigma_v=float("nan"))
        q = np.zeros(3)
        assert lib.pm_generate(sim.ctx, 0, 10, 0, q.ctypes.data, 3) != 0      # uniform takes one parameter
        assert lib.pm_generate(sim.ctx, 7, 10, 0, q.ctypes.data, 1) != 0      # no such kind
        sim.generate("uniform", 0)
        assert sim.n == 0
        sim.step(0.005, 1)                                                      # an empty set steps


def test_headline_size_set_in_distribution(pm):
    # 16,777,216 particles drawn on the device: Plummer medians, momentum balance, and the set steps
    n, G = 16_777_216, 512
    q = I.default_params(I.TWO_GALAXY, L)
    with pm.Simulation(G, L, capacity=n, fused_gradient=True) as sim:
        sim.generate("two_galaxy", n, 1052)
        x, v = sim.positions(), sim.velocities()
        for half, a, c in ((slice(0, n // 2), q[0], q[3:6]), (slice(n // 2, n), q[1], q[6:9])):
            sep = (x[half].astype(np.float64) - np.array(c) + L / 2) % L - L / 2
            r = np.sqrt((sep * sep).sum(1))
            half_mass = 0.5 * (1.0 + (a / (0.45 * L)) ** 2) ** -1.5        # half of the mass inside the truncation radius
            assert np.median(r) == pytest.approx(a / np.sqrt(half_mass ** (-2.0 / 3.0) - 1.0), rel=0.003)
            assert r.max() < 0.45 * L * (1 + 1e-6)
        assert np.allclose(v.mean(0, dtype=np.float64), 0, atol=0.02)
        # the first 4096 particles are those of any shorter set with the same seed
        rx = I.generate(I.TWO_GALAXY, 8192, 1052, L)[0][:4096]
        assert np.mean(x[:4096] == rx) > 0.999 and periodic_diff(x[:4096], rx).max() <= 1e-6 * L
        sim.step(0.005, 2)
        keys, perm = sim.cell_keys()
        assert np.all(np.diff(keys) >= 0) and np.array_equal(np.sort(perm), np.arange(n))
