"""The library on the coordinates that tests/test_properties_cpu.py throws at the restatement: positions outside the box,
at its faces, a hair inside them, tiny negatives whose modulo is the box length itself -- cell keys and cell order bit for
bit against the oracle, the mass in full, forces of a constant gradient, the drift's float32 wrap."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import pm_oracle as O

pytestmark = pytest.mark.gpu

G, L = 16, 100
EDGES = [0.0, -0.0, float(L), float(np.nextafter(np.float32(L), np.float32(0))), -1e-9, 1e-9, -float(L), 2.0 * L,
         float(np.nextafter(np.float32(0), np.float32(-1))), -1.1364755291527169e-35, L / G, L - L / G]
coordinate = st.one_of(st.floats(-3 * L, 4 * L, allow_nan=False, width=32), st.sampled_from(EDGES))


@pytest.fixture(scope="module")
def sim():
    import pm
    with pm.Simulation(G, L, capacity=64) as s:
        yield s


@settings(max_examples=120, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(st.lists(st.tuples(coordinate, coordinate, coordinate), min_size=1, max_size=24), st.sampled_from([np.float32, np.float64]))
def test_cells_order_and_mass_for_any_coordinates(sim, points, dtype):
    pos = np.array(points, np.float32).astype(dtype)
    mass = (np.arange(len(pos)) % 5 + 1).astype(np.float32)
    sim.set_particles(pos, None, mass)
    keys, perm = sim.cell_keys()
    ref = O.cell_keys(pos, G, L)
    order = O.sort_order(ref)
    assert ref.min() >= 0 and ref.max() < G ** 3
    assert np.array_equal(perm, order) and np.array_equal(keys, ref[order])
    sim.deposit()
    d = sim.density()
    want = O.CIC_f64(pos, G, L, mass)
    assert np.abs(d - want).max() <= 2e-6 * max(want.max(), 1.0)
    assert abs(d.sum(dtype=np.float64) - mass.sum()) <= 1e-5 * mass.sum()


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(st.lists(st.tuples(coordinate, coordinate, coordinate), min_size=1, max_size=24),
       st.floats(-400, 400, allow_nan=False, width=32))
def test_one_step_for_any_coordinates(sim, points, speed):
    """A whole step from these positions: the oracle's positions (1e-5 L) and its float32 wrap, which may return L itself."""
    pos = np.array(points, np.float32)
    vel = np.full(pos.shape, speed, np.float32)
    mass = np.ones(len(pos), np.float32)
    sim.set_particles(pos, vel, mass)
    sim.step(0.005)
    x, v = sim.positions(), sim.velocities()
    xo, vo = O.pm_step(pos, vel, mass, G, L, 0.005)
    dx = np.abs(x.astype(np.float64) - xo)
    assert np.minimum(dx, L - dx).max() <= 1e-5 * L
    assert np.abs(v - vo).max() <= 1e-5 * max(np.abs(vo).max(), 1.0)
    assert x.min() >= 0 and x.max() <= L
