"""Known answers and properties of the reference's path that need no reference run (the reference has no tests of its
own), checked on the CPU restatement: mass conservation without a cell volume (functions.py:70), a particle on a node,
one Fourier mode through compute_potential (functions.py:73-89), np.gradient of a ramp (main.py:54), the gather of a
constant gradient (functions.py:140-178), and the wrap at the edges of the box (functions.py:27-29, main.py:60)."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import pm_oracle as O

G, L = 16, 100


def test_mass_is_conserved_and_not_divided_by_a_cell_volume():
    rng = np.random.default_rng(0)
    pos = (rng.random((5000, 3)) * L).astype(np.float32)
    mass = (rng.random(5000) + 0.5).astype(np.float32)
    d = O.CIC(pos, G, L, mass)
    assert d.dtype == np.float32 and d.shape == (G, G, G)
    assert abs(d.sum(dtype=np.float64) - mass.sum(dtype=np.float64)) <= 1e-5 * mass.sum()


def test_a_particle_on_a_node_feeds_one_cell():
    h = L / G
    pos = np.array([[3 * h, 5 * h, 7 * h]], np.float32)        # exact in float32: h = 6.25
    d = O.CIC(pos, G, L, np.array([2.5], np.float32))
    assert d[3, 5, 7] == np.float32(2.5) and np.count_nonzero(d) == 1


def test_a_particle_at_the_last_node_wraps_its_far_corners_to_plane_zero():
    h = L / G
    pos = np.array([[(G - 0.5) * h] * 3], np.float32)
    d = O.CIC(pos, G, L, np.ones(1, np.float32))
    idx = np.argwhere(d > 0)
    assert len(idx) == 8 and set(idx.reshape(-1).tolist()) == {0, G - 1}      # functions.py:56-58: (i + 1) % G
    assert np.allclose(d[d > 0], 0.125)


@pytest.mark.parametrize("mode", [(1, 0, 0), (0, 2, 0), (1, 1, 3), (0, 0, G // 2)])
def test_one_fourier_mode_is_divided_by_k_squared(mode):
    """delta = cos(2 pi m.x / G) -> potential = +delta / |m / G|^2 (k in cycles per sample, functions.py:75-85)."""
    i, j, k = np.meshgrid(*(np.arange(G),) * 3, indexing="ij")
    delta = np.cos(2 * np.pi * (mode[0] * i + mode[1] * j + mode[2] * k) / G).astype(np.float32)
    phi = O.compute_potential(delta + np.float32(3.0), G)                     # the mean does not matter (:81-82, :86)
    k2 = sum((m / G if m <= G // 2 else (m - G) / G) ** 2 for m in mode)
    assert np.abs(phi - delta / k2).max() <= 1e-5 * np.abs(delta / k2).max()


def test_gradient_of_a_ramp_is_exact_inside_and_one_sided_on_the_faces():
    i, j, k = np.meshgrid(*(np.arange(G, dtype=np.float32),) * 3, indexing="ij")
    gx, gy, gz = O.gradient(2 * i - 3 * j + 0.5 * k)
    assert np.all(gx == 2) and np.all(gy == -3) and np.all(gz == 0.5)       # faces included: first differences of a ramp
    # ... and a periodic field is NOT differenced periodically: the faces use their inner neighbour only (main.py:54)
    f = np.cos(2 * np.pi * i / G).astype(np.float32)
    gx = O.gradient(f)[0]
    assert np.allclose(gx[0], f[1] - f[0]) and np.allclose(gx[-1], f[-1] - f[-2])
    assert np.allclose(gx[1:-1], (f[2:] - f[:-2]) / 2)
    assert not np.allclose(gx[0], (f[1] - f[-1]) / 2)


def test_gather_of_a_constant_gradient_returns_it():
    rng = np.random.default_rng(1)
    pos = (rng.random((2000, 3)) * L).astype(np.float32)
    g = [np.full((G, G, G), c, np.float32) for c in (1.5, -2.0, 0.25)]
    f = O.force_CIC(g, pos, G, L)
    assert np.abs(f - np.array([1.5, -2.0, 0.25], np.float32)).max() <= 2e-6      # the eight weights add up to one


coordinate = st.one_of(
    st.floats(-3 * L, 4 * L, allow_nan=False, width=32),
    st.sampled_from([0.0, -0.0, float(L), float(np.nextafter(np.float32(L), np.float32(0))), -1e-9, 1e-9, -float(L), 2.0 * L,
                     float(np.nextafter(np.float32(0), np.float32(-1)))]))


@settings(max_examples=300, deadline=None)
@given(st.lists(st.tuples(coordinate, coordinate, coordinate), min_size=1, max_size=8))
def test_any_coordinate_lands_in_a_cell_of_the_box(points):
    """functions.py:27-39 for positions outside [0, L), at L, just below L, and at negative zero: every particle gets a
    cell in [0, G)^3, its mass arrives in full, and shifting it by a box length changes nothing but rounding."""
    pos = np.array(points, np.float32)
    c = O.cell_index(pos, G, L)
    assert c.min() >= 0 and c.max() < G
    mass = np.ones(len(pos), np.float32)
    d = O.CIC_f64(pos, G, L, mass)
    assert abs(d.sum() - len(pos)) <= 1e-9 * len(pos) and d.min() >= -1e-12
    inside = (pos >= 0).all(axis=1) & (pos < L).all(axis=1)
    if inside.any():
        p = pos[inside].astype(np.float64)
        assert np.array_equal(O.cell_index(p, G, L), O.cell_index(p + L, G, L)) or \
            np.abs(O.CIC_f64(p, G, L, mass[inside]) - O.CIC_f64(p + L, G, L, mass[inside])).max() <= 1e-9


@settings(max_examples=200, deadline=None)
@given(st.floats(-2 * L, 3 * L, allow_nan=False, width=32), st.floats(-500, 500, allow_nan=False, width=32))
def test_drift_wraps_like_numpy_float32_remainder(x, v):
    """main.py:59-60 in float32: the result is np.remainder's, which lies in [0, L] -- L itself for a tiny negative."""
    pos = np.full((1, 3), x, np.float32)
    vel = np.full((1, 3), v, np.float32)
    xn, vn = O.kick_drift(pos, vel, np.zeros((1, 3), np.float32), 0.005, L)
    want = (pos + vel * np.float32(0.005)).astype(np.float32) % np.float32(L)
    assert np.array_equal(xn, want) and 0 <= xn.min() and xn.max() <= L
    assert np.array_equal(vn, vel)
