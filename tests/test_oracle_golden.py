"""The CPU oracle against the golden vectors written by the reference itself.

Everything here is bit-exact (np.array_equal) unless a comment says otherwise.
"""
import os

import numpy as np
import pytest

from oracle import pm_oracle as O
from conftest import GOLDEN


def test_edge_cases_cells_density_forces(golden):
    g = golden("edge_cases_g8.npz")
    G, L = int(g["G"]), int(g["L"])
    for tag in ("64", "32"):
        pos, mass = g["pos" + tag], g["mass" + tag]
        assert np.array_equal(O.cell_index(pos, G, L), g["cell" + tag])
        assert np.array_equal(O.CIC(pos, G, L, mass), g["density" + tag])
        f = O.force_CIC((g["gx"], g["gy"], g["gz"]), pos, G, L)
        assert np.array_equal(f, g["forces" + tag])


def test_edge_wrap_float32_remainder(golden):
    g = golden("edge_cases_g8.npz")
    L = int(g["L"])
    x = g["wrap_in"].astype(np.float32)
    n = len(x) - len(x) % 3
    x3 = x[:n].reshape(-1, 3)
    zeros = np.zeros_like(x3)
    out, _ = O.kick_drift(x3, zeros, zeros, 0.005, L)
    assert np.array_equal(out.ravel(), g["wrap_out"][:n])
    assert (g["wrap_out"] == np.float32(L)).any()      # the float32 remainder does return exactly L


@pytest.mark.parametrize("name", ["uniform_n10000_g32.npz", "c1mini_g50.npz"])
def test_step0_stages(golden, name):
    g = golden(name)
    G, L = int(g["G"]), int(g["L"])
    pos, mass = g["pos0"], g["mass"]
    assert np.array_equal(O.cell_index(pos, G, L), g["s0_cell"])
    dens = O.CIC(pos, G, L, mass)
    assert np.array_equal(dens, g["s0_density"])
    pot = O.compute_potential(g["s0_density"], G)
    assert pot.dtype == np.float32 and np.array_equal(pot, g["s0_potential"])
    gx, gy, gz = O.gradient(g["s0_potential"])
    assert np.array_equal(gx, g["s0_gx"]) and np.array_equal(gy, g["s0_gy"]) and np.array_equal(gz, g["s0_gz"])
    f = O.force_CIC((g["s0_gx"], g["s0_gy"], g["s0_gz"]), pos, G, L)
    assert np.array_equal(f, g["s0_forces"])


@pytest.mark.parametrize("name", ["uniform_n10000_g32.npz", "c1mini_g50.npz"])
def test_k_steps_bit_exact(golden, name):
    g = golden(name)
    G, L, dt, K = int(g["G"]), int(g["L"]), float(g["dt"]), int(g["K"])
    x, v, m = g["pos0"], g["vel0"], g["mass"]
    for step in range(K):
        x, v, inter = O.pm_step(x, v, m, G, L, dt, keep=True)
        if step % 10 == 0:
            ke = O.compute_kinetic_energy(v, m)
            pe = O.compute_potential_energy(x, m, inter["potential"], G, L)
            assert ke == g[f"s{step}_KE"] and pe == g[f"s{step}_PE"]
        if f"x{step + 1}" in g:
            assert x.dtype == np.float32 and v.dtype == g["vel0"].dtype
            assert np.array_equal(x, g[f"x{step + 1}"]), step
            assert np.array_equal(v, g[f"v{step + 1}"]), step
        if f"s{step}_forces" in g:
            assert np.array_equal(inter["forces"], g[f"s{step}_forces"])
            assert np.array_equal(inter["density"], g[f"s{step}_density"])


def test_shipped_grid_one_step(golden):
    base = golden("c1mini_g50.npz")
    g = golden("c1mini_g100_step1.npz")
    G, L, dt = int(g["G"]), int(base["L"]), float(base["dt"])
    x, v, inter = O.pm_step(base["pos0"], base["vel0"], base["mass"], G, L, dt, keep=True)
    assert np.array_equal(inter["density"], g["s0_density"])
    assert np.array_equal(inter["forces"], g["s0_forces"])
    assert np.array_equal(inter["potential"][::7, ::7, ::7], g["s0_potential_sample"])
    assert np.array_equal(x, g["x1"]) and np.array_equal(v, g["v1"])


def test_ngp_variants(golden):
    g = golden("uniform_n10000_g32.npz")
    G, L = int(g["G"]), int(g["L"])
    assert np.array_equal(O.NGP(g["pos0"], G, L, g["mass"]), g["ngp_density"])
    grad = (g["s0_gx"], g["s0_gy"], g["s0_gz"])
    assert np.array_equal(O.force_NGP(grad, g["pos0"], G, L), g["ngp_forces"])


@pytest.mark.parametrize("tag", ["", "d_"])
def test_ngp_branch_k_steps_bit_exact(golden, tag):
    """main.py:50-60 with CIC_interpolation = False: float64 meshes and forces; float32 particles (tag "") and the
    shipped run's float64 velocities with float64 positions during step 0 (tag "d_")."""
    g = golden("ngp_n10000_g32.npz")
    G, L, dt, K = int(g["G"]), int(g["L"]), float(g["dt"]), int(g["K"])
    x, v = g["pos0"], g["vel0"]
    if tag:
        x, v = x.astype(np.float64), v.astype(np.float64)
    for step in range(K):
        if step == 0:
            x, v, keep = O.pm_step_ngp(x, v, g["mass"], G, L, dt, keep=True)
            assert keep["potential"].dtype == np.float64 and keep["forces"].dtype == np.float64
            assert np.array_equal(keep["density"], g[tag + "s0_density"])
            assert np.array_equal(keep["potential"], g[tag + "s0_potential"])
            for a, name in zip(keep["grad"], ("gx", "gy", "gz")):
                assert np.array_equal(a, g[tag + "s0_" + name])
            assert np.array_equal(keep["forces"], g[tag + "s0_forces"])
        else:
            x, v = O.pm_step_ngp(x, v, g["mass"], G, L, dt)
        if step + 1 in (1, 5, K):
            assert x.dtype == np.float32 and v.dtype == g[f"{tag}v{step + 1}"].dtype
            assert np.array_equal(x, g[f"{tag}x{step + 1}"]) and np.array_equal(v, g[f"{tag}v{step + 1}"])


def test_import_galaxy(golden):
    g = golden("c1mini_g50.npz")
    p1 = os.path.join(GOLDEN, "galaxy_ellipse_importance_1052_n5000.npz")
    p2 = os.path.join(GOLDEN, "galaxy_diffuse_sphere_importance_1052_n5000.npz")
    x, v, m, lab = O.import_galaxy(p1, p2, separation=400.0, direction=(0.67, 0.33, 0), velocity=(-3000, 300, 0))
    L = int(g["L"])
    x += 0.5 * np.asarray(L, dtype=float) - x.mean(axis=0)          # main.py:33
    assert x.dtype == np.float64 and v.dtype == np.float64 and m.dtype == np.float32
    assert np.array_equal(x, g["pos0"]) and np.array_equal(v, g["vel0"]) and np.array_equal(m, g["mass"])
    assert np.array_equal(lab, g["labels"])
    one = O.import_galaxy(p1)
    assert len(one) == 3 and np.array_equal(one[0], g["single_pos"]) and np.array_equal(one[2], g["single_mass"])


def test_sort_order_is_stable():
    keys = np.array([5, 1, 5, 0, 1, 5, 0])
    assert O.sort_order(keys).tolist() == [3, 6, 1, 4, 0, 2, 5]


def test_deposit_conserves_mass_and_f64_yardstick(golden):
    g = golden("c1mini_g50.npz")
    G, L = int(g["G"]), int(g["L"])
    d64 = O.CIC_f64(g["pos0"], G, L, g["mass"])
    assert abs(d64.sum() - g["mass"].astype(np.float64).sum()) < 1e-9 * d64.sum()
    d32 = O.CIC(g["pos0"], G, L, g["mass"])
    assert np.abs(d32 - d64).max() <= 2e-5 * np.abs(d64).max()
