"""CPU-side checks: the C-ABI library loads and exports every symbol include/pm.h declares, the
Python host validates its inputs, the bench workload generators are well formed.  No GPU calls."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from pm import _lib
    assert os.path.exists(_lib.LIB_PATH), "run `make -C nbody-simulation_amd` (or __graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "pm.h")).read()
    declared = set(re.findall(r"\b(pm_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found in pm.h"
    for name in declared:
        assert hasattr(lib, name), f"libpm.so lacks {name}"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_missing_library_is_an_error(monkeypatch):
    from pm import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libpm.so")
    with pytest.raises(_lib.PMError):
        _lib.load()


def test_particle_array_validation():
    from pm import _lib
    with pytest.raises(ValueError):
        _lib.as_particle_array(np.zeros((4, 2)), "positions")
    a = _lib.as_particle_array(np.zeros((4, 3), np.int32), "positions")
    assert a.dtype == np.float64
    b = _lib.as_particle_array(np.zeros((6, 3), np.float32)[::2], "positions")
    assert b.flags.c_contiguous and b.dtype == np.float32


def test_simulation_arguments_are_checked_before_the_gpu_is_touched():
    import pm
    with pytest.raises(ValueError):
        pm.Simulation(8, 100.0, interpolation="TSC")            # main.py:9 knows CIC and NGP only


def test_spectrum_row_interleave_is_a_permutation():
    """Row (i, j) -> row of the half-spectrum buffer (pm_spec_row in csrc/pm_internal.hpp), restated: every buffer
    row is used exactly once, and with no interleave it is i * G + j."""
    def spec_row(G, sb, i, j):
        return (((i >> sb) * G + j) << sb) | (i & ((1 << sb) - 1))
    G = 32
    i, j = np.meshgrid(np.arange(G), np.arange(G), indexing="ij")
    assert np.array_equal(spec_row(G, 0, i, j), i * G + j)
    for sb in (1, 2, 4):
        rows = spec_row(G, sb, i, j).ravel()
        assert np.array_equal(np.sort(rows), np.arange(G * G))
        # along i, 2^sb consecutive planes are consecutive rows; along j the stride is 2^sb rows
        assert spec_row(G, sb, 5, 7) - spec_row(G, sb, 4, 7) == 1 and spec_row(G, sb, 4, 8) - spec_row(G, sb, 4, 7) == 1 << sb


def test_bench_workloads():
    import bench
    for name in ("uniform", "two_galaxy", "disk"):
        pos, vel, mass = bench.make_workload(name, 20000, 10000, 1)
        assert pos.shape == (20000, 3) and pos.dtype == np.float32 and vel.dtype == np.float32
        assert pos.min() >= 0 and pos.max() <= 10000 and mass.dtype == np.float32
    b = bench.stage_bytes(100, 1000)
    assert sum(b.values()) == 64 * 100 + 80 * 1000
    assert sum(bench.stage_bytes(100, 1000, fused=True).values()) == 64 * 100 + 80 * 1000
    # what the fused stages cannot avoid is less than their share of 64 N + 80 M, stage by stage, and the single-launch
    # stages are priced the same way in both tables
    f, share = bench.fused_stage_bytes(100, 1000), bench.stage_bytes(100, 1000, fused=True)
    assert all(f[s] <= share[s] for s in f) and sum(f.values()) == 64 * 100 + 40 * 1000
    # the kernels the roofline object can name are priced like the stages they are (or are the longest part of)
    k = bench.kernel_bytes(100, 1000)
    assert k["deposit_rows"][1] == f["deposit"] and k["gather"][1] == f["gather"]
    assert k["col_y_fwd"][1] == f["fft_fwd"] and k["col_x_green"][1] == f["green"]
    assert k["col_y_inv"][1] + k["row_inv"][1] == f["fft_inv"]
    assert k["sort_pass"][1] == 0 and k["reorder"][1] == 0             # cell order has no algorithmic bytes
    from pm import _lib
    assert set(k) == set(_lib.KERNELS)


def test_import_galaxy_matches_golden(golden):
    """pm.functions.import_galaxy on the galaxy files the reference's generators wrote."""
    from conftest import GOLDEN
    from pm import functions as F
    g = golden("c1mini_g50.npz")
    p1 = os.path.join(GOLDEN, "galaxy_ellipse_importance_1052_n5000.npz")
    p2 = os.path.join(GOLDEN, "galaxy_diffuse_sphere_importance_1052_n5000.npz")
    x, v, m, lab = F.import_galaxy(p1, p2, separation=400.0, direction=(0.67, 0.33, 0), velocity=(-3000, 300, 0))
    x += 0.5 * np.asarray(int(g["L"]), dtype=float) - x.mean(axis=0)        # main.py:33
    assert x.dtype == np.float64 and v.dtype == np.float64 and m.dtype == np.float32
    assert np.array_equal(x, g["pos0"]) and np.array_equal(v, g["vel0"]) and np.array_equal(m, g["mass"])
    assert np.array_equal(lab, g["labels"])
    one = F.import_galaxy(p1)
    assert len(one) == 3 and np.array_equal(one[0], g["single_pos"]) and np.array_equal(one[1], g["single_vel"])


def test_star_import_supplies_what_main_py_takes_from_it():
    """main.py:1 is `from functions import *` and the script then uses, without importing them itself, `time`
    (main.py:20), `np` (:33, :54, :60), import_galaxy (:25), CIC / NGP / force_CIC / force_NGP (:40-41),
    compute_potential (:51), the energy functions (:65-66) and elapsed_time (:69)."""
    from pm import functions as F
    ns = {}
    exec("from pm.functions import *", ns)
    for name in ("time", "np", "import_galaxy", "CIC", "NGP", "force_CIC", "force_NGP", "compute_potential",
                 "compute_kinetic_energy", "compute_potential_energy", "elapsed_time"):
        assert name in F.__all__ and name in ns, name
    assert ns["np"] is np and ns["time"].time() > 0 and 0 <= ns["elapsed_time"](ns["time"].time()) < 1


def test_every_free_name_of_the_reference_main_loop_is_exported():
    """Where the reference is at hand (this container, not the GPU box): parse its main.py, lines 1-67, and check that
    every name the script reads but neither binds nor imports itself -- i.e. expects from the star import or the
    builtins -- is exported by pm.functions."""
    import ast
    import builtins
    path = "/root/reference/main.py"
    if not os.path.exists(path):
        pytest.skip("the reference is not on this machine")
    from pm import functions as F
    lines = open(path, encoding="utf-8").read().splitlines()[:67]
    assert lines[0].strip() == "from functions import *"
    tree = ast.parse("\n".join(lines[1:]))
    bound, read = set(), set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Name):
            (bound if isinstance(node.ctx, ast.Store) else read).add(node.id)
        elif isinstance(node, (ast.Import, ast.ImportFrom)):
            bound.update((a.asname or a.name).split(".")[0] for a in node.names)
    free = {n for n in read - bound if not hasattr(builtins, n)}
    assert free and free <= set(F.__all__), free - set(F.__all__)


def test_galaxy_classes():
    from pm.galaxy_classes import DiskGalaxy, Galaxy, generate_galaxy
    # positional order of galaxy_classes.py:26-35: (morphology, sampling, N_particles, Rd, z0, M_tot, seed)
    d = DiskGalaxy("disk", "importance", 100, 10, 1)
    assert d.Rd == 10 and d.z0 == 1 and d.M_tot == 100 and d.seed is None
    d = DiskGalaxy("disk", "importance", 100, 10, 1, 2.5, 7)
    assert (d.Rd, d.z0, d.M_tot, d.seed) == (10, 1, 2.5, 7) and str(d) == "disk_importance_7"
    # galaxy_classes.py:6-13: (morphology, sampling, N_particles, M_tot, seed)
    b = Galaxy("disk", "equal-mass", 50, 3.0, 9)
    assert (b.N_particles, b.M_tot, b.seed) == (50, 3.0, 9) and Galaxy("disk", "equal-mass", 50).M_tot == 50
    gal = generate_galaxy("disk", "equal-mass", 10000, 10, 1, seed=42)
    assert isinstance(gal, DiskGalaxy) and isinstance(gal, Galaxy)
    assert str(gal) == "disk_equal-mass_42" and gal.M_tot == 10000 and gal.Rd == 10 and gal.z0 == 1
    assert gal.galaxy_type() == "disk"
    assert generate_galaxy("disk", "importance", 5, M_tot=2.5).M_tot == 2.5
    with pytest.raises(ValueError):
        generate_galaxy("ellipse", "importance", 10)


def test_oracle_softening_is_a_gaussian_smoothing_of_the_density():
    # exp(-soft k2) in the solve == the bare solve of the density convolved with a periodic Gaussian of that width
    from scipy.ndimage import gaussian_filter
    from oracle import pm_oracle as O
    G, L, cells = 32, 10000.0, 1.5
    rng = np.random.default_rng(2)
    dens = rng.random((G, G, G)).astype(np.float32) ** 8
    soft = O.softening_coefficient(cells * L / G, G, L)
    assert soft == np.float32(0.5 * (2 * np.pi * cells) ** 2)
    a = O.compute_potential(dens, G, soft=soft)
    b = O.compute_potential(gaussian_filter(dens.astype(np.float64), cells, mode="wrap", truncate=8.0).astype(np.float32), G)
    assert np.abs(a - b).max() <= 1e-4 * np.abs(b).max()
    assert np.abs(a - O.compute_potential(dens, G)).max() > 0.05 * np.abs(a).max()
    assert np.array_equal(O.compute_potential(dens, G, soft=0.0), O.compute_potential(dens, G))


def test_lattice_case_agrees_with_the_oracle():
    """tests/lattice_case.py (the closed form that the 2048^3 slab test is checked against) against one oracle step."""
    import lattice_case as LC
    from oracle import pm_oracle as O
    G, L, kstride, mode, eps, dt = 32, 10000, 8, 2, 0.5, 0.005
    v0 = 0.8 * (L / G) / dt
    scale = LC.scale_for(G, kstride, mode, eps, v0, dt)
    pos, vel, mass, ids = LC.slab_particles(G, L, kstride, mode, eps, scale, v0, 0, G)
    x, v = O.pm_step(pos, vel, mass, G, L, dt)
    want, dv = LC.expected_velocities(ids, G, kstride, mode, eps, scale, v0, dt)
    assert 0.02 * v0 < dv < 0.1 * v0 and np.abs(v - want).max() <= 1e-4 * dv
    assert np.abs(want[:, 0] - np.float32(v0)).max() > 0.5 * dv        # the forces are not nothing
    assert np.array_equal(x, (pos + v * np.float32(dt)) % np.float32(L))


def test_float64_restatement_is_the_float32_one_up_to_float32():
    """oracle.pm_step_f64 (what the float64 build mode is compared with) against the reference-pinned float32 step."""
    from oracle import pm_oracle as O
    rng = np.random.default_rng(3)
    n, G, L, dt = 5000, 16, 100, 0.01
    pos = (rng.random((n, 3)) * L).astype(np.float32)
    vel = rng.normal(0, 30, (n, 3)).astype(np.float32)
    mass = (rng.random(n) + 0.5).astype(np.float32)
    x32, v32, k32 = O.pm_step(pos, vel, mass, G, L, dt, keep=True)
    x64, v64, k64 = O.pm_step_f64(pos, vel, mass, G, L, dt, keep=True)
    assert k64["density"].dtype == np.float64 and k64["forces"].dtype == np.float64 and x64.dtype == np.float64
    assert np.abs(k64["density"] - k32["density"]).max() <= 1e-5 * k32["density"].max()
    assert np.abs(k64["forces"] - k32["forces"]).max() <= 1e-4 * np.abs(k32["forces"]).max()
    dx = np.abs(x64 - x32)
    assert np.minimum(dx, L - dx).max() <= 1e-5 * L and np.abs(v64 - v32).max() <= 1e-5 * np.abs(v32).max()
    assert abs(k64["density"].sum() - mass.sum(dtype=np.float64)) <= 1e-12 * mass.sum()
