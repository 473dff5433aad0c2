"""The restatement of the reference's importance-sampled galaxy generators (oracle/pm_galaxy.py) against files the
reference itself wrote (tests/golden/galaxy_*.npz, tools/make_golden.py): bit for bit."""
import os

import numpy as np

from oracle import pm_galaxy as OG

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_ellipse_positions_and_masses_are_the_references():
    e = np.load(os.path.join(GOLD, "galaxy_ellipse_importance_1052_n5000.npz"))
    nrm, z1, z2 = OG.reference_normals("ellipse", 5000, 1052)
    p, v, m, labels = OG.ellipse_importance(nrm, z1, z2, 3, 1.8, 1.5, 100000)       # galaxy_generation.py:4 at n = 5000
    assert np.array_equal(p, e["positions"]) and np.array_equal(m, e["masses"]) and np.array_equal(labels, e["labels"])
    assert abs(m.sum() - 100000) < 1e-6
    # the reference's velocities came from an unseeded generator: only what holds for any draw can be compared
    for vel in (v, e["velocities"]):
        assert np.abs(np.average(vel, weights=m, axis=0)).max() < 1e-9 * np.abs(vel).max()


def test_virial_ellipsoid_is_the_references():
    e = np.load(os.path.join(GOLD, "galaxy_ellipse_importance_1052_n5000.npz"))
    g = np.load(os.path.join(GOLD, "galaxy_virial_seed7_n5000.npz"))
    rng = np.random.default_rng(int(g["seed"]))
    z1, z2 = rng.standard_normal(5000), rng.standard_normal(5000)
    v = OG.virial_ellipsoid(e["positions"], e["masses"], float(g["a"]), float(g["b"]), float(g["c"]), z1, z2)
    assert np.array_equal(v, g["velocities"])


def test_diffuse_sphere_is_the_references():
    s = np.load(os.path.join(GOLD, "galaxy_diffuse_sphere_importance_1052_n5000.npz"))
    nrm, q = OG.reference_normals("diffuse_sphere", 5000, 1052)
    p, v, m, labels = OG.diffuse_sphere_importance(nrm, q, 1.25, 1.55)              # galaxy_generation.py:7 at n = 5000
    assert np.array_equal(p, s["positions"]) and np.array_equal(m, s["masses"])
    assert np.array_equal(v, s["velocities"]) and np.array_equal(labels, s["labels"])
    assert (labels == "1").sum() == 2500 and abs(m[labels == "1"].sum() - 0.25 * 1.55) < 1e-12
