"""pm_generate_galaxy (the reference's importance-sampled generators on the device) through the C ABI: against files
the reference wrote, against the CPU restatement, and the device's own random stream through what holds for any draw."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-12                                 # relative to the largest value of the array


def close(a, b, tol=TOL):
    return np.abs(a - b).max() <= tol * np.abs(b).max()


def test_diffuse_sphere_same_seed_same_galaxy_as_the_reference():
    from pm import galaxy_generation_functions as G
    s = np.load(os.path.join(GOLD, "galaxy_diffuse_sphere_importance_1052_n5000.npz"))
    p, v, m, labels = G.generate_diffuse_sphere_importance(5000, 1.25, 1.55, seed=1052)
    assert np.array_equal(p, s["positions"])                # a product of two doubles: the same bits
    assert close(m, s["masses"]) and close(v, s["velocities"]) and np.array_equal(labels, s["labels"])
    assert m.dtype == np.float64 and v.shape == (5000, 3)


def test_ellipse_same_seed_same_galaxy_as_the_reference():
    from pm import galaxy_generation_functions as G
    from oracle import pm_galaxy as OG
    e = np.load(os.path.join(GOLD, "galaxy_ellipse_importance_1052_n5000.npz"))
    p, v, m, labels = G.generate_ellipse_importance(5000, 3, 1.8, 1.5, 100000, seed=1052, virial_seed=7)
    assert np.array_equal(p, e["positions"]) and close(m, e["masses"]) and np.array_equal(labels, e["labels"])
    # velocities: the reference's virial_ellipsoid run with seed 7 on these positions and masses
    g = np.load(os.path.join(GOLD, "galaxy_virial_seed7_n5000.npz"))
    assert close(v, g["velocities"], 1e-11)
    rng = np.random.default_rng(7)
    z1, z2 = rng.standard_normal(5000), rng.standard_normal(5000)
    assert close(v, OG.virial_ellipsoid(p, m, 3, 1.8, 1.5, z1, z2), 1e-11)


@pytest.mark.parametrize("n", [2, 3, 257, 100003])
def test_against_the_restatement_at_other_sizes(n):
    from pm import galaxy_generation_functions as G
    from oracle import pm_galaxy as OG
    nrm, q = OG.reference_normals("diffuse_sphere", n, 5)
    po, vo, mo, lo = OG.diffuse_sphere_importance(nrm, q, 2.0, 3.0)
    p, v, m, labels = G.generate_diffuse_sphere_importance(n, 2.0, 3.0, seed=5)
    assert np.array_equal(p, po) and close(m, mo, 1e-11) and close(v, vo, 1e-10) and np.array_equal(labels, lo)
    nrm, z1, z2 = OG.reference_normals("ellipse", n, 5)
    # (reference_normals takes the dispersion deviates from default_rng(seed + 1))
    po, vo, mo, lo = OG.ellipse_importance(nrm, z1, z2, 2.0, 1.0, 0.5, 7.0)
    p, v, m, labels = G.generate_ellipse_importance(n, 2.0, 1.0, 0.5, 7.0, seed=5, virial_seed=6)
    assert np.array_equal(p, po) and close(m, mo, 1e-11) and np.array_equal(labels, lo)
    if n > 2:
        assert close(v, vo, 1e-9)


def test_device_stream_has_the_reference_properties():
    """Deviates drawn on the device: another stream, so only what holds for every draw is checked -- total masses, the
    50/50 split, tangential velocities with the circular speed of the enclosed mass, no bulk motion of the ellipse."""
    from pm import galaxy_generation_functions as G
    n = 200000
    p, v, m, labels = G.generate_diffuse_sphere_importance(n, 1.25, 1.55, seed=3, stream="device")
    star = labels == "1"
    assert star.sum() == n // 2 and abs(m[star].sum() - 0.25 * 1.55) < 1e-9 and abs(m.sum() - 1.55) < 1e-9
    r = np.linalg.norm(p, axis=1)
    assert 0.9 * 1.25 < np.median(r[star]) / 1.538 < 1.1 * 1.25         # median radius of a 3-d Gaussian: 1.538 sigma
    assert np.abs((p * v).sum(axis=1)).max() < 1e-9 * (r * np.linalg.norm(v, axis=1)).max()
    order = np.argsort(r)
    menc = np.empty(n)
    menc[order] = np.cumsum(m[order])
    assert np.allclose(np.linalg.norm(v, axis=1), np.sqrt(menc / r), rtol=1e-9)
    p2, v2, m2, _ = G.generate_diffuse_sphere_importance(n, 1.25, 1.55, seed=3, stream="device")
    assert np.array_equal(p, p2) and np.array_equal(v, v2) and np.array_equal(m, m2)
    p3 = G.generate_diffuse_sphere_importance(n, 1.25, 1.55, seed=4, stream="device")[0]
    assert not np.array_equal(p, p3)
    p, v, m, labels = G.generate_ellipse_importance(n, 3, 1.8, 1.5, 100000, seed=3, stream="device")
    assert abs(m.sum() - 100000) < 1e-6 and np.abs(np.average(v, weights=m, axis=0)).max() < 1e-9 * np.abs(v).max()
    assert abs(p[:, 0].std() / 3 - 1) < 0.02 and abs(p[:, 2].std() / 1.5 - 1) < 0.02


def test_device_stream_at_four_million_particles():
    from pm import galaxy_generation_functions as G
    n = 1 << 22
    p, v, m, labels = G.generate_diffuse_sphere_importance(n, 200.0, 1e6, seed=11, stream="device")
    assert abs(m.sum() / 1e6 - 1) < 1e-9 and np.isfinite(v).all()
    r = np.linalg.norm(p, axis=1)
    k = np.argmax(r)                                   # the outermost particle encloses everything
    assert abs(np.linalg.norm(v[k]) / np.sqrt(m.sum() / r[k]) - 1) < 1e-9


def test_errors():
    from pm import galaxy_generation_functions as G, PMError
    with pytest.raises(PMError):
        G.generate_ellipse_importance(100, -1.0, 1.0, 1.0)
    with pytest.raises(ValueError):
        G.generate_galaxy("spiral")
    with pytest.raises(ValueError):
        G.generate_ellipse_importance(100, 1.0, 1.0, 1.0, stream="pcg")


def test_generated_files_feed_import_galaxy(tmp_path):
    """generate_galaxy writes what main.py:25-27 loads through import_galaxy."""
    from pm import galaxy_generation_functions as G
    from pm import functions as F
    a = G.generate_galaxy("ellipse", "importance", 2000, a=3, b=1.8, c=1.5, M_tot=2000, seed=1, out_dir=str(tmp_path))
    b = G.generate_galaxy("diffuse_sphere", "importance", N_particles=2000, R=1.25, M_tot=1.55, seed=1, out_dir=str(tmp_path))
    assert os.path.basename(a) == "ellipse_importance_1.npz"
    pos, vel, mass, labels = F.import_galaxy(a, b, separation=1000, direction=(0.67, 0.33, 0), velocity=(-3000, 300, 0))
    assert pos.shape == (4000, 3) and mass.dtype == np.float32 and len(labels) == 4000
