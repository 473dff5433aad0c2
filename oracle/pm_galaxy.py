"""CPU restatement (numpy, float64) of the reference's importance-sampled galaxy generators, split where the random
numbers enter: every function takes the standard normal deviates it consumes as arrays, in the order the reference
draws them from its NumPy Generator, and is otherwise deterministic.

TEST INFRASTRUCTURE: imported by tests/ only; the product never imports it.

Follows galaxy_generation_functions.py:
  ellipse_importance   :181-244  (positions :215, weights :222-236, velocities through virial_ellipsoid)
  virial_ellipsoid     :9-94
  diffuse_sphere_importance :267-383 (positions :302-304, Plummer weights :308-313, NFW weights :316-322,
                        enclosed mass and circular speed :336-350, tangential directions :353-381)

Pinned by tests/golden/galaxy_ellipse_importance_1052_n5000.npz, galaxy_diffuse_sphere_importance_1052_n5000.npz (files
the reference's generate_galaxy wrote, tools/make_golden.py) and galaxy_virial_seed7_n5000.npz (virial_ellipsoid called
with a seed; inside generate_ellipse_importance the reference calls it without one, so the velocities of the first
file cannot be reproduced by anyone): tests/test_galaxy_cpu.py.
"""
import numpy as np

EPS = 1e-12
STAR, DARK = "1", "2"


def reference_normals(kind, n, seed):
    """The standard normal deviates the reference's generator consumes for `seed`, as NumPy's default_rng hands them
    out.  ellipse: (n,3) for the positions (galaxy_generation_functions.py:215; rng.normal(0, sigmas) scales standard
    deviates); the two dispersion arrays z1, z2 come from a second, unseeded generator (:25, 76-77) and are drawn here
    from default_rng(seed + 1) instead.  diffuse_sphere: (n_star,3), (n_dark,3) positions (:303-304), then (n,3)
    directions (:357)."""
    rng = np.random.default_rng(seed)
    if kind == "ellipse":
        pos = rng.standard_normal((n, 3))
        rng2 = np.random.default_rng(seed + 1)
        return pos, rng2.standard_normal(n), rng2.standard_normal(n)
    n_star = int(0.5 * n)
    star = rng.standard_normal((n_star, 3))
    dark = rng.standard_normal((n - n_star, 3))
    q = rng.standard_normal((n, 3))
    return np.vstack((star, dark)), q


def virial_ellipsoid(pos, mass, a, b, c, z1, z2, G=1.0, disp=0.1):
    """galaxy_generation_functions.py:9-94 with the default spin axis; z1, z2: the two standard normal arrays of :76-77."""
    x = pos - np.average(pos, weights=mass, axis=0)
    inv2 = np.array([1.0 / (a * a), 1.0 / (b * b), 1.0 / (c * c)])
    m = np.sqrt(np.maximum((x * x * inv2).sum(axis=1), EPS))
    r = np.maximum(m * (abs(a * b * c)) ** (1.0 / 3.0), EPS)
    order = np.argsort(r)
    Menc = np.empty_like(r)
    Menc[order] = np.cumsum(mass[order])
    v_c = np.sqrt(G * Menc / r)
    n = x * inv2
    n /= (np.linalg.norm(n, axis=1, keepdims=True) + EPS)
    k = np.array([0.0, 0.0, 1.0])
    k /= (np.linalg.norm(k) + EPS)
    k = np.broadcast_to(k, x.shape)
    t1 = np.cross(k, n)
    t1_norm = np.linalg.norm(t1, axis=1, keepdims=True)
    bad = t1_norm[:, 0] < 1e-8
    if np.any(bad):
        alt = np.array([0.0, 1.0, 0.0])
        t1_alt = np.cross(np.broadcast_to(alt, x[bad].shape), n[bad])
        t1_alt /= (np.linalg.norm(t1_alt, axis=1, keepdims=True) + EPS)
        t1[bad] = t1_alt
        t1_norm = np.linalg.norm(t1, axis=1, keepdims=True)
    t1 /= (t1_norm + EPS)
    t2 = np.cross(n, t1)
    v = t1 * v_c[:, None] + (t1 * z1[:, None] + t2 * z2[:, None]) * (disp * v_c)[:, None]
    v -= n * (v * n).sum(axis=1, keepdims=True)
    v -= np.average(v, weights=mass, axis=0)
    num = np.sum(mass * v_c * v_c)
    den = np.sum(mass * (v * v).sum(axis=1))
    if den > EPS:
        v *= np.sqrt(num / den)
    return v


def ellipse_importance(normals, z1, z2, a, b, c, M_tot=1.0):
    """galaxy_generation_functions.py:181-244: positions = normals * (a, b, c), masses from the weights
    exp(-m) / exp(-m^2/2) normalised to M_tot, velocities from virial_ellipsoid."""
    sigmas = np.array([a, b, c], dtype=np.float64)
    positions = 0.0 + sigmas * normals                       # Generator.normal(loc, scale): loc + scale * deviate
    inv = np.array([1.0 / (a * a), 1.0 / (b * b), 1.0 / (c * c)])
    m2 = np.einsum("ij,j,ij->i", positions, inv, positions)
    m = np.sqrt(m2)
    log_w = -0.5 * m2 - (-m)
    log_w -= np.max(log_w)
    w = np.exp(log_w)
    w /= np.sum(w)
    masses = M_tot * w
    labels = np.array([STAR] * len(positions), dtype="U5")
    return positions, virial_ellipsoid(positions, masses, a, b, c, z1, z2), masses, labels


def diffuse_sphere_importance(normals, q, R, M_tot, stellar_fraction=0.25, R_dm=390.0):
    """galaxy_generation_functions.py:267-383.  normals: (n,3), the first int(0.5 n) rows for the stars; q: (n,3)."""
    n = len(normals)
    n_star = int(0.5 * n)
    labels = np.array([STAR] * n_star + [DARK] * (n - n_star), dtype="U5")
    star = labels == STAR
    dark = ~star
    positions = np.empty((n, 3))
    positions[star] = 0.0 + R * normals[:n_star]
    positions[dark] = 0.0 + R_dm * normals[n_star:]
    M_stellar, M_dark = stellar_fraction * M_tot, (1 - stellar_fraction) * M_tot
    r_star = np.linalg.norm(positions[star], axis=1)
    x_star = r_star / R
    lw = -2.5 * np.log1p(x_star * x_star) + 0.5 * (x_star * x_star)
    lw -= lw.max()
    w = np.exp(lw)
    m_star = M_stellar * w / np.sum(w)
    r_dark = np.linalg.norm(positions[dark], axis=1)
    x_dark = np.clip(r_dark / R_dm, 1e-12, None)
    lw = -np.log(x_dark) - 2.0 * np.log1p(x_dark) + 0.5 * (x_dark * x_dark)
    lw -= lw.max()
    w = np.exp(lw)
    m_dark = M_dark * w / np.sum(w)
    masses = np.empty(n)
    masses[star] = m_star
    masses[dark] = m_dark
    r = np.empty(n)
    r[star] = r_star
    r[dark] = r_dark
    idx = np.argsort(r)
    Menc_sorted = np.cumsum(masses[idx])
    v_sorted = np.zeros(n)
    nz = r[idx] > 0
    v_sorted[nz] = np.sqrt(Menc_sorted[nz] / r[idx][nz])
    vmag = np.empty(n)
    vmag[idx] = v_sorted
    rhat = np.zeros_like(positions)
    safe = r > 0
    rhat[safe] = positions[safe] / r[safe, None]
    q = np.array(q, dtype=np.float64)
    tang = np.cross(rhat, q)
    tn = np.linalg.norm(tang, axis=1)
    fix = tn < 1e-12
    if np.any(fix):
        q[fix] = np.array([1.0, 0.0, 0.0])
        tang[fix] = np.cross(rhat[fix], q[fix])
        tn[fix] = np.linalg.norm(tang[fix], axis=1)
    # (r = 0 exactly draws a fresh direction in the reference, :372-376; its speed is zero either way)
    tn = np.where(safe, tn, 1.0)
    tang = tang / tn[:, None]
    return positions, tang * vmag[:, None], masses, labels
