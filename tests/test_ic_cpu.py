"""CPU tests of the synthetic particle sets: the random stream against its known-answer vectors, the restated
distributions against their closed forms, the host-side parameter lists."""
import numpy as np
import pytest

from oracle import pm_ic as I

L = 10000.0

# known-answer vectors of Philox4x32-10 (Random123 distribution, kat_vectors): counter, key, output
KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


@pytest.mark.parametrize("ctr,key,out", KAT)
def test_philox_known_answers(ctr, key, out):
    r = I.philox4x32(*[np.array([c]) for c in ctr], *key)
    assert tuple(int(x[0]) for x in r) == out


def test_sets_are_functions_of_index_and_seed():
    # a particle's draw depends on its index only: a longer uniform set starts with the shorter one
    a = I.generate(I.UNIFORM, 1000, 7, L)
    b = I.generate(I.UNIFORM, 3000, 7, L)
    assert np.array_equal(a[0], b[0][:1000]) and np.array_equal(a[1], b[1][:1000])
    c = I.generate(I.UNIFORM, 1000, 8, L)
    assert not np.array_equal(a[0], c[0])


def test_uniform_moments():
    x, v, m = I.generate(I.UNIFORM, 400_000, 1, L)
    assert x.dtype == v.dtype == m.dtype == np.float32 and x.min() >= 0 and x.max() <= L
    assert np.allclose(x.mean(0), L / 2, atol=0.005 * L) and np.allclose(x.std(0), L / np.sqrt(12), rtol=0.01)
    assert np.allclose(v.mean(0), 0, atol=0.1) and np.allclose(v.std(0), 10.0, rtol=0.01)
    assert np.all(m == 1)


def test_two_galaxy_profile_and_momentum():
    n = 400_000
    q = I.default_params(I.TWO_GALAXY, L)
    x, v, _ = I.generate(I.TWO_GALAXY, n, 1052, L)
    for half, a, c in ((slice(0, n // 2), q[0], q[3:6]), (slice(n // 2, n), q[1], q[6:9])):
        # the set is wrapped into the periodic box: distances by the nearest image
        sep = (x[half].astype(np.float64) - np.array(c) + L / 2) % L - L / 2
        r = np.linalg.norm(sep, axis=1)
        # Plummer: M(<r)/M = (1 + a^2/r^2)^-3/2; the truncation at 0.45 L removes up to 2 % of the mass
        half_mass = 0.5 * (1.0 + (a / (0.45 * L)) ** 2) ** -1.5        # half of the mass inside the truncation radius
        assert np.median(r) == pytest.approx(a / np.sqrt(half_mass ** (-2.0 / 3.0) - 1.0), rel=0.01)
        assert np.mean(r < a) == pytest.approx(2.0 ** -1.5 / (2.0 * half_mass), abs=0.005)
        assert r.max() < 0.45 * L * (1 + 1e-6)
        # isotropy
        d = sep / r[:, None]
        assert np.allclose(d.mean(0), 0, atol=0.01)
    # functions.py:329-338: the two bulk velocities differ by the relative velocity and cancel
    v1, v2 = v[: n // 2].mean(0, dtype=np.float64), v[n // 2:].mean(0, dtype=np.float64)
    assert np.allclose(v2 - v1, q[9:12], atol=0.2) and np.allclose(v1 + v2, 0, atol=0.2)


def test_disk_profile():
    n = 400_000
    Rd, z0, Rmax, zmax = I.default_params(I.DISK, L)[:4]
    x, _, _ = I.generate(I.DISK, n, 42, L)
    d = x.astype(np.float64) - L / 2
    R, z = np.hypot(d[:, 0], d[:, 1]), d[:, 2]
    assert R.max() <= Rmax * (1 + 1e-6) and np.abs(z).max() <= zmax * (1 + 1e-5)
    # truncated Gamma(2, Rd): cdf(R) = (1 - e^-t (1 + t)) / (1 - e^-5 * 6) with t = R / Rd
    norm = 1 - np.exp(-5.0) * 6.0
    for t in (0.5, 1.0, 2.0, 3.0):
        assert np.mean(R < t * Rd) == pytest.approx((1 - np.exp(-t) * (1 + t)) / norm, abs=0.005)
    # truncated Laplace: P(|z| < t z0) = (1 - e^-t) / (1 - e^-5)
    for t in (0.5, 1.0, 2.0):
        assert np.mean(np.abs(z) < t * z0) == pytest.approx((1 - np.exp(-t)) / (1 - np.exp(-5.0)), abs=0.005)
    assert abs(np.mean(z)) < 0.02 * z0
    phi = np.arctan2(d[:, 1], d[:, 0])
    assert np.allclose(np.histogram(phi, 8, (-np.pi, np.pi))[0] / n, 0.125, atol=0.004)


def test_rejection_runs_out_of_attempts_deterministically():
    # a truncation radius far inside the scale radius: nearly every attempt is rejected, the fallback is the radius itself
    q = I.default_params(I.TWO_GALAXY, L)
    q[2] = 1e-3 * q[0]
    x, _, _ = I.generate(I.TWO_GALAXY, 2000, 3, L, q)
    r = np.linalg.norm(x[:1000].astype(np.float64) - np.array(q[3:6]), axis=1)
    assert r.max() <= q[2] * 1.01 + 1e-3


def test_host_parameter_lists_match_the_oracle_defaults():
    from pm.simulation import ic_parameters
    for name, kind in (("uniform", I.UNIFORM), ("two_galaxy", I.TWO_GALAXY), ("disk", I.DISK)):
        assert np.allclose(ic_parameters(name, L), I.default_params(kind, L), rtol=0, atol=0)
    q = ic_parameters("disk", L, Rd=100.0, centre=(1, 2, 3))
    assert q[0] == 100.0 and q[2] == 500.0 and q[4:7] == [1.0, 2.0, 3.0]
    with pytest.raises(TypeError):
        ic_parameters("uniform", L, Rd=1.0)
    with pytest.raises(ValueError):
        ic_parameters("ring", L)
