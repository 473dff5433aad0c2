"""CPU restatement (numpy) of the device-side synthetic particle sets of libpm.so (pm_generate).

TEST INFRASTRUCTURE: imported by tests/ only; the product never imports it.

What is pinned and what is not.  The random stream is Philox4x32-10, checked here against the known-answer vectors
of the Random123 distribution (tests/test_ic_cpu.py).  The sampling recipes restate the *distributions* of the
reference's generators -- galaxy_generation_functions.py:97-136 (exponential disk: R ~ Gamma(2, Rd), z ~ Laplace(0, z0),
rejection at Rmax / zmax, uniform azimuth), :267-288 (Plummer profile) and the two-galaxy placement of
functions.py:320-338 -- but not their NumPy PCG64 streams, which are consumed array by array and cannot be drawn
particle by particle.  Against the reference itself these sets are therefore **parity unpinned** (they agree in
distribution only, which the tests check through moments and quantiles); against this restatement the device must
agree to the last bit of float64 rounding of log / pow / sin / cos, i.e. to one float32 ulp after the cast.
"""
import numpy as np

UNIFORM, TWO_GALAXY, DISK = 0, 1, 2
MAX_ATTEMPTS = 64
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=10):
    """Philox4x32 on arrays of counter words (uint32-valued) with one key (k0, k1); returns four uint64 arrays
    holding 32-bit values."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & _MASK for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(rounds):
        p0, p1 = _M0 * c0, _M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> _S32, p0 & _MASK, p1 >> _S32, p1 & _MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def _unit(x):
    return (x.astype(np.float64) + 0.5) * (1.0 / 4294967296.0)


def _draw(idx, purpose, attempt, seed):
    lo, hi = idx & 0xFFFFFFFF, idx >> 32
    return philox4x32(lo, hi, np.full_like(idx, purpose), np.full_like(idx, attempt), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)


def _first_accepted(idx, purpose, seed, sample, accept, fallback):
    """Per particle: the first attempt t = 0, 1, ... whose sample is accepted.  Returns (value, block of that
    attempt); particles that run out of attempts take `fallback` with the block of the last attempt."""
    n = len(idx)
    value = np.full(n, fallback, np.float64)
    block = [np.zeros(n, np.uint64) for _ in range(4)]
    todo = np.ones(n, bool)
    for t in range(MAX_ATTEMPTS):
        if not todo.any():
            break
        where = np.nonzero(todo)[0]
        b = _draw(idx[where], purpose, t, seed)
        v = sample(b)
        ok = accept(v)
        for k in range(4):
            block[k][where] = b[k]
        value[where[ok]] = v[ok]
        todo[where[ok]] = False
    return value, block


def default_params(kind, L):
    """The sets bench.py measures, as parameter lists of pm_generate."""
    c = 0.5 * L
    if kind == UNIFORM:
        return [10.0]
    if kind == TWO_GALAXY:
        d = np.array([0.67, 0.33, 0.0])
        d /= np.linalg.norm(d)
        return [0.02 * L, 0.05 * L, 0.45 * L, *(c - 0.1 * L * d), *(c + 0.1 * L * d), -3000.0, 300.0, 0.0, 10.0]
    if kind == DISK:
        return [0.03 * L, 0.003 * L, 0.15 * L, 0.015 * L, c, c, c, 10.0]
    raise ValueError(kind)


def generate(kind, n, seed, L, params=None):
    """(positions float32 (n,3), velocities float32 (n,3), masses float32 (n,)) of pm_generate(kind, n, seed, params)."""
    q = [float(v) for v in (default_params(kind, L) if params is None else params)]
    idx = np.arange(n, dtype=np.uint64)
    seed = int(seed)
    bulk = np.zeros((n, 3))
    with np.errstate(divide="ignore", invalid="ignore"):
        if kind == UNIFORM:
            b = _draw(idx, 0, 0, seed)
            pos = np.stack([_unit(b[0]) * L, _unit(b[1]) * L, _unit(b[2]) * L], axis=1)
            sigma = q[0]
        elif kind == TWO_GALAXY:
            second = idx >= np.uint64(n // 2)
            a = np.where(second, q[1], q[0])
            rmax = q[2]
            centre = np.where(second[:, None], np.array(q[6:9])[None, :], np.array(q[3:6])[None, :])
            r = np.empty(n)
            blk = [np.empty(n, np.uint64) for _ in range(4)]
            for g in (False, True):             # the scale radius differs between the galaxies
                m = second == g
                ag = q[1] if g else q[0]
                rg, bg = _first_accepted(idx[m], 0, seed, lambda b: ag / np.sqrt(np.power(_unit(b[0]), -2.0 / 3.0) - 1.0),
                                         lambda v: v < rmax, rmax)
                r[m] = rg
                for k in range(4):
                    blk[k][m] = bg[k]
            cz = 1.0 - 2.0 * _unit(blk[1])
            phi = 2.0 * np.pi * _unit(blk[2])
            s = np.sqrt(np.maximum(0.0, 1.0 - cz * cz))
            pos = centre + np.stack([r * (s * np.cos(phi)), r * (s * np.sin(phi)), r * cz], axis=1)
            h = np.where(second, 0.5, -0.5)
            bulk = h[:, None] * np.array(q[9:12])[None, :]
            sigma = q[12]
        elif kind == DISK:
            Rd, z0, Rmax, zmax = q[0:4]
            R, blk = _first_accepted(idx, 0, seed, lambda b: -Rd * np.log(_unit(b[0]) * _unit(b[1])), lambda v: v <= Rmax, Rmax)
            phi = 2.0 * np.pi * _unit(blk[2])

            def height(b):
                w = _unit(b[0])
                return np.where(w < 0.5, z0 * np.log(2.0 * w), -z0 * np.log(2.0 * (1.0 - w)))
            h, _ = _first_accepted(idx, 2, seed, height, lambda v: np.abs(v) <= zmax, 0.0)
            pos = np.stack([q[4] + R * np.cos(phi), q[5] + R * np.sin(phi), q[6] + h], axis=1)
            sigma = q[7]
        else:
            raise ValueError(kind)
    b = _draw(idx, 1, 0, seed)
    m0, a0 = np.sqrt(-2.0 * np.log(_unit(b[0]))), 2.0 * np.pi * _unit(b[1])
    m1, a1 = np.sqrt(-2.0 * np.log(_unit(b[2]))), 2.0 * np.pi * _unit(b[3])
    vel = bulk + np.stack([sigma * (m0 * np.cos(a0)), sigma * (m0 * np.sin(a0)), sigma * (m1 * np.cos(a1))], axis=1)
    pos32 = np.remainder(pos.astype(np.float32), np.float32(L))        # main.py:60: float32 remainder
    return pos32, vel.astype(np.float32), np.ones(n, np.float32)
