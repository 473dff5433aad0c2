"""A particle set whose PM step is known from a two-dimensional computation, at any mesh size: test infrastructure.

Particles sit at the same offset FRAC inside their cells, one in every cell (i, j, k) with k a multiple of `kstride`,
with masses that depend on i only.  The cloud-in-cell density (functions.py:19-70) is then A(i) * C(k), independent of
j; so are the potential (functions.py:73-89: only ky = 0 is excited) and its np.gradient (main.py:54), and force_CIC
(functions.py:99-180) at a particle is a four-corner sum in the (i, k) plane.  All of it is computed here in float64
on a G x G array -- for a 2048^3 mesh, which no single array of the reference's own code could hold on the test box."""
import numpy as np

FRAC = 0.25


def plane_masses(i, G, mode, eps, scale):
    return (scale * (1.0 + eps * np.cos(2.0 * np.pi * mode * np.asarray(i, np.float64) / G))).astype(np.float32)


def slab_particles(G, L, kstride, mode, eps, scale, v0, plane0, nplanes):
    """Positions, velocities, masses and numbers of the particles in mesh planes [plane0, plane0 + nplanes).  L / G must
    be a binary fraction (L = 10000 with a power-of-two G is): the positions are then exact in float32."""
    h = L / G
    i = np.arange(plane0, plane0 + nplanes, dtype=np.int64)
    j = np.arange(G, dtype=np.int64)
    k = np.arange(0, G, kstride, dtype=np.int64)
    I, J, K = (a.reshape(-1) for a in np.meshgrid(i, j, k, indexing="ij"))
    pos = np.empty((I.size, 3), np.float32)
    pos[:, 0], pos[:, 1], pos[:, 2] = (I + FRAC) * h, (J + FRAC) * h, (K + FRAC) * h
    assert np.array_equal(pos[:, 0].astype(np.float64), (I + FRAC) * h)
    vel = np.zeros((I.size, 3), np.float32)
    vel[:, 0] = np.float32(v0)
    ids = ((I * G + J) * (G // kstride) + K // kstride).astype(np.uint32)
    return pos, vel, plane_masses(I, G, mode, eps, scale), ids


def cells_of(ids, G, kstride):
    """(i, j, k) of the cell a particle started in, from its number."""
    nk = G // kstride
    ids = ids.astype(np.int64)
    return ids // (G * nk), (ids // nk) % G, (ids % nk) * kstride


def forces_2d(G, kstride, mode, eps, scale):
    """(Fx, Fz) as (G, G) float64 arrays indexed by the particle's cell (i, k); Fy = 0."""
    m = plane_masses(np.arange(G), G, mode, eps, scale).astype(np.float64)
    A = (1.0 - FRAC) * m + FRAC * np.roll(m, 1)                   # functions.py:45-68 along i: node n gets cell n and cell n - 1
    comb = np.zeros(G)
    comb[::kstride] = 1.0
    C = (1.0 - FRAC) * comb + FRAC * np.roll(comb, 1)
    rho = np.outer(A, C)                                           # the weights along j add up to one
    k = np.fft.fftfreq(G).astype(np.float32).astype(np.float64)   # functions.py:75
    k2 = k[:, None] ** 2 + k[None, :] ** 2
    k2[0, 0] = 1.0
    pk = np.fft.fft2(rho - rho.mean()) / k2                       # functions.py:81-85 on the ky = 0 plane
    pk[0, 0] = 0.0
    phi = np.fft.ifft2(pk).real
    gx, gz = np.gradient(phi)                                      # main.py:54, one-sided on the faces
    w = (1.0 - FRAC, FRAC)
    Fx = sum(w[a] * w[c] * np.roll(np.roll(gx, -a, 0), -c, 1) for a in (0, 1) for c in (0, 1))     # functions.py:113-178
    Fz = sum(w[a] * w[c] * np.roll(np.roll(gz, -a, 0), -c, 1) for a in (0, 1) for c in (0, 1))
    return Fx, Fz


def expected_velocities(ids, G, kstride, mode, eps, scale, v0, dt):
    """main.py:58 after one step, float64: v0 + forces * dt."""
    Fx, Fz = forces_2d(G, kstride, mode, eps, scale)
    i, _, k = cells_of(ids, G, kstride)
    v = np.zeros((len(ids), 3))
    v[:, 0] = np.float64(np.float32(v0)) + Fx[i, k] * dt
    v[:, 2] = Fz[i, k] * dt
    return v, max(np.abs(Fx).max(), np.abs(Fz).max()) * dt


def scale_for(G, kstride, mode, eps, v0, dt, share=0.05):
    """A power of two for `scale` that makes the largest kick about `share` of v0: large against the float32 spacing
    of v0 + kick, small enough that the drift of a step stays far inside a cell."""
    Fx, Fz = forces_2d(G, kstride, mode, eps, 1.0)
    kick = max(np.abs(Fx).max(), np.abs(Fz).max()) * dt
    return float(2.0 ** np.round(np.log2(share * abs(v0) / kick)))
