"""CPU stand-in for pm.distributed.HipSlabEngine, built on the oracle: test infrastructure only.  It does the local
part of every stage of a slab with the oracle's functions so that the exchange code of pm.distributed.SlabSimulation
can be run, and checked against the single-process oracle, on machines without a GPU."""
import numpy as np
import scipy.fft as sfft
import torch

from oracle import pm_oracle as O
from pm.distributed import equal_cuts, owner_by_cuts, plane_of


class OracleSlabEngine:
    """windowed = True: the particles are those of a window of planes set by set_window (the slab to begin with), the
    deposit writes the window's planes into a buffer of its own, and the slab's planes (self.density, what fft2 reads;
    self.mesh_pot, what it writes) are put together by the caller's exchanges -- as HipSlabEngine(window_planes=...) does."""

    def __init__(self, grid_size, box_size, plane0, nplanes, windowed=False, window_planes=None):
        self.G, self.L, self.i0, self.Gs = int(grid_size), int(box_size), int(plane0), int(nplanes)
        self.world = self.G // self.Gs
        self.rank = self.i0 // self.Gs
        self.pitch = self.G // 2 + 1
        G, Gs = self.G, self.Gs
        self.windowed = bool(windowed)
        self.window_planes = int(window_planes) if window_planes else (G - 3 if windowed else Gs)
        self.cuts = equal_cuts(G, self.world)
        self.w0, self.wn = self.i0, Gs
        self.density = np.zeros((Gs + 1, G, G), np.float32)
        self.potential = np.zeros((Gs + 3, G, G), np.float32)
        self.mesh_pot = np.zeros((Gs, G, G), np.float32)
        self.spec = np.zeros((Gs, G, self.pitch), np.complex64)
        self._t = torch.zeros((G * Gs * self.pitch, 2), dtype=torch.float32)
        self._out = (np.zeros(0, np.uint32), np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), np.zeros(0, np.float32))

    def close(self):
        pass

    def owner(self, pos):
        return owner_by_cuts(pos, self.G, self.L, self.cuts)

    def set_particles(self, pos, vel, mass, ids):
        self.pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        self.vel = np.ascontiguousarray(vel, np.float32).reshape(-1, 3)
        self.mass = np.ascontiguousarray(mass, np.float32)
        self.ids = np.ascontiguousarray(ids, np.uint32)
        assert np.all(self.owner(self.pos) == self.rank)

    # ---- windows -------------------------------------------------------------------------------------------
    def set_window(self, cuts):
        assert self.windowed and cuts[0] == 0 and cuts[-1] == self.G and len(cuts) == self.world + 1
        self.cuts = [int(x) for x in cuts]
        self.w0, self.wn = self.cuts[self.rank], self.cuts[self.rank + 1] - self.cuts[self.rank]
        assert 1 <= self.wn <= self.window_planes
        self.potential = np.zeros((self.wn + 3, self.G, self.G), np.float32)

    def rekey(self):
        self._leave(self.pos, self.vel)

    def plane_counts(self):
        return torch.from_numpy(np.bincount(plane_of(self.pos, self.G, self.L), minlength=self.G).astype(np.int64))

    @property
    def deposit_plane_elems(self):
        return self.G * self.G

    def deposit_planes_out(self, first, count):
        return torch.from_numpy(self.wdens[first:first + count].copy()).reshape(-1)

    def mesh_planes_in(self, first, planes, add):
        a = planes.numpy().reshape(-1, self.G, self.G)
        if add:
            self.density[first:first + len(a)] += a
        else:
            self.density[first:first + len(a)] = a

    def own_deposit_planes(self, first, mesh_first, count, add):
        self.mesh_planes_in(mesh_first, torch.from_numpy(self.wdens[first:first + count].copy()), add)

    def mesh_potential_out(self, first, count):
        return torch.from_numpy(self.mesh_pot[first:first + count].copy()).reshape(-1)

    def own_potential_planes(self, mesh_first, first, count):
        self.potential[first:first + count] = self.mesh_pot[mesh_first:mesh_first + count]

    @property
    def n(self):
        return len(self.pos)

    def deposit(self):
        G, Gs = self.G, self.Gs
        zero = np.zeros((G,) * 3, np.float32)
        if Gs == G:
            # one slab: what the particles of the last plane put into plane 0 is the "extra" plane
            i = O.cell_index(self.pos, G, self.L)[:, 0] if self.n else np.zeros(0, np.int64)
            last = i == G - 1
            a = O.CIC(self.pos[~last], G, self.L, self.mass[~last]) if (~last).any() else zero
            b = O.CIC(self.pos[last], G, self.L, self.mass[last]) if last.any() else zero
            self.density[:G] = a + b
            self.density[0] = a[0]
            self.density[G] = b[0]
            return
        full = O.CIC(self.pos, G, self.L, self.mass) if self.n else zero
        planes = (self.w0 + np.arange(self.wn + 1)) % G
        if self.windowed:
            self.wdens = full[planes].copy()                # the caller's exchanges put self.density together from these
            self.density[:] = np.nan                        # (every plane of the slab must be written by them)
        else:
            self.density = full[planes].copy()
        # nothing of a rank's particles lands outside the planes of its window and the one above
        rest = np.ones(G, bool)
        rest[planes] = False
        assert not full[rest].any()

    def _mesh(self, which):
        return self.density if which == 0 else self.potential

    def planes_out(self, which, first, count):
        return torch.from_numpy(self._mesh(which)[first:first + count].copy())

    def planes_in(self, which, first, planes, add=False):
        a = planes.numpy()
        m = self._mesh(which)
        if add:
            m[first:first + a.shape[0]] += a
        else:
            m[first:first + a.shape[0]] = a

    def ghost_out(self):
        return self.planes_out(0, self.Gs, 1).reshape(-1)

    def ghost_in(self, plane):
        self.planes_in(0, 0, plane.reshape(1, self.G, self.G), add=True)

    def fft2(self, direction):
        if direction < 0:
            self.spec = sfft.rfft2(self.density[:self.Gs], axes=(1, 2)).astype(np.complex64)
        else:
            self._potential_planes(0, sfft.irfft2(self.spec, s=(self.G, self.G), axes=(1, 2)).astype(np.float32))

    def _potential_planes(self, first, planes):
        """planes first ... of the slab's potential"""
        if self.windowed:
            self.mesh_pot[first:first + len(planes)] = planes
        else:
            self.potential[1 + first:1 + first + len(planes)] = planes

    def pack(self):
        Gs, w = self.Gs, self.world
        blocks = np.stack([self.spec[:, p * Gs:(p + 1) * Gs, :] for p in range(w)])        # (world, Gs_x, Gs_y, pitch)
        return torch.from_numpy(np.ascontiguousarray(blocks).view(np.float32).reshape(-1, 2))

    def unpack(self, blocks):
        Gs, w = self.Gs, self.world
        b = blocks.numpy().reshape(-1).view(np.complex64).reshape(w, Gs, Gs, self.pitch)
        self.spec = np.concatenate([b[p] for p in range(w)], axis=1)

    def transposed_buffer(self):
        return self._t

    # the transposes in groups of planes (pm_slab_forward_planes / pm_slab_inverse_planes)
    planes_in_turn = True

    def blocks_buffer(self):
        if not hasattr(self, "_b"):
            self._b = torch.zeros((self.G * self.Gs * self.pitch, 2), dtype=torch.float32)
        return self._b

    def forward_planes(self, first, count):
        Gs, w = self.Gs, self.world
        b = self.blocks_buffer().numpy().reshape(-1).view(np.complex64).reshape(w, Gs, Gs, self.pitch)
        spec = sfft.rfft2(self.density[first:first + count], axes=(1, 2)).astype(np.complex64)      # (count, G, pitch)
        for p in range(w):
            b[p, first:first + count] = spec[:, p * Gs:(p + 1) * Gs, :]

    def inverse_planes(self, blocks, first, count):
        Gs, w = self.Gs, self.world
        b = blocks.numpy().reshape(-1).view(np.complex64).reshape(w, Gs, Gs, self.pitch)
        spec = np.concatenate([b[p, first:first + count] for p in range(w)], axis=1)                # (count, G, pitch)
        self._potential_planes(first, sfft.irfft2(spec, s=(self.G, self.G), axes=(1, 2)).astype(np.float32))

    def solve_x(self, spectrum_t):
        G, Gs = self.G, self.Gs
        a = spectrum_t.numpy().reshape(-1).view(np.complex64).reshape(G, Gs, self.pitch)
        j = np.arange(G)
        j = np.where(j < (G + 1) // 2, j, j - G)
        k = (j * (1.0 / G)).astype(np.float32)
        kk = k * k
        ky = kk[self.i0:self.i0 + Gs]
        k2 = (kk[:, None, None] + ky[None, :, None]) + kk[None, None, :self.pitch]
        dc = k2 == 0
        k2[dc] = 1
        pk = sfft.fft(a, axis=0) / k2
        pk[dc] = 0
        a[...] = sfft.ifft(pk, axis=0).astype(np.complex64)

    def gather_kick_drift(self, dt):
        G = self.G
        full = np.full((G, G, G), np.nan, np.float32)       # (planes the gather must not touch)
        for l in range(self.wn + 3):
            full[(self.w0 - 1 + l) % G] = self.potential[l]
        if self.n:
            grad = O.gradient(full)
            forces = O.force_CIC(grad, self.pos, G, self.L)
            assert np.isfinite(forces).all()
            x, v = O.kick_drift(self.pos, self.vel, forces, dt, self.L)
        else:
            x, v = self.pos, self.vel
        self._leave(x, v)

    def _leave(self, x, v):
        """x, v: the live particles' positions and velocities; those whose cell lies in another rank's window go to the outbox."""
        dest = self.owner(x) if self.n else np.zeros(0, np.int64)
        out = dest != self.rank
        order = np.lexsort((self.ids[out], dest[out]))
        self._out = (self.ids[out][order], x[out][order], v[out][order], self.mass[out][order])
        self._counts = [int(np.sum(dest[out] == p)) for p in range(self.world)]
        keep = ~out
        self.pos, self.vel, self.mass, self.ids = x[keep], v[keep], self.mass[keep], self.ids[keep]

    def outbox(self):
        ids, x, v, m = self._out
        rec = np.zeros((len(ids), 8), np.float32)
        rec[:, 0:3] = x
        rec[:, 3] = m
        rec[:, 4:7] = v
        rec[:, 7] = ids.view(np.float32)
        return self._counts, torch.from_numpy(rec)

    def outbox_counts(self):
        return torch.tensor(self._counts + [0], dtype=torch.int64)

    def outbox_records(self, counts):
        assert [int(c) for c in counts] == self._counts
        return self.outbox()[1]

    def inbox(self, records):
        r = records.numpy().reshape(-1, 8)
        if len(r) == 0:
            return
        self.pos = np.vstack((self.pos, r[:, 0:3]))
        self.mass = np.concatenate((self.mass, r[:, 3]))
        self.vel = np.vstack((self.vel, r[:, 4:7]))
        self.ids = np.concatenate((self.ids, np.ascontiguousarray(r[:, 7]).view(np.uint32)))

    def particles(self):
        return self.ids.copy(), self.pos.copy(), self.vel.copy()

    def synchronize(self):
        pass
