"""CPU stand-in for pm.distributed.HipSlabEngine, built on the oracle: test infrastructure only.  It does the local
part of every stage of a slab with the oracle's functions so that the exchange code of pm.distributed.SlabSimulation
can be run, and checked against the single-process oracle, on machines without a GPU."""
import numpy as np
import scipy.fft as sfft
import torch

from oracle import pm_oracle as O
from pm.distributed import slab_of


class OracleSlabEngine:
    def __init__(self, grid_size, box_size, plane0, nplanes):
        self.G, self.L, self.i0, self.Gs = int(grid_size), int(box_size), int(plane0), int(nplanes)
        self.world = self.G // self.Gs
        self.rank = self.i0 // self.Gs
        self.pitch = self.G // 2 + 1
        G, Gs = self.G, self.Gs
        self.density = np.zeros((Gs + 1, G, G), np.float32)
        self.potential = np.zeros((Gs + 3, G, G), np.float32)
        self.spec = np.zeros((Gs, G, self.pitch), np.complex64)
        self._t = torch.zeros((G * Gs * self.pitch, 2), dtype=torch.float32)
        self._out = (np.zeros(0, np.uint32), np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), np.zeros(0, np.float32))

    def close(self):
        pass

    def set_particles(self, pos, vel, mass, ids):
        self.pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        self.vel = np.ascontiguousarray(vel, np.float32).reshape(-1, 3)
        self.mass = np.ascontiguousarray(mass, np.float32)
        self.ids = np.ascontiguousarray(ids, np.uint32)
        assert np.all(slab_of(self.pos, self.G, self.L, self.world) == self.rank)

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
        planes = (self.i0 + np.arange(Gs + 1)) % G
        self.density = full[planes].copy()
        # nothing of a slab's particles lands outside its planes and the one above
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
            self.potential[1:self.Gs + 1] = sfft.irfft2(self.spec, s=(self.G, self.G), axes=(1, 2)).astype(np.float32)

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
        self.potential[1 + first:1 + first + count] = sfft.irfft2(spec, s=(self.G, self.G), axes=(1, 2)).astype(np.float32)

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
        G, Gs = self.G, self.Gs
        full = np.zeros((G, G, G), np.float32)
        for l in range(Gs + 3):
            full[(self.i0 - 1 + l) % G] = self.potential[l]
        if self.n:
            forces = O.force_CIC(O.gradient(full), self.pos, G, self.L)
            x, v = O.kick_drift(self.pos, self.vel, forces, dt, self.L)
        else:
            x, v = self.pos, self.vel
        dest = slab_of(x, G, self.L, self.world) if self.n else np.zeros(0, np.int64)
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
