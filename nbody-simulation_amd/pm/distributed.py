"""The PM step over several GPUs: one process per GPU, the mesh cut into slabs of planes along axis 0.

Rank r owns mesh planes [r*Gs, (r+1)*Gs), Gs = G / world, and the particles whose cell (functions.py:27-39)
has its i in that range.  One step of main.py:50-60 becomes

    deposit (local)                    -> the extra plane goes to the next rank, which adds it
    FFT along z, y (local)             -> all-to-all: every rank gets a range of y rows of every plane
    FFT along x, 1/k^2, inverse x      -> all-to-all back
    inverse FFT along y, z (local)     -> one potential plane from the previous rank, two from the next
    gradient + gather + kick + drift   -> particles whose new cell lies in another slab move there

With `transpose_chunks` = C > 1 the two all-to-alls are cut into C groups of planes: going out, the all-to-all of group c
(asynchronous, on the collective's own stream) runs while the transforms of group c + 1 do; coming back, all C
all-to-alls are posted after the solve along x, and the inverse transforms of group c start as soon as its all-to-all
has arrived, beside the all-to-alls of the groups after it.

The local work is done by an engine: `HipSlabEngine` drives libpm.so (include/pm.h, the pm_slab_* calls)
on device buffers held as torch tensors; the exchanges are torch.distributed collectives on those tensors
(backend "nccl" = RCCL over xGMI; with "gloo" the tensors are staged through the host).  An engine is
anything with the methods of HipSlabEngine; the tests drive the same exchange code with a CPU engine.
"""
import ctypes
import threading
import time

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


def plane_of(positions, grid_size, box_size):
    """Plane i of each particle's cell (functions.py:27-39, float64 like numba)."""
    x = np.asarray(positions)[:, 0].astype(np.float64)
    i = ((x % float(box_size)) * (grid_size / float(box_size))).astype(np.int64)
    i[i >= grid_size] -= grid_size
    return i


def slab_of(positions, grid_size, box_size, world):
    """Rank that owns each particle when every rank's particles are those of its slab: plane i of the cell // (G / world)."""
    return plane_of(positions, grid_size, box_size) // (grid_size // world)


def equal_cuts(grid_size, world):
    """First plane of every rank's window when the windows are the slabs themselves, and G behind them."""
    return [r * (grid_size // world) for r in range(world + 1)]


def balanced_cuts(plane_counts, world, max_planes=None, plane_weight=0.0):
    """Windows of (nearly) equal particle count: cuts[r] = first plane of rank r's window, cuts[world] = G, from the
    number of particles in every plane of the mesh.  Rank r's window ends at the plane boundary nearest to r + 1 shares
    of the particles; every window has at least one plane and at most max_planes.  A plane is not divided, so a plane
    that holds more than a share of the particles bounds the balance that can be had.

    plane_weight: what a plane of the window costs its rank, in particles -- its rows are deposited, sent to the slab that
    transforms them and their potential is brought back whether they hold particles or not (measured on 512^3: a plane
    costs about as much as half the mean number of particles per plane).  The shares are then of particles + plane_weight
    per plane, so that a rank in the empty outskirts of a clustered set gets fewer particles to make up for its planes."""
    counts = np.asarray(plane_counts, dtype=np.float64) + float(plane_weight)
    G = len(counts)
    wmax = G - 3 if max_planes is None else int(max_planes)
    if world * wmax < G or world > G:
        raise ValueError(f"{world} windows of at most {wmax} planes cannot cover {G} planes")
    before = np.concatenate(([0.0], np.cumsum(counts)))          # before[b]: particles in planes < b
    cuts = [0]
    for r in range(1, world):
        b = int(np.argmin(np.abs(before - before[-1] * r / world)))
        lo = max(cuts[-1] + 1, G - (world - r) * wmax)            # the ranks after this one can still cover the rest
        hi = min(cuts[-1] + wmax, G - (world - r))                # ... and have a plane each
        cuts.append(min(max(b, lo), hi))
    cuts.append(G)
    return cuts


def owner_by_cuts(positions, grid_size, box_size, cuts):
    """Rank whose window [cuts[r], cuts[r + 1]) holds the plane of each particle's cell."""
    return np.searchsorted(np.asarray(cuts), plane_of(positions, grid_size, box_size), side="right") - 1


def plane_segments(cuts, grid_size, rank, below, above):
    """The planes cuts[rank] - below .. cuts[rank + 1] - 1 + above (modulo G) of rank's window and its surroundings, as
    runs that lie in one slab each: [(t0, count, owner, first)] -- buffer planes t0 .. t0 + count - 1 (t counts from the
    first of these planes) are planes first .. of the slab of rank `owner`.  The last `above` planes are runs of their
    own (they are what the deposit adds to other windows' planes)."""
    world = len(cuts) - 1
    Gs = grid_size // world
    w0, wn = cuts[rank], cuts[rank + 1] - cuts[rank]
    total = below + wn + above
    segs = []
    for t in range(total):
        p = (w0 - below + t) % grid_size
        owner, first = p // Gs, p % Gs
        new_run = not segs or segs[-1][2] != owner or segs[-1][3] + segs[-1][1] != first or t == below + wn
        if new_run:
            segs.append([t, 1, owner, first])
        else:
            segs[-1][1] += 1
    return [tuple(x) for x in segs]


def generated_plane_counts(grid_size, box_size, kind, n, seed=0, device_index=0, **params):
    """Particles of the synthetic set (kind, n, seed) in every mesh plane (host int64 array of G), drawn on the device without
    keeping anything (pm_generate_plane_counts): the same on every rank, so that windows can be cut from it before any
    context is made and without a collective."""
    from .simulation import ic_parameters
    lib = _lib.load()
    arr = np.asarray(ic_parameters(kind, float(box_size), **params), dtype=np.float64)
    out = np.zeros(int(grid_size), np.int64)
    rc = lib.pm_generate_plane_counts(int(device_index), int(grid_size), float(box_size), _lib.IC_KINDS[kind], int(n), int(seed) & (2 ** 64 - 1),
                                      arr.ctypes.data, len(arr), out.ctypes.data)
    if rc != 0:
        raise _lib.PMError(f"pm_generate_plane_counts: error {rc}: " + (lib.pm_last_error(None) or b"?").decode())
    return out


class HipSlabEngine:
    """Local stages of one slab on one GPU through libpm.so."""

    def __init__(self, grid_size, box_size, plane0, nplanes, capacity, device_index=0, rocfft_only=False, full_sort=False, window_planes=None):
        """full_sort: PM_FLAG_FULL_SORT (every binning sorts the whole slab; for comparisons).  window_planes: room for a
        window of that many planes of particles (pm_create_slab_windowed); None: the particles are those of the slab."""
        self.lib = _lib.load()
        self.G, self.L, self.i0, self.Gs = int(grid_size), float(box_size), int(plane0), int(nplanes)
        self.world = self.G // self.Gs
        self.rank = self.i0 // self.Gs
        torch.cuda.set_device(device_index)
        self.device = torch.device("cuda", device_index)
        self.cap = int(capacity)
        # FLAG_FUSED_GRADIENT on a slab: where the hand-written transforms apply, the deposit writes its planes already
        # transformed along z (no density mesh); elsewhere the flag changes nothing
        flags = _lib.FLAG_FUSED_GRADIENT | (_lib.FLAG_ROCFFT_ONLY if rocfft_only else 0) | (_lib.FLAG_FULL_SORT if full_sort else 0)
        self._flags = flags
        self.windowed = window_planes is not None and int(window_planes) != self.Gs
        if self.windowed:
            self.ctx = self.lib.pm_create_slab_windowed(self.G, self.L, self.cap, device_index, flags, self.i0, self.Gs, int(window_planes))
        else:
            self.ctx = self.lib.pm_create_slab(self.G, self.L, self.cap, device_index, flags, self.i0, self.Gs)
        if not self.ctx:
            raise _lib.PMError("pm_create_slab failed: " + (self.lib.pm_last_error(None) or b"?").decode())
        self.window_planes = int(window_planes) if self.windowed else self.Gs
        self.w0, self.wn = self.i0, self.Gs                      # the window: the slab until set_window says otherwise
        # the library works on torch's current stream: its kernels, torch's copies and the collectives order themselves
        self._check(self.lib.pm_set_stream(self.ctx, ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        self.pitch = self.lib.pm_slab_pitch(self.ctx)
        f32 = dict(dtype=torch.float32, device=self.device)
        self._blocks = [torch.empty((self.Gs * self.G * self.pitch, 2), **f32) for _ in range(2)]
        self._records = torch.empty((self.cap if self.windowed else self.cap // 4 + 1024, 8), **f32)
        self._counts = torch.zeros(self.world + 1, dtype=torch.int64, device=self.device)
        # what the deposit leaves: z-transformed planes (G rows of `pitch` complex64) or density planes (G rows of G floats)
        probe = torch.empty(2 * self.G * self.pitch, **f32)
        self.spectral = self.lib.pm_slab_planes(self.ctx, _lib.SLAB_ZSPECTRUM, 0, 1, probe.data_ptr(), 0) == 0

    def close(self):
        if self.ctx:
            self.lib.pm_destroy(self.ctx)
            self.ctx = None

    def last_binning(self):
        """(from the order before?, movers) of the slab's last binning (pm_last_binning)"""
        movers = ctypes.c_int64(0)
        rc = self.lib.pm_last_binning(self.ctx, ctypes.byref(movers))
        if rc < 0:
            self._check(-rc)
        return bool(rc), int(movers.value)

    def set_timing(self, on):
        """PM_FLAG_TIMING on the slab's context: HIP events around its kernels (kernel_times)."""
        flags = (self._flags & (_lib.FLAG_FUSED_GRADIENT | _lib.FLAG_FULL_SORT)) | (_lib.FLAG_TIMING if on else 0)
        self._check(self.lib.pm_set_flags(self.ctx, flags))

    def kernel_times(self):
        """{kernel: (milliseconds, launches)} of this slab's context since the last call (pm_kernel_times; needs set_timing(True))"""
        ms = (ctypes.c_float * len(_lib.KERNELS))()
        calls = (ctypes.c_int * len(_lib.KERNELS))()
        self._check(self.lib.pm_kernel_times(self.ctx, ms, calls))
        return {k: (float(ms[i]), int(calls[i])) for i, k in enumerate(_lib.KERNELS)}

    def _check(self, rc):
        _lib.check(self.ctx, rc)

    def _sync(self):
        pass        # nothing to wait for: one stream carries torch's work and the library's

    @property
    def n(self):
        return int(self.lib.pm_num_particles(self.ctx))

    def set_particles(self, pos, vel, mass, ids):
        pos = np.ascontiguousarray(pos, np.float32)
        vel = np.ascontiguousarray(vel, np.float32)
        mass = np.ascontiguousarray(mass, np.float32)
        ids = np.ascontiguousarray(ids, np.uint32)
        if len(pos) > self.cap:
            raise ValueError("more particles than the engine's capacity")
        self._check(self.lib.pm_slab_set_particles(self.ctx, pos.ctypes.data, vel.ctypes.data, mass.ctypes.data, ids.ctypes.data, len(pos)))

    def generate(self, kind, n, seed=0, **params):
        """The particles of the synthetic set (kind, n, seed) whose cell lies in this slab's window, drawn on the device
        (pm_generate on a slab context: every rank draws every number and keeps its own; ids = the numbers)."""
        from .simulation import ic_parameters
        arr = np.asarray(ic_parameters(kind, self.L, **params), dtype=np.float64)
        self._check(self.lib.pm_generate(self.ctx, _lib.IC_KINDS[kind], int(n), int(seed) & (2 ** 64 - 1), arr.ctypes.data, len(arr)))

    def generate_plane_counts(self, kind, n, seed=0, **params):
        return generated_plane_counts(self.G, self.L, kind, n, seed, self.device.index or 0, **params)

    # ---- a window of particles apart from the slab (pm_create_slab_windowed) ----
    def set_window(self, cuts):
        """cuts: first plane of every rank's window, and G (pm_slab_set_window)."""
        arr = (ctypes.c_int * len(cuts))(*[int(x) for x in cuts])
        self._check(self.lib.pm_slab_set_window(self.ctx, arr, len(cuts) - 1, self.rank))
        self.w0, self.wn = int(cuts[self.rank]), int(cuts[self.rank + 1] - cuts[self.rank])

    def rekey(self):
        """After set_window: the particles outside the new window go to the outbox (pm_slab_rekey)."""
        self._check(self.lib.pm_slab_rekey(self.ctx))

    def plane_counts(self):
        """This slab's live particles in every mesh plane: int64 tensor of G on the device."""
        out = torch.zeros(self.G, dtype=torch.int64, device=self.device)
        self._check(self.lib.pm_slab_plane_counts(self.ctx, out.data_ptr()))
        return out

    @property
    def deposit_plane_elems(self):
        """float32 elements of one plane as the deposit leaves it"""
        return 2 * self.G * self.pitch if self.spectral else self.G * self.G

    def deposit_planes_out(self, first, count):
        """Planes [first, first + count) of what the deposit wrote (the window's planes, then the one above), flat."""
        out = torch.empty(count * self.deposit_plane_elems, dtype=torch.float32, device=self.device)
        which = _lib.SLAB_ZSPECTRUM if self.spectral else _lib.MESH_DENSITY
        self._check(self.lib.pm_slab_planes(self.ctx, which, int(first), int(count), out.data_ptr(), 0))
        return out

    def mesh_planes_in(self, first, planes, add):
        """Copy or add deposit planes (flat) into the slab's own planes [first, ...) -- the input of the transforms."""
        planes = planes.to(self.device, torch.float32).contiguous()
        count = planes.numel() // self.deposit_plane_elems
        self._check(self.lib.pm_slab_planes(self.ctx, _lib.SLAB_MESH_ZSPECTRUM, int(first), int(count), planes.data_ptr(), 2 if add else 1))

    def mesh_potential_out(self, first, count):
        out = torch.empty(count * self.G * self.G, dtype=torch.float32, device=self.device)
        self._check(self.lib.pm_slab_planes(self.ctx, _lib.SLAB_MESH_POTENTIAL, int(first), int(count), out.data_ptr(), 0))
        return out

    def own_deposit_planes(self, first, mesh_first, count, add):
        """The same for the part of the window that lies in this rank's own slab: one copy inside the context."""
        self._check(self.lib.pm_slab_copy_planes(self.ctx, 1, int(first), int(mesh_first), int(count), 1 if add else 0))

    def own_potential_planes(self, mesh_first, first, count):
        self._check(self.lib.pm_slab_copy_planes(self.ctx, 0, int(first), int(mesh_first), int(count), 0))

    def set_softening(self, length):
        """Gaussian softening of the solve along x (pm_set_softening); every rank must set the same length."""
        self._check(self.lib.pm_set_softening(self.ctx, float(length)))

    def deposit(self):
        self._check(self.lib.pm_slab_deposit(self.ctx))

    def planes_out(self, which, first, count):
        """Copy of planes [first, first+count) of the density (which=0) or potential (which=1) buffer."""
        out = torch.empty((count, self.G, self.G), dtype=torch.float32, device=self.device)
        self._sync()
        self._check(self.lib.pm_slab_planes(self.ctx, which, first, count, out.data_ptr(), 0))
        return out

    def planes_in(self, which, first, planes, add=False):
        planes = planes.to(self.device, torch.float32).contiguous()
        self._sync()
        self._check(self.lib.pm_slab_planes(self.ctx, which, first, planes.shape[0], planes.data_ptr(), 2 if add else 1))

    def ghost_out(self):
        """The plane the deposit wrote beyond the slab (it belongs to the next slab) as a flat float32 tensor."""
        if not self.spectral:
            return self.planes_out(_lib.MESH_DENSITY, self.Gs, 1).reshape(-1)
        out = torch.empty(2 * self.G * self.pitch, dtype=torch.float32, device=self.device)
        self._check(self.lib.pm_slab_planes(self.ctx, _lib.SLAB_ZSPECTRUM, self.Gs, 1, out.data_ptr(), 0))
        return out

    def ghost_in(self, plane):
        """Add the previous slab's extra plane to this slab's first one."""
        plane = plane.to(self.device, torch.float32).contiguous()
        which = _lib.SLAB_ZSPECTRUM if self.spectral else _lib.MESH_DENSITY
        self._check(self.lib.pm_slab_planes(self.ctx, which, 0, 1, plane.data_ptr(), 2))

    def fft2(self, direction):
        self._check(self.lib.pm_slab_fft2(self.ctx, direction))

    def pack(self):
        """The z,y-transformed slab as world blocks (world, Gs, Gs, pitch, 2) float32."""
        self._sync()
        self._check(self.lib.pm_slab_pack(self.ctx, self._blocks[0].data_ptr()))
        return self._blocks[0]

    def unpack(self, blocks):
        assert blocks.is_contiguous() and blocks.device == self.device
        self._sync()
        self._check(self.lib.pm_slab_unpack(self.ctx, blocks.data_ptr()))

    def transposed_buffer(self):
        return self._blocks[1]

    def blocks_buffer(self):
        return self._blocks[0]

    @property
    def planes_in_turn(self):
        """Whether forward_planes / inverse_planes work here (hand-written transforms, a power-of-two number of planes)."""
        return self.lib.pm_slab_forward_planes(self.ctx, self._blocks[0].data_ptr(), 0, 0) == 0

    def forward_planes(self, first, count):
        """The transforms along z and y of owned planes [first, first + count) into their part of blocks_buffer()."""
        self._check(self.lib.pm_slab_forward_planes(self.ctx, self._blocks[0].data_ptr(), int(first), int(count)))

    def inverse_planes(self, blocks, first, count):
        assert blocks.is_contiguous() and blocks.device == self.device
        self._check(self.lib.pm_slab_inverse_planes(self.ctx, blocks.data_ptr(), int(first), int(count)))

    def solve_x(self, spectrum_t):
        assert spectrum_t.is_contiguous() and spectrum_t.device == self.device
        self._sync()
        self._check(self.lib.pm_slab_solve_x(self.ctx, spectrum_t.data_ptr()))

    def gather_kick_drift(self, dt):
        self._check(self.lib.pm_slab_gather_kick_drift(self.ctx, float(dt)))

    def outbox(self):
        """(counts per destination rank, records (n, 8) float32) of the particles that left in the last drift."""
        counts = (ctypes.c_int64 * self.world)()
        self._sync()
        self._check(self.lib.pm_slab_outbox(self.ctx, counts, self._records.data_ptr(), self._records.shape[0]))
        counts = [int(c) for c in counts]
        return counts, self._records[:sum(counts)]

    def outbox_counts(self):
        """int64 tensor on the device, world + 1 entries: leavers per destination rank, then a word that must be 0 (leavers
        that did not fit the outbox; + 2^32 once a particle was found in a slab that does not own its cell).  Nothing
        waits for the drift here."""
        self._check(self.lib.pm_slab_outbox_counts(self.ctx, self._counts.data_ptr()))
        return self._counts

    def outbox_records(self, counts):
        """The leavers' records (n, 8) float32, sorted by (destination, id), given their counts per destination (host)."""
        arr = (ctypes.c_int64 * self.world)(*[int(c) for c in counts])
        self._check(self.lib.pm_slab_outbox_records(self.ctx, arr, self._records.data_ptr(), self._records.shape[0]))
        return self._records[:int(sum(counts))]

    def inbox(self, records):
        records = records.to(self.device, torch.float32).contiguous()
        self._sync()
        self._check(self.lib.pm_slab_inbox(self.ctx, records.data_ptr(), records.shape[0]))

    def particles(self):
        n = self.n
        ids, pos, vel = np.empty(n, np.uint32), np.empty((n, 3), np.float32), np.empty((n, 3), np.float32)
        if n:
            self._check(self.lib.pm_slab_get_particles(self.ctx, ids.ctypes.data, pos.ctypes.data, vel.ctypes.data))
        n = self.n                      # without the particles that left in the last drift
        return ids[:n], pos[:n], vel[:n]

    def synchronize(self):
        self._check(self.lib.pm_synchronize(self.ctx))


class _Comm:
    """all-to-all of flat float32 / int64 tensors; the only collective the step needs."""

    def __init__(self, group=None, always_collective=False):
        # always_collective: go through torch.distributed even when this is the only rank (tests of the backend)
        self.always = always_collective
        self.on = dist.is_available() and dist.is_initialized()
        self.group = group
        self.rank = dist.get_rank(group) if self.on else 0
        self.world = dist.get_world_size(group) if self.on else 1
        self.backend = dist.get_backend(group) if self.on else None
        self.stage_on_host = self.backend == "gloo"        # gloo moves host memory: device tensors are staged

    def all_to_all(self, chunks, recv_sizes):
        """chunks[p]: 1-D tensor for rank p; recv_sizes[p]: elements expected from rank p.  Returns the list received."""
        dtype, device = chunks[0].dtype, chunks[0].device
        if self.world == 1 and not (self.always and self.on):
            return [chunks[0].clone()]
        send = torch.cat([c.reshape(-1) for c in chunks])
        if self.backend == "nccl" and not send.is_cuda:     # RCCL moves device memory only (the particle counts are host integers)
            got = self.all_to_all([c.cuda() for c in chunks], recv_sizes)
            return [t.cpu() for t in got]
        recv = torch.empty(int(sum(recv_sizes)), dtype=dtype, device=device)
        if self.stage_on_host and send.is_cuda:
            s, r = send.cpu(), torch.empty(recv.shape, dtype=dtype)
            dist.all_to_all_single(r, s, [int(x) for x in recv_sizes], [c.numel() for c in chunks], group=self.group)
            recv.copy_(r)
        else:
            dist.all_to_all_single(recv, send, [int(x) for x in recv_sizes], [c.numel() for c in chunks], group=self.group)
        return list(torch.split(recv, [int(x) for x in recv_sizes]))

    def all_to_all_equal(self, send, recv):
        """The same with one contiguous buffer each way and equal shares: rank p's share of `send` lands in share
        `rank-of-sender` of `recv`.  No copies besides the exchange itself."""
        if self.world == 1 and not (self.always and self.on):
            recv.copy_(send)
        elif self.stage_on_host and send.is_cuda:
            s, r = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_to_all_single(r, s, group=self.group)
            recv.copy_(r)
        else:
            dist.all_to_all_single(recv, send, group=self.group)

    def all_to_all_parts(self, send, recv, lo, hi):
        """send, recv: (world, planes, E) tensors.  Rows [lo, hi) of send[p] go to rank p and land in rows [lo, hi) of
        recv[sender] there.  Over RCCL the exchange is asynchronous: the handle returned must be waited for (wait() makes
        the current stream wait) before recv is read or send rewritten; None means it is done."""
        if self.world == 1 and not (self.always and self.on):
            recv[:, lo:hi].copy_(send[:, lo:hi])
            return None
        if self.backend != "nccl":
            # gloo has no list form and moves host memory: one contiguous buffer each way, staged
            s = send[:, lo:hi].contiguous().cpu()
            r = torch.empty_like(s)
            dist.all_to_all_single(r, s, group=self.group)
            recv[:, lo:hi].copy_(r)
            return None
        return dist.all_to_all([recv[p, lo:hi] for p in range(self.world)], [send[p, lo:hi] for p in range(self.world)],
                               group=self.group, async_op=True)

    def shift(self, to_next, to_prev, from_prev_size, from_next_size):
        """Flat tensors to the next and to the previous rank of the ring; returns what the previous and the next rank
        sent here, (from_prev, from_next).  Point-to-point: two sends and two receives posted as one batch (over RCCL one
        grouped call on the two links to the neighbours), not a collective over all ranks.  With two ranks the next and
        the previous rank are the same one: sends and receives pair up in the order they are posted, "to next" first."""
        nxt, prv = (self.rank + 1) % self.world, (self.rank - 1) % self.world
        if self.world == 1:
            if self.always and self.on:         # one rank talking to itself through the backend: the collective path
                got = self.all_to_all([torch.cat([to_next.reshape(-1), to_prev.reshape(-1)])], [from_prev_size + from_next_size])[0]
                return got[:from_prev_size], got[from_prev_size:]
            return to_next.reshape(-1).clone(), to_prev.reshape(-1).clone()
        device, dtype = to_next.device, to_next.dtype
        stage = self.stage_on_host and to_next.is_cuda
        where = "cpu" if stage else device
        send_n, send_p = to_next.reshape(-1).contiguous(), to_prev.reshape(-1).contiguous()
        if stage:
            send_n, send_p = send_n.cpu(), send_p.cpu()
        from_prev = torch.empty(from_prev_size, dtype=dtype, device=where)
        from_next = torch.empty(from_next_size, dtype=dtype, device=where)
        ops = []
        if send_n.numel():
            ops.append(dist.P2POp(dist.isend, send_n, nxt, group=self.group))
        if send_p.numel():
            ops.append(dist.P2POp(dist.isend, send_p, prv, group=self.group))
        if from_prev_size:
            ops.append(dist.P2POp(dist.irecv, from_prev, prv, group=self.group))
        if from_next_size:
            ops.append(dist.P2POp(dist.irecv, from_next, nxt, group=self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if stage:
            from_prev, from_next = from_prev.to(device), from_next.to(device)
        return from_prev, from_next

    def exchange_counts(self, counts):
        """counts: int64 tensor of world equal shares (on the device for RCCL), share p for rank p -> the shares every rank
        sent here, in rank order."""
        if self.world == 1 and not (self.always and self.on):
            return counts.clone()
        if self.stage_on_host and counts.is_cuda:
            s, r = counts.cpu(), torch.empty(counts.shape, dtype=counts.dtype)
            dist.all_to_all_single(r, s, group=self.group)
            return r.to(counts.device)
        if self.backend == "nccl" and not counts.is_cuda:
            return self.exchange_counts(counts.cuda()).cpu()
        recv = torch.empty_like(counts)
        dist.all_to_all_single(recv, counts.contiguous(), group=self.group)
        return recv

    def all_reduce_sum(self, t):
        """Sum of an int64 tensor over the ranks (a new tensor where t lives)."""
        if self.world == 1 and not (self.always and self.on):
            return t.clone()
        if self.stage_on_host and t.is_cuda:
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            return h.to(t.device)
        if self.backend == "nccl" and not t.is_cuda:
            return self.all_reduce_sum(t.cuda()).cpu()
        out = t.clone()
        dist.all_reduce(out, group=self.group)
        return out

    def barrier(self):
        if self.on:
            dist.barrier(group=self.group)


class ThreadComm:
    """The exchanges of _Comm between `world` slabs that live in ONE process on one GPU, each driven by a thread of its own:
    what a rank would send is left where the other threads can see it, and every thread copies its share out on the
    device.  All threads enqueue on the same stream, so a copy is ordered behind the kernels that produced its source as
    long as the producer's calls were made first -- which the barrier before every read sees to.  For rehearsing and
    testing decompositions with more slabs than the box has GPUs (a 2048^3 mesh in eight slabs fits one MI355X); no
    measurement made this way says anything about links.

        comms = ThreadComm.make(world)
        threads = [threading.Thread(target=run, args=(comms[r],)) for r in range(world)]      # run builds SlabSimulation(..., comm=comms[r])
    """

    class _Shared:
        def __init__(self, world):
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    @classmethod
    def make(cls, world):
        shared = cls._Shared(world)
        return [cls(shared, r, world) for r in range(world)]

    def __init__(self, shared, rank, world):
        self._s, self.rank, self.world = shared, rank, world
        self.on, self.always, self.backend, self.group, self.stage_on_host = True, False, "threads", None, False

    def _publish(self, what):
        self._s.slots[self.rank] = what
        self._s.barrier.wait()

    def _retire(self):
        self._s.barrier.wait()          # everybody has read: the slots may be overwritten

    def all_to_all(self, chunks, recv_sizes):
        self._publish(chunks)
        got = [self._s.slots[p][self.rank].reshape(-1).clone() for p in range(self.world)]
        assert [g.numel() for g in got] == [int(x) for x in recv_sizes]
        self._retire()
        return got

    def all_to_all_equal(self, send, recv):
        self._publish(send.view(self.world, -1))
        r = recv.view(self.world, -1)
        for p in range(self.world):
            r[p].copy_(self._s.slots[p][self.rank])
        self._retire()

    def all_to_all_parts(self, send, recv, lo, hi):
        self._publish(send)
        for p in range(self.world):
            recv[p, lo:hi].copy_(self._s.slots[p][self.rank, lo:hi])
        self._retire()
        return None

    def shift(self, to_next, to_prev, from_prev_size, from_next_size):
        nxt, prv = (self.rank + 1) % self.world, (self.rank - 1) % self.world
        self._publish((to_next.reshape(-1), to_prev.reshape(-1)))
        from_prev, from_next = self._s.slots[prv][0].clone(), self._s.slots[nxt][1].clone()
        assert from_prev.numel() == from_prev_size and from_next.numel() == from_next_size
        self._retire()
        return from_prev, from_next

    def exchange_counts(self, counts):
        k = counts.numel() // self.world
        return torch.cat(self.all_to_all([counts[p * k:(p + 1) * k] for p in range(self.world)], [k] * self.world))

    def all_reduce_sum(self, t):
        self._publish(t)
        out = torch.stack([self._s.slots[p] for p in range(self.world)]).sum(dim=0)
        self._retire()
        return out

    def barrier(self):
        self._s.barrier.wait()

    def abort(self):
        """Called by a slab's thread that gives up: every thread waiting at the barrier, now or later, gets BrokenBarrierError."""
        self._s.barrier.abort()


class SlabSimulation:
    """main.py:50-60 over `world` slabs.  Every rank passes its own particles and their numbers in the whole run; `engine`
    defaults to HipSlabEngine on cuda:`device_index`.

    Which particles are a rank's own: those of its slab (slab_of) by default.  With `cuts` (balanced_cuts / equal_cuts:
    the first plane of every rank's window, and G) they are those whose cell lies in the rank's window (owner_by_cuts),
    and the mesh planes a window's deposit writes travel to the slabs that transform them, the potential back: the
    particle work is then divided by particle count, not by volume, whatever the slabs hold.  That needs an engine with
    room for windows (HipSlabEngine(window_planes=...); window_planes here makes one).  rebalance() cuts the windows anew
    from where the particles are now; balance_every = k > 0 does so every k steps when the fullest rank holds more than
    balance_threshold times the mean."""

    def __init__(self, grid_size, box_size, positions, velocities, masses, ids, *, group=None, device_index=0, engine=None,
                 capacity=None, always_collective=False, transpose_chunks=None, comm=None, cuts=None, window_planes=None,
                 balance_every=0, balance_threshold=1.2, generate=None):
        self.comm = comm if comm is not None else _Comm(group, always_collective)
        self.G, self.L = int(grid_size), float(box_size)
        world, rank = self.comm.world, self.comm.rank
        if self.G % world:
            raise ValueError("grid_size must be a multiple of the number of ranks")
        self.Gs = self.G // world
        n = len(positions) if positions is not None else 0
        cap = int(capacity) if capacity else int(1.5 * n) + 4096
        if engine is None and (cuts is not None or window_planes is not None) and world > 1:
            window_planes = int(window_planes) if window_planes is not None else min(self.G - 3, max(self.Gs, self.G // 2))
        self.engine = engine if engine is not None else HipSlabEngine(self.G, self.L, rank * self.Gs, self.Gs, cap, device_index,
                                                                      window_planes=window_planes if world > 1 else None)
        self.windowed = bool(getattr(self.engine, "windowed", False))
        self.cuts = equal_cuts(self.G, world)
        if cuts is not None and list(cuts) != self.cuts:
            if not self.windowed:
                raise ValueError("windows other than the slabs need an engine with room for them (window_planes)")
            self.cuts = [int(x) for x in cuts]
            self.engine.set_window(self.cuts)
        if generate is not None:
            # generate = (kind, n, seed, params): the rank's share of a synthetic set, drawn where it lives
            kind, n_all, seed, params = generate
            self.engine.generate(kind, n_all, seed, **(params or {}))
        else:
            self.engine.set_particles(positions, velocities, masses, ids)
        self._plan()
        self.balance_every, self.balance_threshold = int(balance_every), float(balance_threshold)
        self.steps_done, self.rebalanced = 0, 0
        self.next, self.prev = (rank + 1) % world, (rank - 1) % world
        # groups of planes the two transposes are cut into (see the module text).  By default two where a transpose is long
        # against the launches it costs (a 1024^3 mesh and larger), one -- the plain all-to-all -- below that.
        if transpose_chunks is None:
            transpose_chunks = 2 if (self.G >= 1024 and world > 1) else 1
        self.transpose_chunks = max(1, min(int(transpose_chunks), self.Gs))
        if self.transpose_chunks > 1 and not getattr(self.engine, "planes_in_turn", False):
            self.transpose_chunks = 1
        # profile = True: _step marks the end of each phase (events on torch's current stream for a GPU engine, whose
        # work and collectives all pass through that stream; the host clock otherwise); phase_times() adds them up
        self.profile = False
        self._marks = []
        self._on_gpu = isinstance(self.engine, HipSlabEngine)

    def _plan(self):
        """Who sends which planes to whom, from the cuts (every rank computes the plans of all ranks: the sizes it will
        receive follow from them, nothing about them is exchanged)."""
        world = self.comm.world
        self._dep = [plane_segments(self.cuts, self.G, r, 0, 1) for r in range(world)]      # what a deposit writes
        self._pot = [plane_segments(self.cuts, self.G, r, 1, 2) for r in range(world)]      # what a gather reads

    PHASES = ("deposit", "density_halo", "fft_zy", "transpose", "solve_x", "transpose_back", "ifft_yz", "potential_halo",
              "gather_kick_drift", "migration")

    def _mark(self, name):
        if self.profile:
            if self._on_gpu:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
            else:
                ev = time.perf_counter()
            self._marks.append((name, ev))

    def phase_times(self):
        """{phase: milliseconds} summed over the steps made with profile = True since the last call."""
        if self._on_gpu:
            torch.cuda.synchronize()
        out = dict.fromkeys(self.PHASES, 0.0)
        for (_, a), (name, b) in zip(self._marks, self._marks[1:]):
            if name != "start":
                out[name] += a.elapsed_time(b) if self._on_gpu else (b - a) * 1e3
        self._marks = []
        return out

    def close(self):
        self.engine.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def step(self, dt, nsteps=1):
        try:
            for _ in range(nsteps):
                if self.windowed and self.balance_every > 0 and self.steps_done > 0 and self.steps_done % self.balance_every == 0:
                    self.rebalance(only_above=self.balance_threshold)
                self._step(dt)
                self.steps_done += 1
        except BaseException:
            # slabs that are threads of one process: the others must not wait at a barrier this one will never reach
            abort = getattr(self.comm, "abort", None)
            if abort is not None:
                abort()
            raise

    def _step(self, dt):
        e, G, Gs, world, comm = self.engine, self.G, self.Gs, self.comm.world, self.comm
        plane = G * G
        mark = self._mark
        mark("start")
        # deposit; the plane beyond the slab belongs to the next rank (functions.py:56: i1 = (i + 1) % G)
        e.deposit()
        mark("deposit")
        if self.windowed:
            self._window_planes_to_slabs()
        else:
            ghost = e.ghost_out()
            from_prev, _ = comm.shift(ghost, ghost.new_empty(0), ghost.numel(), 0)
            e.ghost_in(from_prev)
        mark("density_halo")
        # compute_potential (functions.py:73-89)
        exchange = world > 1 or (comm.always and comm.on)
        if self.transpose_chunks > 1 and exchange:
            C = self.transpose_chunks
            bounds = [Gs * c // C for c in range(C + 1)]
            blocks, spec_t = e.blocks_buffer(), e.transposed_buffer()
            b3, t3 = blocks.view(world, Gs, -1), spec_t.view(world, Gs, -1)
            pending = []
            for lo, hi in zip(bounds, bounds[1:]):
                e.forward_planes(lo, hi - lo)
                pending.append(comm.all_to_all_parts(b3, t3, lo, hi))   # on its way while the next group is transformed
            mark("fft_zy")
            for w in pending:
                if w is not None:
                    w.wait()
            mark("transpose")
            e.solve_x(spec_t)
            mark("solve_x")
            pending = [comm.all_to_all_parts(t3, b3, lo, hi) for lo, hi in zip(bounds, bounds[1:])]
            mark("transpose_back")
            for w, lo, hi in zip(pending, bounds, bounds[1:]):
                if w is not None:
                    w.wait()
                e.inverse_planes(blocks, lo, hi - lo)                   # beside the all-to-alls of the groups after it
            mark("ifft_yz")
        else:
            e.fft2(-1)
            blocks = e.pack()
            mark("fft_zy")
            if not exchange:
                spec_t = blocks                         # one slab: the blocks are the transposed spectrum already
            else:
                spec_t = e.transposed_buffer()
                comm.all_to_all_equal(blocks.reshape(-1), spec_t.reshape(-1))
            mark("transpose")
            e.solve_x(spec_t)
            mark("solve_x")
            if spec_t is not blocks:
                comm.all_to_all_equal(spec_t.reshape(-1), blocks.reshape(-1))
            mark("transpose_back")
            e.unpack(blocks)
            e.fft2(+1)
            mark("ifft_yz")
        if self.windowed:
            self._slab_potential_to_windows()
        else:
            # potential planes beside the slab: one below (from the previous rank), two above (from the next rank)
            to_next = e.planes_out(1, Gs, 1)            # my last owned plane
            to_prev = e.planes_out(1, 1, 2)             # my first two owned planes
            below, above = comm.shift(to_next, to_prev, plane, 2 * plane)
            e.planes_in(1, 0, below.reshape(1, G, G))
            e.planes_in(1, Gs + 1, above.reshape(2, G, G))
        mark("potential_halo")
        # gradient, gather, kick, drift (main.py:54-60); then the particles that changed slab: how many go where is
        # exchanged between the ranks before anyone reads it, so the host waits for the device once per step
        e.gather_kick_drift(dt)
        mark("gather_kick_drift")
        self._migrate()
        mark("migration")

    def _migrate(self):
        """The particles in the engine's outbox (leavers of a drift, or of a change of windows) to the ranks that own them now."""
        e, world, comm = self.engine, self.comm.world, self.comm
        mine = e.outbox_counts()                    # world + 1 entries, where the engine's data lives
        # every rank learns of every rank's trouble word (it travels beside the count), so that all of them raise in the
        # same step instead of one raising and the others waiting for it in the next exchange
        pairs = torch.stack([mine[:world], mine[world:world + 1].expand(world)], dim=1).reshape(-1)
        theirs = comm.exchange_counts(pairs)
        both = torch.cat([mine, theirs]).cpu().tolist()
        counts, incoming = both[:world], both[world + 1::2]
        trouble = max([both[world]] + both[world + 2::2])
        if trouble:
            raise _lib.PMError("particle migration: " + ("a particle was delivered to a slab that does not own its cell"
                                                          if trouble >> 32 else f"{trouble} leavers did not fit the outbox"))
        records = e.outbox_records(counts)
        flat = records.reshape(-1)
        offs = np.concatenate(([0], np.cumsum(counts))) * 8
        got = comm.all_to_all([flat[offs[p]:offs[p + 1]] for p in range(world)], [8 * k for k in incoming])
        e.inbox(torch.cat(got).reshape(-1, 8))

    def _window_planes_to_slabs(self):
        """What the deposit wrote -- the planes of this rank's window and the one above -- to the ranks whose slabs hold those
        planes.  A slab's plane gets the plane of the window that contains it (copied), then the plane above the window
        before that one (added, functions.py:56): one fixed order."""
        e, world, comm, rank = self.engine, self.comm.world, self.comm, self.comm.rank
        elems = e.deposit_plane_elems
        out = [[] for _ in range(world)]
        for t0, count, owner, first in self._dep[rank]:
            if owner != rank:
                out[owner].append(e.deposit_planes_out(t0, count))
        dev = getattr(e, "device", torch.device("cpu"))
        chunks = [torch.cat(x) if x else torch.empty(0, dtype=torch.float32, device=dev) for x in out]
        sizes = [sum(c for _, c, o, _ in self._dep[src] if o == rank) * elems if src != rank else 0 for src in range(world)]
        got = comm.all_to_all(chunks, sizes) if world > 1 else [chunks[0]]
        for ghosts in (False, True):
            for src in range(world):
                wn, at = self.cuts[src + 1] - self.cuts[src], 0
                for t0, count, owner, first in self._dep[src]:
                    if owner != rank:
                        continue
                    is_ghost = t0 >= wn
                    if is_ghost == ghosts:
                        if src == rank:
                            e.own_deposit_planes(t0, first, count, add=ghosts)
                        else:
                            e.mesh_planes_in(first, got[src][at:at + count * elems], add=ghosts)
                    if src != rank:
                        at += count * elems

    def _slab_potential_to_windows(self):
        """The potential of every window's planes, the one below and the two above, from the ranks whose slabs hold them."""
        e, world, comm, rank, G = self.engine, self.comm.world, self.comm, self.comm.rank, self.G
        plane = G * G
        out = [[] for _ in range(world)]
        for dest in range(world):
            if dest != rank:
                out[dest] = [e.mesh_potential_out(first, count) for _, count, owner, first in self._pot[dest] if owner == rank]
        dev = getattr(e, "device", torch.device("cpu"))
        chunks = [torch.cat(x) if x else torch.empty(0, dtype=torch.float32, device=dev) for x in out]
        sizes = [sum(c for _, c, o, _ in self._pot[rank] if o == src) * plane if src != rank else 0 for src in range(world)]
        got = comm.all_to_all(chunks, sizes) if world > 1 else [chunks[0]]
        at = [0] * world
        for t0, count, owner, first in self._pot[rank]:
            if owner == rank:
                e.own_potential_planes(first, t0, count)
            else:
                e.planes_in(1, t0, got[owner][at[owner]:at[owner] + count * plane].reshape(count, G, G))
                at[owner] += count * plane

    def loads(self):
        """Particles every rank holds now (a list of world numbers, the same on every rank)."""
        world = self.comm.world
        n = torch.full((world,), int(self.engine.n), dtype=torch.int64, device=self.engine.device if self._on_gpu else "cpu")
        return [int(x) for x in self.comm.exchange_counts(n).cpu().tolist()]

    def plane_weight(self, n_total):
        """balanced_cuts' plane_weight for this mesh and a set of n_total particles: half the mean number of particles per plane"""
        return 0.5 * float(n_total) / self.G

    def rebalance(self, only_above=None):
        """Cuts the windows anew so that every rank holds (nearly) the same number of particles, from the particles per
        mesh plane summed over the ranks; particles change hands accordingly.  Between steps only.  only_above = x: nothing
        is done unless the fullest rank holds more than x times the mean.  Returns whether the windows changed."""
        if not self.windowed:
            raise ValueError("this simulation's particles are those of its slabs: make it with cuts or window_planes")
        e, world = self.engine, self.comm.world
        if only_above is not None:
            # (the measure is the one the windows are cut by: particles, and plane_weight per plane of the window)
            loads = self.loads()
            w = self.plane_weight(sum(loads))
            cost = [n + w * (b - a) for n, a, b in zip(loads, self.cuts, self.cuts[1:])]
            if max(cost) <= only_above * (sum(cost) / world):
                return False
        counts = self.comm.all_reduce_sum(e.plane_counts()).cpu().numpy()
        cuts = balanced_cuts(counts, world, e.window_planes, self.plane_weight(counts.sum()))
        if cuts == self.cuts:
            return False
        e.set_window(cuts)
        self.cuts = cuts
        self._plan()
        e.rekey()
        self._migrate()
        self.rebalanced += 1
        return True

    def particles(self):
        """(ids, positions, velocities) of the particles this rank owns now."""
        return self.engine.particles()

    def synchronize(self):
        self.engine.synchronize()
        self.comm.barrier()
