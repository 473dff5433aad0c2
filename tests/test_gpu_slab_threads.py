"""Several slabs on the one GPU of the test box as threads of one process (pm.distributed.ThreadComm: the exchanges are
device copies).  Against the one-context run at small sizes; and the mesh of the 8-GPU configuration -- 2048^3 in eight
slabs of 256 planes, which fits one MI355X with 2^26 particles -- against the closed form of tests/lattice_case.py."""
import threading

import numpy as np
import pytest

import lattice_case as LC
from test_gpu_slab import CASES, make, single_context

pytestmark = pytest.mark.gpu

_create = threading.Lock()


def free_device_memory():
    """Bytes of device memory that are free once what earlier tests left behind (contexts awaiting collection, blocks that
    torch's allocator keeps) has been given back."""
    import gc
    import torch
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return torch.cuda.mem_get_info()[0]


def run_threads(world, build, steps, dt, chunks=None, keep="all", info=None, **sim_args):
    """build(rank) -> (G, L, pos, vel, mass, ids, capacity); returns per rank [(ids, pos, vel) after each step] (keep =
    "last": after the last step only).  sim_args go to SlabSimulation; info(rank, sim) -> anything is called at the end and
    its results are returned as a second list."""
    from pm.distributed import SlabSimulation, ThreadComm
    comms = ThreadComm.make(world)
    results, errors, infos = [None] * world, [], [None] * world

    def run(r):
        try:
            G, L, pos, vel, mass, ids, cap = build(r)
            with _create:                                   # contexts are made one at a time
                sim = SlabSimulation(G, L, pos, vel, mass, ids, device_index=0, capacity=cap, comm=comms[r], transpose_chunks=chunks, **sim_args)
            del pos, vel, mass, ids
            out = []
            for k in range(steps):
                sim.step(dt)
                if keep == "all" or k == steps - 1:
                    out.append(sim.particles())
            if info is not None:
                infos[r] = info(r, sim)
            sim.synchronize()
            sim.close()
            results[r] = out
        except BaseException as e:                          # a thread that dies must not leave the others at a barrier
            errors.append(e)
            comms[r]._s.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not errors, errors
    return results if info is None else (results, infos)


@pytest.mark.parametrize("case,world,chunks", [("g128", 4, None), ("g512", 8, 2)])
def test_thread_slabs_against_one_context(case, world, chunks):
    from pm.distributed import slab_of
    c = CASES[case]
    pos, vel, mass = make(c["n"], c["L"], 3, c["crowd"])
    owner = slab_of(pos, c["G"], c["L"], world)
    ids = np.arange(c["n"], dtype=np.uint32)

    def build(r):
        mine = owner == r
        return c["G"], c["L"], pos[mine], vel[mine], mass[mine], ids[mine], c["n"]

    res = run_threads(world, build, c["K"], c["dt"], chunks)
    xr, vr = single_context(case)
    seen = np.zeros(c["n"], bool)
    for out in res:
        i, x, v = out[-1]
        assert not seen[i].any()
        seen[i] = True
        dx = np.abs(x - xr[i])
        assert np.minimum(dx, c["L"] - dx).max() <= 1e-5 * c["L"]
        assert np.abs(v - vr[i]).max() <= 1e-5 * np.abs(vr).max()
    assert seen.all()


def lattice_run(G, world, kstride, mode, chunks):
    from pm.distributed import slab_of
    L, eps, dt = 10000, 0.5, 0.005
    h = L / G
    v0 = 0.8 * h / dt                   # every particle moves on by one cell: the top plane of each slab changes hands
    scale = LC.scale_for(G, kstride, mode, eps, v0, dt)
    Gs = G // world

    def build(r):
        pos, vel, mass, ids = LC.slab_particles(G, L, kstride, mode, eps, scale, v0, r * Gs, Gs)
        return G, L, pos, vel, mass, ids, len(ids) + len(ids) // 8 + 4096

    res = run_threads(world, build, 2, dt, chunks)
    n_total = G * G * (G // kstride)
    Fx, Fz = LC.forces_2d(G, kstride, mode, eps, scale)
    kick = max(np.abs(Fx).max(), np.abs(Fz).max()) * dt
    assert 0.02 * v0 < kick < 0.1 * v0
    worst = 0.0
    for step in (0, 1):
        count = 0
        for r, out in enumerate(res):
            ids, x, v = out[step]
            count += len(ids)
            assert np.all(slab_of(x, G, L, world) == r)             # everybody is where its cell is
            i0, j0, k0 = LC.cells_of(ids, G, kstride)
            if step == 0:
                # main.py:58 against the closed form; main.py:59-60 with the velocities as they came out, bit for bit
                want = np.zeros((len(ids), 3))
                want[:, 0] = np.float64(np.float32(v0)) + Fx[i0, k0] * dt
                want[:, 2] = Fz[i0, k0] * dt
                worst = max(worst, np.abs(v - want).max())
                x0 = np.stack([(i0 + LC.FRAC) * h, (j0 + LC.FRAC) * h, (k0 + LC.FRAC) * h], 1).astype(np.float32)
                assert np.array_equal(x, (x0 + v * np.float32(dt)) % np.float32(L))
                moved_up = (i0 + 1) % G // (G // world) != i0 // (G // world)
                assert np.all((i0[moved_up] + 1) % G // (G // world) == r) and moved_up.sum() == G * (G // kstride)
            else:
                # a second step from the shifted lattice: nobody lost, and 1.85 cells (give or take the kicks) from the start
                cell = np.floor(x[:, 0].astype(np.float64) / h).astype(np.int64) % G
                assert np.all((cell == (i0 + 1) % G) | (cell == (i0 + 2) % G))
        assert count == n_total
    assert worst <= 1e-4 * kick, (worst, kick)
    return worst / kick


def test_lattice_closed_form_small():
    lattice_run(256, 4, 32, 4, None)


def test_mesh_of_the_eight_gpu_configuration():
    """2048^3 in eight slabs of 256 planes (the hand-written 32 x 8 x 8 column and 1024-point row transforms, the fused
    x pass on the transposed spectrum, both transposes in two groups of planes, halos, migration of the 2^15 particles
    at the top of every slab) with 2^26 particles on a lattice, against the closed form."""
    import torch
    free = free_device_memory()
    if free < 235e9:
        pytest.skip(f"needs about 200 GB of device memory for the eight slabs, {free / 1e9:.0f} GB are free")
    lattice_run(2048, 8, 128, 16, 2)


def collect(results, n):
    """(ids seen once each, positions, velocities) in the order of the particles' numbers, from every rank's last record."""
    x, v, seen = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32), np.zeros(n, bool)
    for out in results:
        i, xi, vi = out[-1]
        assert not seen[i].any()
        seen[i] = True
        x[i], v[i] = xi, vi
    assert seen.all()
    return x, v


@pytest.mark.parametrize("mode", ["cut at the start", "cut anew during the run"])
def test_windows_of_equal_particle_count_on_a_clustered_set(mode):
    """The two-galaxy collision on 512^3 in eight slabs, the particles held by windows of equal particle count instead of
    by slab (the fullest of the eight slabs holds half of this set): the planes a window's deposit writes travel to the
    slabs that transform them as z-transformed planes, the potential comes back, leavers go to the rank whose window
    holds their new cell.  Against the one-context run: 1e-5 L and 1e-5 max|v| after three steps."""
    import bench
    import pm
    from pm.distributed import balanced_cuts, equal_cuts, owner_by_cuts, plane_of, slab_of
    n, G, L, dt, world, K = 1 << 21, 512, 10000, 0.005, 8, 3
    pos, vel, mass = bench.make_workload("two_galaxy", n, L, 1052)
    ids = np.arange(n, dtype=np.uint32)
    by_slab = np.bincount(slab_of(pos, G, L, world), minlength=world)
    assert by_slab.max() > 0.4 * n                          # what the windows are for
    if mode == "cut at the start":
        cuts = balanced_cuts(np.bincount(plane_of(pos, G, L), minlength=G), world, G // 2)
        owner = owner_by_cuts(pos, G, L, cuts)
        assert np.bincount(owner, minlength=world).max() < 1.25 * n / world
        args = dict(cuts=cuts, window_planes=G // 2)
    else:
        cuts, owner = equal_cuts(G, world), slab_of(pos, G, L, world)
        args = dict(window_planes=G // 2, balance_every=1, balance_threshold=1.2)

    def build(r):
        mine = owner == r
        return G, L, pos[mine], vel[mine], mass[mine], ids[mine], int(max(mine.sum(), n // world) * 1.5) + 4096

    def info(r, sim):
        i, x, _ = sim.particles()
        assert np.all(owner_by_cuts(x, G, L, sim.cuts) == r)        # everybody is where its cell's window is
        return sim.loads(), sim.cuts, sim.rebalanced

    res, infos = run_threads(world, build, K, dt, keep="last", info=info, **args)
    loads, final_cuts, recuts = infos[0]
    assert sum(loads) == n
    if mode == "cut at the start":
        assert max(loads) < 1.25 * n / world, loads
    else:
        # cut anew by particles + half a mean plane's particles per plane of the window (SlabSimulation.plane_weight)
        assert recuts >= 1 and final_cuts != equal_cuts(G, world)
        cost = [k + 0.5 * n / G * (b - a) for k, a, b in zip(loads, final_cuts, final_cuts[1:])]
        assert max(cost) < 1.25 * sum(cost) / world, (loads, final_cuts)
    x, v = collect(res, n)
    with pm.Simulation(G, L, pos, vel, mass, fused_gradient=True) as sim:
        sim.step(dt, K)
        xr, vr = sim.positions(), sim.velocities()
    dx = np.abs(x.astype(np.float64) - xr)
    assert np.minimum(dx, L - dx).max() <= 1e-5 * L
    assert np.abs(v - vr).max() <= 1e-5 * np.abs(vr).max()


def test_synthetic_sets_drawn_slab_by_slab():
    """pm_generate on slab contexts: every slab draws the whole set and keeps the particles of its window, in the order
    of their numbers -- together they are the set a one-context pm_generate draws, bit for bit; and the plane counts every
    rank computes for itself are the set's."""
    import pm
    from pm.distributed import HipSlabEngine, balanced_cuts, plane_of
    n, G, L, world = 300_000, 128, 10000, 4
    for kind in ("two_galaxy", "uniform"):
        with pm.Simulation(G, L, capacity=n) as sim:
            sim.generate(kind, n, seed=7)
            xr, vr = sim.positions(), sim.velocities()
        counts = np.bincount(plane_of(xr, G, L), minlength=G)
        seen = np.zeros(n, bool)
        cuts = None
        for r in range(world):
            eng = HipSlabEngine(G, L, r * (G // world), G // world, n, 0, window_planes=G - 3)
            try:
                got = eng.generate_plane_counts(kind, n, seed=7)
                assert np.array_equal(got, counts)
                cuts = balanced_cuts(got, world, G - 3)
                eng.set_window(cuts)
                eng.generate(kind, n, seed=7)
                i, x, v = eng.particles()
            finally:
                eng.close()
            assert np.all(np.diff(i.astype(np.int64)) > 0) and not seen[i].any()
            seen[i] = True
            assert np.array_equal(x, xr[i]) and np.array_equal(v, vr[i])
            w = plane_of(x, G, L)
            assert np.all((w >= cuts[r]) & (w < cuts[r + 1]))
        assert seen.all()


def test_disk_of_the_clustered_configuration_in_eight_windows():
    """The clustered-disk configuration (N = 67,108,864 on 1024^3: crowded rows, deposit units) over eight ranks as threads
    on the one GPU: every rank draws its share on the device and holds a window of equal load (the two middle slabs of
    eight would hold nearly all of this set); one step with migration, and a second one, against the one-context run."""
    import torch
    import pm
    from pm.distributed import balanced_cuts, generated_plane_counts, owner_by_cuts
    free = free_device_memory()
    if free < 150e9:
        pytest.skip(f"needs about 120 GB of device memory, {free / 1e9:.0f} GB are free")
    n, G, L, dt, world, K = 67_108_864, 1024, 10000, 0.005, 8, 2
    counts = generated_plane_counts(G, L, "disk", n, 42)
    assert counts.sum() == n and counts.reshape(world, -1).sum(axis=1).max() > 0.45 * n
    cuts = balanced_cuts(counts, world, G // 2, 0.5 * n / G)
    loads = [int(counts[a:b].sum()) for a, b in zip(cuts, cuts[1:])]

    def build(r):
        return G, L, None, None, None, None, int(loads[r] * 1.25) + (1 << 20)

    def info(r, sim):
        i, x, _ = sim.particles()
        assert np.all(owner_by_cuts(x, G, L, sim.cuts) == r)
        return sim.loads()

    res, infos = run_threads(world, build, K, dt, keep="last", info=info, cuts=cuts, window_planes=G // 2, generate=("disk", n, 42, None))
    assert sum(infos[0]) == n
    x, v = collect(res, n)
    del res
    with pm.Simulation(G, L, capacity=n, fused_gradient=True) as sim:
        sim.generate("disk", n, seed=42)
        sim.step(dt, K)
        xr, vr = sim.positions(), sim.velocities()
    dx = np.abs(x.astype(np.float64) - xr)
    assert np.minimum(dx, L - dx).max() <= 1e-5 * L
    # this set starts nearly at rest (sigma 10) and its velocities after two steps are the kicks of the dense disk: the two
    # ways of summing a 1024^3 solve in float32 (planes added as z-spectra and transformed slab by slab; one context) agree to
    # 1e-6 of max|phi|, which the difference quotient of the gradient carries into the kicks at 1e-5 of the largest (measured
    # 1.02e-5; test_fused_step_1024 allows the rocFFT pipeline 5e-5 for the same reason)
    assert np.abs(v - vr).max() <= 2e-5 * np.abs(vr).max()


def gather_on_planes(phi, i_first, pos, vel, G, L, dt):
    """force_CIC (functions.py:99-180) and main.py:58-60 for particles whose cells lie in mesh planes that `phi` covers:
    phi[q] is plane i_first + q of the potential.  np.gradient's central difference along the first axis needs no plane
    outside phi for the cells' two planes as long as those are not phi's first or last.  float64 weights, float32 sums
    rounded at every add, the reference's corner order."""
    from oracle import pm_oracle as O
    c, d = O.cell_and_fraction_f64(pos, G, L)
    grad = np.gradient(phi)
    q = c[:, 0] - i_first
    assert q.min() >= 1 and q.max() + 1 <= len(phi) - 2
    c1 = (c + 1) % G
    idx = np.stack((c, c1))
    qi = np.stack((q, q + 1))
    w = np.stack((1.0 - d, d))
    f = np.zeros((len(c), 3), np.float32)
    for di, dj, dk in ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1)):
        wt = w[di, :, 0] * w[dj, :, 1] * w[dk, :, 2]
        node = (qi[di], idx[dj, :, 1], idx[dk, :, 2])
        for a in range(3):
            f[:, a] = (f[:, a].astype(np.float64) + grad[a][node].astype(np.float64) * wt).astype(np.float32)
    v1 = vel + f * np.float32(dt)
    x1 = np.remainder(pos + v1 * np.float32(dt), np.float32(L))
    return f, x1.astype(np.float32), v1.astype(np.float32)


def test_gather_on_planes_is_the_oracle_gather():
    """(the restatement above against the oracle's own force_CIC and kick_drift, where the whole mesh fits)"""
    from oracle import pm_oracle as O
    G, L, dt = 32, 10000, 0.005
    rng = np.random.default_rng(2)
    phi = rng.normal(0, 1e4, (G, G, G)).astype(np.float32)
    pos = (rng.random((5000, 3)) * L).astype(np.float32)
    pos[:, 0] = (rng.random(5000) * 10 + 7).astype(np.float32) * np.float32(L / G)       # cells in planes 7 .. 16
    vel = rng.normal(0, 100, (5000, 3)).astype(np.float32)
    f, x1, v1 = gather_on_planes(phi[5:20], 5, pos, vel, G, L, dt)
    fo = O.force_CIC(O.gradient(phi), pos, G, L)
    xo, vo = O.kick_drift(pos, vel, fo, dt, L)
    assert np.array_equal(f, fo) and np.array_equal(x1, xo) and np.array_equal(v1, vo)


def test_uniform_set_of_the_eight_gpu_configuration_on_its_mesh():
    """The 8-GPU configuration's mesh (2048^3 in eight slabs of 256 planes) with the uniform random set, 2^26 of its 2^29
    particles (what one MI355X holds beside the eight slabs' meshes), drawn on the device slab by slab.  No oracle runs a
    2048^3 mesh, so the step is checked through what does not need one: every particle is owned once, by the slab that
    holds its cell, before and after the step; the slabs' cell orders are the oracle's keys in order; the deposit
    conserves the mass; and on a sample of one slab the kick and drift are the oracle's gather (force_CIC with
    np.gradient, functions.py:99-180, main.py:54-60) of the potential planes that slab holds, bit for bit."""
    import torch
    from oracle import pm_oracle as O
    from pm.distributed import slab_of
    free = free_device_memory()
    if free < 235e9:
        pytest.skip(f"needs about 200 GB of device memory for the eight slabs, {free / 1e9:.0f} GB are free")
    n, G, L, dt, world = 1 << 26, 2048, 10000, 0.005, 8
    Gs = G // world
    start, seen_mass = [None] * world, [0.0] * world

    def build(r):
        return G, L, None, None, None, None, n // world + n // 64 + 4096

    # (run_threads is asked for zero steps: `info0` reads the particles as they were drawn, steps once itself and reads them again)
    def info0(r, sim):
        e = sim.engine
        e.deposit()                                             # bins the particles: storage order = cell order
        i, x, v = sim.particles()
        assert np.all(slab_of(x, G, L, world) == r)
        keys = O.cell_keys(x, G, L) - r * Gs * G * G
        assert keys.min() >= 0 and keys.max() < Gs * G * G and np.all(np.diff(keys) >= 0)
        same = np.diff(keys) == 0
        assert np.all(np.diff(i.astype(np.int64))[same] > 0)    # ties keep the order of the particles' numbers
        # the deposit's planes, transformed along z: their k = 0 terms are the planes' masses
        dc = sum(float(e.deposit_planes_out(p0, min(32, Gs + 1 - p0)).view(-1, G, e.pitch, 2)[:, :, 0, 0].double().sum())
                 for p0 in range(0, Gs + 1, 32))
        seen_mass[r] = dc
        start[r] = (i, x, v)
        sim.step(dt)
        i1, x1, v1 = sim.particles()
        assert np.all(slab_of(x1, G, L, world) == r)
        out = None
        if r == 3:
            # planes 0 .. 11 of this slab's potential buffer are mesh planes 3 Gs - 1 .. 3 Gs + 10
            phi = e.planes_out(1, 0, 12).cpu().numpy()
            cell_i = O.cell_index(x, G, L)[:, 0]
            pick = np.flatnonzero((cell_i >= 3 * Gs + 1) & (cell_i <= 3 * Gs + 7))[::17][:40_000]
            out = (phi, i[pick], x[pick], v[pick])
        return i1, x1, v1, out

    res, infos = run_threads(world, build, 0, dt, info=info0, generate=("uniform", n, 1052, None))
    assert abs(sum(seen_mass) - n) <= 1e-6 * n
    before = np.zeros(n, np.int32)
    after = np.zeros(n, np.int32)
    for r in range(world):
        before[start[r][0]] += 1
        after[infos[r][0]] += 1
    assert np.all(before == 1) and np.all(after == 1)
    moved = sum(int((slab_of(start[r][1], G, L, world) != r).sum()) for r in range(world))
    assert moved == 0
    # the sample of slab 3: where its particles are now (they may have changed slab), against the oracle's gather
    phi, ids, x0, v0 = infos[3][3]
    assert len(ids) > 10_000
    f, xe, ve = gather_on_planes(phi, 3 * Gs - 1, x0, v0, G, L, dt)
    x_all, v_all = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32)
    for r in range(world):
        x_all[infos[r][0]], v_all[infos[r][0]] = infos[r][1], infos[r][2]
    assert np.abs(f).max() > 0
    assert np.array_equal(v_all[ids], ve) and np.array_equal(x_all[ids], xe)
