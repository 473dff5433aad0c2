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


def run_rank(rank, world, port, out, chunks=1, comm=None):
    from oracle import pm_oracle as O
    O.build()
    from pm.distributed import SlabSimulation, slab_of
    from slab_oracle_engine import OracleSlabEngine
    if world > 1 and comm is None:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    pos, vel, mass = particles()
    mine = slab_of(pos, G, L, world) == rank
    ids = np.arange(N, dtype=np.uint32)
    eng = OracleSlabEngine(G, L, rank * (G // world), G // world)
    sim = SlabSimulation(G, L, pos[mine], vel[mine], mass[mine], ids[mine], engine=eng, transpose_chunks=chunks, comm=comm)
    assert sim.transpose_chunks == min(chunks, G // world)
    sim.profile = True                       # marks after every phase must not change anything
    moved = 0
    for _ in range(K):
        before = set(sim.particles()[0].tolist())
        sim.step(DT)
        moved += len(before - set(sim.particles()[0].tolist()))
    phases = sim.phase_times()
    assert set(phases) == set(sim.PHASES) and all(v >= 0.0 for v in phases.values()) and phases["deposit"] > 0.0
    assert sim.phase_times() == dict.fromkeys(sim.PHASES, 0.0)
    i, x, v = sim.particles()
    np.savez(out, ids=i, pos=x, vel=v, moved=moved)
    if world > 1 and comm is None:
        dist.barrier()
        dist.destroy_process_group()


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


@pytest.mark.parametrize("world,chunks", [(2, 1), (4, 2)])
def test_slab_world_n_threads(tmp_path, world, chunks):
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
            run_rank(r, world, 0, outs[r], chunks, comms[r])
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
