#!/usr/bin/env python3
"""Local phases of ONE slab of a P-way decomposition on one GPU, exchanges left out: what one GPU of the 8-GPU
configuration (uniform, 2^29 particles, 2048^3 mesh: 256 planes and 2^26 particles per GPU) spends between its
collectives.  The all-to-alls are replaced by nothing (the slab solves along x on its own blocks: wrong physics, the
right kernels, sizes and traffic), the halo planes by the slab's own planes.

    python tools/slab_local_phases.py --grid 2048 --slabs 8 --n 67108864 --steps 3
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nbody-simulation_amd")):
    sys.path.insert(0, p)

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=2048)
ap.add_argument("--slabs", type=int, default=8)
ap.add_argument("--n", type=int, default=1 << 26, help="particles in this slab")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--rocfft-only", action="store_true")
ap.add_argument("--workload", default="uniform", help="uniform: --n particles inside slab 0; or a workload of bench.py (two_galaxy, disk): --n is then the WHOLE "
                                                     "set, and --rank picks the slab whose share is timed (-1: every slab in turn)")
ap.add_argument("--rank", type=int, default=-1)
ap.add_argument("--quiet-set", action="store_true", help="uniform workload: light particles (1e-6) with the velocities of bench.py's uniform set "
                                                        "(sigma 10), so that few particles change their cell per step")
ap.add_argument("--full-sort", action="store_true", help="PM_FLAG_FULL_SORT: every binning sorts the whole slab")
ap.add_argument("--windows", action="store_true", help="a whole-set workload: the ranks hold windows of equal particle count (balanced_cuts) instead of "
                                                      "their slabs; the copies that stage the planes for the exchanges are timed as phases of their own")
args = ap.parse_args()

import torch  # noqa: E402
from pm import _lib  # noqa: E402
from pm.distributed import HipSlabEngine  # noqa: E402

G, P, L, dt = args.grid, args.slabs, 10000.0, 0.005
Gs = G // P
PH = ("deposit", "fft_zy", "solve_x", "ifft_yz", "gather_kick_drift", "migration")


def one_slab(rank, pos, vel, mass, ids):
    n = len(pos)
    e = HipSlabEngine(G, L, rank * Gs, Gs, int(n * 1.1) + 4096, 0, rocfft_only=args.rocfft_only, full_sort=args.full_sort)
    e.set_particles(pos, vel, mass, ids)
    acc = dict.fromkeys(PH, 0.0)

    def step(record):
        marks = []

        def mark(name):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append((name, ev))
        mark("start")
        e.deposit()
        ghost = e.ghost_out()
        e.ghost_in(ghost * 0)
        mark("deposit")
        e.fft2(-1)
        blocks = e.pack()
        mark("fft_zy")
        e.solve_x(blocks)           # (P, Gs, Gs, pitch) read as (G, Gs, pitch): this slab's rows of every plane, were they exchanged
        mark("solve_x")
        e.unpack(blocks)
        e.fft2(+1)
        mark("ifft_yz")
        e.planes_in(1, 0, e.planes_out(1, Gs, 1))
        e.planes_in(1, Gs + 1, e.planes_out(1, 1, 2))
        e.gather_kick_drift(dt)
        mark("gather_kick_drift")
        counts = e.outbox_counts().cpu().tolist()
        e.outbox_records(counts[:P])
        e.inbox(torch.empty((0, 8), dtype=torch.float32, device="cuda"))
        mark("migration")
        torch.cuda.synchronize()
        if record:
            for (_, a), (name, b) in zip(marks, marks[1:]):
                acc[name] += a.elapsed_time(b)

    step(False)
    for _ in range(args.steps):
        step(True)
    out = {k: round(v / args.steps, 3) for k, v in acc.items()}
    total = sum(out.values())
    m_local = G * G * Gs
    line = {"grid": G, "slabs": P, "slab": rank, "planes": Gs, "particles_in_slab_at_start": n, "hand_written_transforms": bool(e.spectral),
            "phases_ms": out, "local_ms_per_step": round(total, 3), "last_binning_from_order_before_and_movers": e.last_binning(),
            "local_algorithmic_GB": round((64 * n + 80 * m_local) / 1e9, 2),
            "local_frac_of_8TBps": round((64 * n + 80 * m_local) / (total * 1e-3) / 8e12, 3)}
    print(json.dumps(line), flush=True)
    e.close()
    return total


def one_window(rank, cuts, pos, vel, mass, ids):
    """The same for a rank that holds the particles of window [cuts[rank], cuts[rank + 1]) and transforms slab `rank`: what
    it does between the exchanges, with the staging copies of the planes that travel (out of the window's deposit, into the
    slab's planes; out of the slab's potential, into the window's potential buffer) -- the planes themselves stand in for
    the ones other ranks would send."""
    from pm.distributed import plane_segments
    n = len(pos)
    e = HipSlabEngine(G, L, rank * Gs, Gs, int(n * 1.1) + 4096, 0, full_sort=args.full_sort, window_planes=min(G - 3, max(Gs, G // 2)))
    e.set_window(cuts)
    e.set_particles(pos, vel, mass, ids)
    wn = cuts[rank + 1] - cuts[rank]
    dep, pot = plane_segments(cuts, G, rank, 0, 1), plane_segments(cuts, G, rank, 1, 2)
    phases = ("deposit", "planes_out", "planes_in", "fft_zy", "solve_x", "ifft_yz", "potential_out", "potential_in", "gather_kick_drift", "migration")
    acc = dict.fromkeys(phases, 0.0)
    stand_in = torch.zeros(Gs * e.deposit_plane_elems, dtype=torch.float32, device="cuda")
    sent_planes = sum(c for _, c, o, _ in dep if o != rank)
    got_potential = sum(c for _, c, o, _ in pot if o != rank)

    def step(record):
        marks = []

        def mark(name):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append((name, ev))
        mark("start")
        e.deposit()
        mark("deposit")
        out = [e.deposit_planes_out(t0, c) for t0, c, o, _ in dep if o != rank]
        if out:
            torch.cat(out)
        mark("planes_out")
        own = [(t0, c, f) for t0, c, o, f in dep if o == rank and t0 < wn]
        covered = sum(c for _, c, _ in own)
        for t0, c, f in own:
            e.own_deposit_planes(t0, f, c, add=False)
        if covered < Gs:                                    # the rest of the slab's planes arrive from other ranks
            e.mesh_planes_in(0, stand_in[:(Gs - covered) * e.deposit_plane_elems], add=False)
        e.mesh_planes_in(0, stand_in[:e.deposit_plane_elems], add=True)         # and a plane above somebody's window
        mark("planes_in")
        e.fft2(-1)
        blocks = e.pack()
        mark("fft_zy")
        e.solve_x(blocks)
        mark("solve_x")
        e.unpack(blocks)
        e.fft2(+1)
        mark("ifft_yz")
        e.mesh_potential_out(0, Gs)                         # (at most: what other ranks' windows need of this slab)
        mark("potential_out")
        at = 0
        for t0, c, o, f in pot:
            if o == rank:
                e.own_potential_planes(f, t0, c)
            else:
                e.planes_in(1, t0, torch.zeros((c, G, G), dtype=torch.float32, device="cuda"))
        mark("potential_in")
        e.gather_kick_drift(dt)
        mark("gather_kick_drift")
        counts = e.outbox_counts().cpu().tolist()
        e.outbox_records(counts[:P])
        e.inbox(torch.empty((0, 8), dtype=torch.float32, device="cuda"))
        mark("migration")
        torch.cuda.synchronize()
        if record:
            for (_, a), (name, b) in zip(marks, marks[1:]):
                acc[name] += a.elapsed_time(b)

    step(False)
    for _ in range(args.steps):
        step(True)
    out = {k: round(v / args.steps, 3) for k, v in acc.items()}
    total = sum(out.values())
    plane_mb = e.deposit_plane_elems * 4 / 1e6
    line = {"grid": G, "slabs": P, "rank": rank, "window": [cuts[rank], cuts[rank + 1]], "particles": n, "phases_ms": out,
            "local_ms_per_step": round(total, 3),
            "planes_sent_MB": round(sent_planes * plane_mb, 1), "potential_received_MB": round(got_potential * G * G * 4 / 1e6, 1)}
    print(json.dumps(line), flush=True)
    e.close()
    return total


if args.workload == "uniform":
    rng = np.random.default_rng(7)
    n = args.n
    pos = np.empty((n, 3), np.float32)
    pos[:, 0] = (rng.random(n) * (L / P) * 0.998 + 0.001 * L / P).astype(np.float32)
    pos[:, 1:] = (rng.random((n, 2)) * L).astype(np.float32)
    if args.quiet_set:
        one_slab(0, pos, rng.normal(0, 10, (n, 3)).astype(np.float32), np.full(n, 1e-6, np.float32), np.arange(n, dtype=np.uint32))
    else:
        one_slab(0, pos, np.zeros((n, 3), np.float32), np.ones(n, np.float32), np.arange(n, dtype=np.uint32))
else:
    # a slab's share of a whole set, with the velocities set to zero so that nobody leaves (the halos are the slab's own
    # planes: a particle that left would be handed to a neighbour that is not there)
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from pm.distributed import slab_of, balanced_cuts, owner_by_cuts, plane_of
    pos, vel, mass = bench.make_workload(args.workload, args.n, L, 1052)
    cuts = balanced_cuts(np.bincount(plane_of(pos, G, L), minlength=G), P, min(G - 3, max(Gs, G // 2)), 0.5 * args.n / G) if args.windows else None
    owner = owner_by_cuts(pos, G, L, cuts) if args.windows else slab_of(pos, G, L, P)
    ids = np.arange(args.n, dtype=np.uint32)
    totals = {}
    for r in (range(P) if args.rank < 0 else [args.rank]):
        mine = owner == r
        if mine.sum() == 0:
            continue
        zero = np.zeros((int(mine.sum()), 3), np.float32)
        totals[r] = one_window(r, cuts, pos[mine], zero, mass[mine], ids[mine]) if args.windows else one_slab(r, pos[mine], zero, mass[mine], ids[mine])
    print(json.dumps({"workload": args.workload, "n": args.n, "grid": G, "slabs": P, "local_ms_per_slab": {str(k): round(v, 3) for k, v in totals.items()},
                      "largest": round(max(totals.values()), 3), "sum": round(sum(totals.values()), 3)}))
