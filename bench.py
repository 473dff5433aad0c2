#!/usr/bin/env python3
"""Benchmark of the PM step on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of deposit -> Poisson solve -> gradient -> gather + kick + drift over the
whole particle set, with all state resident in HBM before the timed region starts.  At N=1 the
workload is the headline configuration: a two-galaxy collision of 16,777,216 particles on a
512^3 mesh.  With --gpus N > 1 the same workload runs as ONE simulation over N slabs of the mesh
(strong scaling; one process per GPU, exchanges over RCCL), started either by torch.distributed.run
or, without a launcher, by this script itself.  Prints ONE JSON line (rank 0).

The metric is particle-updates per second.  `roofline` prices the dominant stage against the
8 TB/s HBM peak using the algorithmic bytes of that stage (one read and one write of each
operand per stage: deposit 16N+4M, forward FFT 24M, inverse FFT 24M, gradient 16M,
gather+kick+drift 48N+12M; whole step 64N+80M).  `cpu_baseline` is the CPU oracle (single
thread, as the reference ships) timed on one step of the same workload.
"""
import argparse
import datetime
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "nbody-simulation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBPS = 8000.0
BOX = 10000
DT = 0.005


def plummer_blob(rng, n, scale, rmax):
    """Plummer sphere positions (density ~ (1 + r^2/a^2)^-5/2), truncated at rmax."""
    out = np.empty((n, 3), np.float32)
    done = 0
    while done < n:
        m = int((n - done) * 1.1) + 16
        u = rng.random(m)
        r = scale / np.sqrt(u ** (-2.0 / 3.0) - 1.0)
        r = r[r < rmax][: n - done]
        k = len(r)
        d = rng.normal(size=(k, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        out[done:done + k] = (d * r[:, None]).astype(np.float32)
        done += k
    return out


def warm_up(sim, steps):
    """`steps` untimed steps.  The library chooses how it brings a step's particles into cell order by an estimate the
    step before sends to the host without anyone waiting for it: the host waits once after the first step, so that the
    steps after it already run the way the timed ones will (and load the kernels of that way now, not in the timed region)."""
    if steps > 1:
        sim.step(DT, 1)
        sim.synchronize()
        sim.step(DT, steps - 1)
    elif steps == 1:
        sim.step(DT, 1)


def make_workload(name, n, L, seed):
    """Synthetic particle sets (float32 positions, velocities, masses)."""
    rng = np.random.default_rng(seed)
    if name == "uniform":
        pos = (rng.random((n, 3)) * L).astype(np.float32)
        vel = rng.normal(0, 10, (n, 3)).astype(np.float32)
        return pos, vel, np.ones(n, np.float32)
    if name == "two_galaxy":
        # two Plummer blobs (scale radii 0.02 L and 0.05 L) placed +-0.1 L apart along (0.67, 0.33, 0)
        # about the box centre, relative velocity (-3000, 300, 0) split for zero total momentum
        n1 = n // 2
        n2 = n - n1
        d = np.array([0.67, 0.33, 0.0])
        d /= np.linalg.norm(d)
        c = 0.5 * L
        p1 = plummer_blob(rng, n1, 0.02 * L, 0.45 * L) + (c - 0.1 * L * d).astype(np.float32)
        p2 = plummer_blob(rng, n2, 0.05 * L, 0.45 * L) + (c + 0.1 * L * d).astype(np.float32)
        vrel = np.array([-3000.0, 300.0, 0.0])
        v1 = rng.normal(0, 10, (n1, 3)) - 0.5 * vrel
        v2 = rng.normal(0, 10, (n2, 3)) + 0.5 * vrel
        pos = np.vstack((p1, p2)).astype(np.float32)
        pos %= np.float32(L)
        vel = np.vstack((v1, v2)).astype(np.float32)
        return pos, vel, np.ones(n, np.float32)
    if name == "disk":
        # exponential disk rho ~ exp(-R/Rd) exp(-|z|/z0), Rd = 0.03 L, z0 = 0.003 L, equal masses
        R = rng.gamma(2.0, 0.03 * L, n)
        phi = rng.uniform(0, 2 * np.pi, n)
        z = rng.laplace(0.0, 0.003 * L, n)
        pos = np.stack((0.5 * L + R * np.cos(phi), 0.5 * L + R * np.sin(phi), 0.5 * L + z), axis=1)
        pos = (pos % L).astype(np.float32)
        vel = rng.normal(0, 10, (n, 3)).astype(np.float32)
        return pos, vel, np.ones(n, np.float32)
    raise ValueError(name)


def stage_bytes(n, m, fused=False):
    """Algorithmic bytes of each timed stage as the per-stage figures that add up to 64 N + 80 M: deposit 16N + 4M,
    three forward axis passes 24M, Green multiply 0, three inverse passes 24M, gradient 16M, gather 48N + 12M.  In the
    fused pipeline (G = 512 / 1024) a timed stage covers more than one of these: the deposit kernels also do the forward
    z pass, `green` is the forward and inverse x passes around the multiply, the gather also differences the potential."""
    if fused:
        return {"bin": 0, "deposit": 16 * n + 4 * m + 8 * m, "fft_fwd": 8 * m, "green": 16 * m, "fft_inv": 16 * m,
                "gradient": 0, "gather": 48 * n + 12 * m + 16 * m}
    return {"bin": 0, "deposit": 16 * n + 4 * m, "fft_fwd": 24 * m, "green": 0, "fft_inv": 24 * m,
            "gradient": 16 * m, "gather": 48 * n + 12 * m}


def kernel_bytes(n, m, fused=True):
    """Single kernel launches that pm_kernel_times reports, with the name rocprofv3 gives the kernel and the bytes one
    launch cannot avoid moving.  The row kernel of the deposit is priced with the whole deposit's figure (it reads the
    16-byte fraction records and writes the 4M bytes of the z-transformed density; the kernels beside and behind it
    -- crowded rows in units, their sum -- share the same records and have no bytes of their own in this table); a
    column pass reads and writes the half spectrum once (8M, also when it transforms forward and back in one go); the
    inverse row transform reads the half spectrum and writes the potential (4M + 4M); the gather reads and writes
    position and velocity (48N) and reads the potential once (4M; in the staged pipeline the three gradient meshes,
    12M).  The sort passes and the reorder have no algorithmic bytes (cell order is not part of 64N + 80M): they are
    reported with their durations only and cannot be the roofline kernel."""
    return {"deposit_rows": ("k_deposit_rows_loop", 16 * n + 4 * m), "col_y_fwd": ("k_col_fft<G, -1, false>", 8 * m),
            "col_x_green": ("k_col_fft<G, -1, true>", 8 * m), "col_y_inv": ("k_col_fft<G, 1, false>", 8 * m),
            "row_inv": ("k_row_c2r", 8 * m), "gather": ("k_gather", 48 * n + (4 * m if fused else 12 * m)),
            "sort_pass": ("k_radix_pass", 0), "reorder": ("k_reorder", 0)}


def fused_stage_bytes(n, m):
    """What each timed stage of the fused pipeline (G = 512 / 1024) cannot avoid moving: the deposit reads the 16-byte
    fraction records and writes the half spectrum (4M) already transformed along z; each column pass reads and writes
    the half spectrum once (8M; the pass along x does so once for forward, Green multiply and inverse together);
    `fft_inv` is the inverse pass along y plus the inverse row transform (4M in, 4M out); the gather reads and writes
    position and velocity records (48N) and reads the potential once (4M).  These are smaller than the per-stage shares
    of 64N + 80M, which count every mesh the staged pipeline materialises."""
    return {"bin": 0, "deposit": 16 * n + 4 * m, "fft_fwd": 8 * m, "green": 8 * m, "fft_inv": 16 * m, "gradient": 0,
            "gather": 48 * n + 4 * m}


# kernels that make up each timed stage (substrings of the names rocprofv3 reports)
STAGE_KERNELS = {"bin": ["k_radix_pass", "k_tile_counts", "k_chunk_sums", "k_scan_chunks", "k_digit_counts", "k_digit_offsets", "k_reorder", "k_row_start",
                         "k_cell_keys", "fillBuffer", "k_clear", "k_scan_movers", "k_compact_movers", "k_place_movers", "k_tile_first", "k_place_stayers",
                         "k_row_start_update"],
                 "deposit": ["k_deposit_rows", "k_deposit_units", "k_sum_units"],
                 "fft_fwd": ["k_row_r2c", "k_col_fft<512, -1, false>"], "green": ["k_col_fft<512, -1, true>", "k_green"],
                 "fft_inv": ["k_col_fft<512, 1, false>", "k_row_c2r"], "gradient": ["k_gradient"], "gather": ["k_gather"]}


def measured_traffic(names, per_launch=False, workload="two_galaxy"):
    """HBM bytes of the kernels whose names contain one of `names`, from the newest committed rocprofv3 counter passes
    of the headline workload (profiles/r*_pmc_traffic.json; the uniform set of the same size: r*_pmc_traffic_uniform.json:
    FETCH_SIZE and WRITE_SIZE in separate --pmc runs, KiB units, FETCH_SIZE doubled for gfx950's half-counted wide
    reads): per step, or per launch of the kernel.  None if there is no such file or kernel."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json" if workload == "two_galaxy" else f"r*_pmc_traffic_{workload}.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        per_kernel = json.load(f)["kernels"]
    total = 0.0
    for name, v in per_kernel.items():
        if any(sub in name for sub in names):
            b = v["read_bytes_per_step"] + v["write_bytes_per_step"]
            total += b / max(v.get("launches_per_step", 1), 1) if per_launch else b
    return total or None


def cpu_baseline(pos, vel, mass, G, L, dt):
    """One step of the CPU oracle on the same workload, single thread."""
    from oracle import pm_oracle as O
    O.build()
    O.cell_keys(pos[:1000], G, L)                 # load the library outside the timed region
    steps = 3
    t0 = time.perf_counter()
    x, v = pos, vel
    for _ in range(steps):
        x, v = O.pm_step(x, v, mass, G, L, dt)
    dt_s = time.perf_counter() - t0
    return {"value": steps * len(pos) / dt_s, "unit": "particle-updates/s", "cores": 1, "kind": "port",
            "sample": f"{steps} full PM steps of the same workload (N={len(pos)}, G={G}) in {dt_s:.1f} s: C loops for "
                      "deposit/gather, numpy/scipy.fft (1 worker) for Poisson/gradient/integrator",
            "host_cpus": os.cpu_count()}


def metric_name(args):
    return f"particle-updates/sec per PM step (N={args.n}, {args.grid}^3 mesh); achieved HBM GB/s"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="two_galaxy", choices=["two_galaxy", "uniform", "disk"])
    ap.add_argument("--n", "--particles", dest="n", type=int, default=16_777_216)
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--no-fused-gradient", action="store_true")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "skip"])
    ap.add_argument("--also-uniform", action="store_true", help="(the default now; kept so that old command lines still parse)")
    ap.add_argument("--no-uniform", action="store_true",
                    help="one GPU, default workload: do not also time the uniform particle set of the same size (reported as "
                         "`uniform_set` in the same line)")
    ap.add_argument("--device-ic", action="store_true",
                    help="draw the workload on the GPU (pm_generate: same distributions, a counter-based random stream) "
                         "instead of building it with NumPy on the host; meant for the 64M+ particle configurations")
    ap.add_argument("--no-slab-probe", action="store_true", help="(no longer does anything; kept so that old command lines still parse)")
    ap.add_argument("--repeats", type=int, default=4,
                    help="single GPU: after the timed K steps, time this many more blocks of K steps and report every block "
                         "(`ms_per_step_blocks`, `ms_per_step_median`); `value` stays the first block, as the contract says")
    ap.add_argument("--parallelism", default=None, choices=["single", "slab", "replicas"],
                    help="single: one context on one GPU (the default for --gpus 1).  slab: ONE run of the workload with the mesh cut "
                         "into N slabs -- all-to-all transposes, halo planes and particle migration over RCCL (the default for "
                         "--gpus N > 1; with --gpus 1 it runs the same code as one slab).  replicas: N independent copies of the "
                         "single-GPU run, no collective")
    ap.add_argument("--transpose-chunks", type=int, default=None,
                    help="slab runs: groups of planes the two all-to-all transposes are cut into and overlapped with the transforms "
                         "(default: 2 from a 1024^3 mesh on, else 1)")
    ap.add_argument("--balance", default="auto", choices=["auto", "on", "off"],
                    help="slab runs: hold the particles by windows of equal particle count instead of by slab (auto: when the "
                         "fullest slab would hold more than 1.2 shares of the set)")
    ap.add_argument("--balance-every", type=int, default=8,
                    help="slab runs with windows: every so many steps the windows are cut anew if the fullest rank is 1.2 shares "
                         "or more (0: the windows stay as they were cut at the start)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="--parallelism slab: torch.distributed backend (gloo stages the exchanges through the host: for "
                         "rehearsing the multi-rank run on a box with fewer GPUs than ranks)")
    args = ap.parse_args()
    if args.parallelism is None:
        args.parallelism = "single" if args.gpus == 1 else "slab"
    if args.parallelism == "single" and args.gpus != 1:
        ap.error("--parallelism single is one GPU")
    return args


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes of this one -- which has not
    touched the GPU and does not -- with the environment torch.distributed.run would give them, pass rank 0's output
    through, and return the worst exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    children = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                         stdout=None if r == 0 else subprocess.DEVNULL))
    worst = 0
    try:
        pending = list(children)
        while pending:
            for ch in list(pending):
                rc = ch.poll()
                if rc is None:
                    continue
                pending.remove(ch)
                worst = worst or rc
                if rc != 0:                         # a rank died: the others would wait for it for ever
                    for other in pending:
                        other.terminate()
            time.sleep(0.05)
    finally:
        for ch in children:
            if ch.poll() is None:
                ch.kill()
    return worst


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        import torch                              # (counting devices does not touch the GPU)
        local_rank %= max(1, torch.cuda.device_count())       # more ranks than GPUs: rehearsals on a one-GPU box

    if args.parallelism == "slab":
        import torch                              # noqa: F401  (its HIP runtime has to be the one libpm.so binds to)
    import pm                                     # loads libpm.so (raises if it is missing)
    pm.load()
    if args.parallelism == "slab":
        return slab_main(args, rank, world, local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist          # rendezvous and barriers only
        dist.init_process_group("gloo", rank=rank, world_size=world)

    n, G, L = args.n, args.grid, BOX
    m = G ** 3
    # replicas: every rank holds an independent copy of the workload (its own seed)
    fused = not args.no_fused_gradient
    if args.device_ic:
        sim = pm.Simulation(G, L, capacity=n, device=local_rank, fused_gradient=fused)
        sim.generate(args.workload, n, 1052 + rank)
        pos = vel = mass = None
        if args.cpu_baseline == "auto" and world == 1:
            pos, vel, mass = sim.positions(), sim.velocities(), np.ones(n, np.float32)
    else:
        pos, vel, mass = make_workload(args.workload, n, L, 1052 + rank)
        sim = pm.Simulation(G, L, pos, vel, mass, device=local_rank, fused_gradient=fused)

    def barrier():
        if dist is not None:
            dist.barrier()
        sim.synchronize()

    def timed_block():
        barrier()
        t0 = time.perf_counter()
        sim.step(DT, args.steps)
        barrier()
        return time.perf_counter() - t0

    warm_up(sim, args.warmup)
    elapsed = timed_block()
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    # per-stage and per-kernel durations: HIP events on the engine's stream around the launches, the K steps that follow
    # the timed ones (the collision goes on and the particle kernels slow down as the galaxies spread: these steps are
    # the nearest to the ones timed above)
    sim.set_flags(timing=True, fused_gradient=fused)
    sim.step(DT, args.steps)
    stages = sim.stage_times()
    kernels = sim.kernel_times()
    sim.set_flags(timing=False, fused_gradient=fused)

    # how steady the number is: the same block of K steps a few more times (the particles keep moving: not the same steps)
    blocks = [elapsed / args.steps * 1e3] + [timed_block() / args.steps * 1e3 for _ in range(max(args.repeats, 0) if world == 1 else 0)]

    binning = sim.last_binning()
    sim.close()
    # beside the headline: the uniform particle set of the same size on the same mesh (one GPU, default workload only)
    also = None
    if world == 1 and args.workload == "two_galaxy" and not args.no_uniform:
        up, uv, um = make_workload("uniform", n, L, 1052)
        with pm.Simulation(G, L, up, uv, um, device=local_rank, fused_gradient=fused) as usim:
            warm_up(usim, args.warmup)
            usim.synchronize()
            t0 = time.perf_counter()
            usim.step(DT, args.steps)
            usim.synchronize()
            ue = time.perf_counter() - t0
        also = {"workload": f"uniform N={n} G={G} L={L} dt={DT}", "value": n * args.steps / ue, "unit": "particle-updates/s",
                "ms_per_step": ue / args.steps * 1e3,
                "step_roofline_frac": (64 * n + 80 * m) / (ue / args.steps) / 1e9 / HBM_PEAK_GBPS,
                # its longest kernel is the gather (particles spread over the whole mesh): HBM bytes per launch from the counter passes
                "gather_traffic": measured_traffic(["k_gather"], per_launch=True, workload="uniform") if n == 16_777_216 and G == 512 else None}
        del up, uv, um

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        custom = fused and any(c for _, c in (kernels["col_x_green"],))     # the hand-written transforms ran: fused pipeline
        b = stage_bytes(n, m, custom)
        sb = fused_stage_bytes(n, m) if custom else b
        per = {s: (ms / c if c else 0.0) for s, (ms, c) in stages.items()}
        kper = {k: (ms / c if c else 0.0) for k, (ms, c) in kernels.items()}
        # the dominant kernel: the single launch that runs longest among those that have algorithmic bytes
        kb = kernel_bytes(n, m, fused)
        dom = max((k for k in kb if kb[k][1] > 0), key=lambda k: kper[k])
        dom_name, dom_bytes = kb[dom][0].replace("<G,", f"<{G},"), kb[dom][1]
        ach = dom_bytes / (kper[dom] * 1e-3) / 1e9 if kper[dom] > 0 else 0.0
        headline = args.workload in ("two_galaxy", "uniform") and n == 16_777_216 and G == 512       # (sets the counter passes were run on)
        b_step, b_pipe = 64 * n + 80 * m, 64 * n + (40 * m if custom else 80 * m)
        bin_traffic = measured_traffic(STAGE_KERNELS["bin"], workload=args.workload) if headline else None
        out = {
            "metric": metric_name(args),
            "value": world * n * args.steps / elapsed, "unit": "particle-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            # (--gpus N runs this same workload over N GPUs; replicas are N copies of it)
            "higher_is_better": True, "scaling": "weak" if world > 1 else "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (drawn on the device)" if args.device_ic else "synthetic",
            "config": {"workload": f"{args.workload} N={n} G={G} L={L} dt={DT}", "fused_gradient": fused,
                       "parallelism": "single GPU" if world == 1 else f"{world} independent replicas"},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBPS,
                         "traffic": measured_traffic([kb[dom][0].split("<")[0]], per_launch=True, workload=args.workload) if headline else None,
                         "algorithmic_bytes_per_launch": dom_bytes, "ms_per_launch": kper[dom],
                         "launches_timed": kernels[dom][1],
                         # the two longest launches are within a few microseconds of each other on this workload and
                         # change places from run to run: every priced launch, longest first, against the same peak
                         "ranking": [{"kernel": kb[k][0].replace("<G,", f"<{G},"), "ms_per_launch": round(kper[k], 5),
                                      "frac": round(kb[k][1] / (kper[k] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
                                     for k in sorted((k for k in kb if kb[k][1] > 0 and kper[k] > 0), key=lambda k: -kper[k])]},
            # the whole step against 64N + 80M (the staged pipeline's compulsory traffic, every mesh written and read once)
            # and against what THIS pipeline cannot avoid (fused: no density mesh, no gradient meshes: 64N + 40M)
            "step_roofline": {"algorithmic_bytes_per_step": b_step, "achieved": b_step / (ms_step * 1e-3) / 1e9,
                              "unit": "GB/s", "frac": b_step / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                              "pipeline_bytes_per_step": b_pipe, "pipeline_frac": b_pipe / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBPS},
            "ms_per_step_blocks": [round(x, 4) for x in blocks], "ms_per_step_median": float(np.median(blocks)),
            # how the last step's particles were brought into cell order: put together from the order before (then with the
            # exact number of particles that had changed their cell), or all sorted (then with the drift's estimate of it)
            "last_binning": {"from_order_before": binning[0], "movers": binning[1]},
            "stage_ms": per,
            "kernel_ms": {k: round(v, 5) for k, v in kper.items()},
            # every timed stage against the same peak, with the bytes the stage cannot avoid in the pipeline that ran; the
            # binning has none (cell order is not part of the algorithmic traffic): it is listed with what the counters saw
            "stage_roofline": {st: {"bytes": sb[st], "achieved": sb[st] / (per[st] * 1e-3) / 1e9, "unit": "GB/s",
                                    "frac": sb[st] / (per[st] * 1e-3) / 1e9 / HBM_PEAK_GBPS} for st in per if sb[st] > 0 and per[st] > 0},
        }
        if per["bin"] > 0:
            out["stage_roofline"]["bin"] = {"bytes": 0, "measured_traffic": bin_traffic,
                                            "achieved": (bin_traffic / (per["bin"] * 1e-3) / 1e9) if bin_traffic else None, "unit": "GB/s",
                                            "frac": 0.0}
        if args.cpu_baseline == "auto" and world == 1:
            out["cpu_baseline"] = cpu_baseline(pos, vel, mass, G, L, DT)
        if also is not None:
            out["uniform_set"] = also
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def slab_main(args, rank, world, local_rank):
    """ONE simulation of the workload, the mesh cut into `world` slabs: strong scaling.  --balance: the particles are
    held by windows of equal particle count instead of by slab (auto: when the fullest slab would hold more than 1.2
    shares of the set).  --device-ic: every rank draws its share of the set on its GPU (no host arrays)."""
    import torch
    import torch.distributed as dist
    from pm import _lib
    from pm.distributed import SlabSimulation, balanced_cuts, equal_cuts, generated_plane_counts, owner_by_cuts, plane_of
    torch.cuda.set_device(local_rank)
    red = "cuda" if args.backend == "nccl" else "cpu"        # where the few scalars reduced below live
    if world > 1:
        if args.backend == "nccl":
            # (a rank that never arrives should end the run in minutes, not after the library's ten)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(seconds=240))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    n, G, L = args.n, args.grid, BOX
    m = G ** 3
    Gs = G // world

    def make_sim(workload):
        """(simulation, particles per rank at the start, the windows' cuts or None)"""
        seed = 1052
        if args.device_ic:
            counts = generated_plane_counts(G, L, workload, n, seed, local_rank)
            pos = None
        else:
            pos, vel, mass = make_workload(workload, n, L, seed)
            counts = np.bincount(plane_of(pos, G, L), minlength=G)
        by_slab = counts.reshape(world, Gs).sum(axis=1)
        balance = world > 1 and (args.balance == "on" or (args.balance == "auto" and by_slab.max() > 1.2 * n / world))
        wmax = min(G - 3, max(Gs, G // 2))
        cuts = balanced_cuts(counts, world, wmax, 0.5 * n / G) if balance else equal_cuts(G, world)
        loads = [int(counts[a:b].sum()) for a, b in zip(cuts, cuts[1:])]
        cap = int(max(loads[rank], n // world) * 1.3) + (1 << 20)
        kw = dict(device_index=local_rank, capacity=cap, transpose_chunks=args.transpose_chunks)
        if balance:
            kw.update(cuts=cuts, window_planes=wmax, balance_every=args.balance_every, balance_threshold=1.2)
        if args.device_ic:
            sim = SlabSimulation(G, L, None, None, None, None, generate=(workload, n, seed, None), **kw)
        else:
            mine = owner_by_cuts(pos, G, L, cuts) == rank
            sim = SlabSimulation(G, L, pos[mine], vel[mine], mass[mine], np.arange(n, dtype=np.uint32)[mine], **kw)
        return sim, loads, (cuts if balance else None)

    def max_over_ranks(values):
        t = torch.tensor(values, dtype=torch.float64, device=red)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t]

    sim, loads, cuts = make_sim(args.workload)
    warm_up(sim, args.warmup)
    sim.synchronize()
    t0 = time.perf_counter()
    sim.step(DT, args.steps)
    sim.synchronize()
    elapsed = max_over_ranks([time.perf_counter() - t0])[0]
    # where the time goes: a few more steps with a mark after every phase, and HIP events around the kernels of this rank's
    # context (largest value over the ranks for the phases; the kernels of the rank whose step is the longest)
    sim.profile = True
    sim.engine.set_timing(True)
    k_prof = max(1, min(args.steps, 5))
    sim.step(DT, k_prof)
    phases = sim.phase_times()
    kernels = sim.engine.kernel_times()
    sim.engine.set_timing(False)
    sim.profile = False
    local_ms = sum(phases[p] for p in ("deposit", "fft_zy", "solve_x", "ifft_yz", "gather_kick_drift")) / k_prof
    phases = dict(zip(sim.PHASES, (round(x, 4) for x in max_over_ranks([phases[p] / k_prof for p in sim.PHASES]))))
    kper = [kernels[k][0] / kernels[k][1] if kernels[k][1] else 0.0 for k in _lib.KERNELS]
    n_now = sim.engine.n
    # every rank's local work, kernel averages and particle count, gathered on all ranks (a few numbers each)
    mine = torch.tensor([local_ms, float(n_now)] + kper, dtype=torch.float64, device=red)
    every = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(every, mine)
    else:
        every = [mine]
    every = [[float(x) for x in t] for t in every]
    slowest = max(range(world), key=lambda r: every[r][0])
    group_world = dist.get_world_size() if world > 1 else 1
    group_backend = dist.get_backend() if world > 1 else None
    final_cuts, recuts = list(sim.cuts), sim.rebalanced
    sim.close()
    # beside the headline: the uniform set of the same size over the same slabs (it divides evenly, the collision does not)
    also = None
    if args.workload == "two_galaxy" and not args.no_uniform:
        usim, _, _ = make_sim("uniform")
        warm_up(usim, args.warmup)
        usim.synchronize()
        t0 = time.perf_counter()
        usim.step(DT, args.steps)
        usim.synchronize()
        ue = max_over_ranks([time.perf_counter() - t0])[0]
        usim.close()
        also = {"workload": f"uniform N={n} G={G} L={L} dt={DT}", "value": n * args.steps / ue, "unit": "particle-updates/s",
                "ms_per_step": ue / args.steps * 1e3,
                "step_roofline_frac": (64 * n + 80 * m) / (ue / args.steps) / 1e9 / (HBM_PEAK_GBPS * world)}
    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        b_step = 64 * n + 80 * m
        ach = b_step / (ms_step * 1e-3) / 1e9
        # the dominant kernel of the slowest rank: the longest single launch among those with algorithmic bytes, priced with
        # the bytes that launch cannot avoid on that rank's share (its particles; its planes of the mesh)
        n_slow = int(every[slowest][1])
        kb = kernel_bytes(n_slow, m // world, True)
        kslow = dict(zip(_lib.KERNELS, every[slowest][2:]))
        priced = [k for k in kb if kb[k][1] > 0 and kslow[k] > 0]
        out = {
            "metric": metric_name(args), "value": n * args.steps / elapsed,
            "unit": "particle-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (drawn on the device)" if args.device_ic else "synthetic",
            "config": {"workload": f"{args.workload} N={n} G={G} L={L} dt={DT}", "parallelism": f"{world} slabs of {Gs} planes",
                       "particle_ownership": "windows of equal particle count" if cuts else "by slab",
                       "window_cuts_at_start": cuts, "window_cuts_at_end": final_cuts if cuts else None, "recuts": recuts,
                       "particles_per_rank_at_start": loads, "particles_per_rank_at_end": [int(e[1]) for e in every],
                       "backend": group_backend, "collective_world_size": group_world, "transpose_chunks": sim.transpose_chunks},
            "phases_ms": phases, "local_work_ms_per_rank": [round(e[0], 4) for e in every],
            "kernel_ms_per_rank": [dict(zip(_lib.KERNELS, (round(x, 5) for x in e[2:]))) for e in every],
            "uniform_set": also,
            "step_roofline": {"bound": "hbm", "what": "whole step, all slabs", "achieved": ach, "peak": HBM_PEAK_GBPS * world, "unit": "GB/s",
                              "frac": ach / (HBM_PEAK_GBPS * world), "algorithmic_bytes_per_step": b_step}}
        if priced:
            dom = max(priced, key=lambda k: kslow[k])
            dach = kb[dom][1] / (kslow[dom] * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": kb[dom][0].replace("<G,", f"<{G},"), "rank": slowest, "achieved": dach, "peak": HBM_PEAK_GBPS,
                               "unit": "GB/s", "frac": dach / HBM_PEAK_GBPS, "traffic": None, "algorithmic_bytes_per_launch": kb[dom][1],
                               "ms_per_launch": kslow[dom], "particles_on_that_rank": n_slow}
        else:
            out["roofline"] = {"bound": "hbm", "kernel": None, "achieved": ach, "peak": HBM_PEAK_GBPS * world, "unit": "GB/s",
                               "frac": ach / (HBM_PEAK_GBPS * world), "traffic": None}
        if args.cpu_baseline == "auto" and world == 1 and not args.device_ic:
            pos, vel, mass = make_workload(args.workload, n, L, 1052)
            out["cpu_baseline"] = cpu_baseline(pos, vel, mass, G, L, DT)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
