"""The binning that starts from the cell order of the step before (csrc/pm_rebin.hip) against the sort of the whole set
(PM_FLAG_FULL_SORT): the cell order is `stable sort of the storage order by cell key` (functions.py:27-39 defines the
cells) whichever way it is computed, so keys, permutation, positions and velocities must agree bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

import pm  # noqa: E402

N = 1 << 21          # the incremental path is for sets of 2^20 particles and more
L = 100.0


def make(seed, sigma, clumped=False):
    rng = np.random.default_rng(seed)
    if clumped:
        pos = (0.5 * L + rng.normal(0, 0.04 * L, (N, 3))) % L
    else:
        pos = rng.random((N, 3)) * L
    vel = rng.normal(0, sigma, (N, 3))
    # (light particles: the kicks of six steps stay small beside these velocities, so the movers stay few)
    return pos.astype(np.float32), vel.astype(np.float32), np.full(N, 1e-4, np.float32)


def run_pair(G, pos, vel, mass, dts, fused=True):
    """The same steps on two contexts, one of them sorting everything: compares after every step, returns what the
    incremental one reported per step."""
    reports = []
    with pm.Simulation(G, L, pos, vel, mass, fused_gradient=fused) as a, \
            pm.Simulation(G, L, pos, vel, mass, fused_gradient=fused, full_sort=True) as b:
        for dt in dts:
            a.step(dt, 1)
            b.step(dt, 1)
            # the drift's estimate of its movers reaches the host without anyone waiting for it: wait here, so that the
            # choice the next binning makes is the same in every run
            a.synchronize(), b.synchronize()
            # (the binning of a step happens at its start: deposit once more to bin the drifted particles)
            a.deposit(), b.deposit()
            ka, pa = a.cell_keys()
            kb, pb = b.cell_keys()
            assert np.array_equal(ka, kb)
            assert np.array_equal(pa, pb)
            assert np.all(np.diff(ka) >= 0)
            reports.append(a.last_binning())
            assert b.last_binning()[0] is False
        assert np.array_equal(a.positions(), b.positions())
        assert np.array_equal(a.velocities(), b.velocities())
    return reports


@pytest.mark.parametrize("G,clumped", [(128, False), (64, True), (100, False)])
def test_few_movers_same_order(G, clumped):
    pos, vel, mass = make(3, 2.0, clumped)
    rep = run_pair(G, pos, vel, mass, [0.01] * 6)
    for inc, movers in rep:
        assert inc and 0 < movers < N // 8, rep


def test_order_from_the_order_before_against_the_oracle():
    """The binning that starts from the order before, directly against the oracle (not against another binning of this
    library): 2^21 particles on 128^3, five steps; after each the cell order -- stable sort by O.cell_keys of the storage
    order, which is the cell order of the step before -- keys and permutation bit for bit, and the trajectory against
    O.pm_step within 1e-5 L and 1e-5 max|v|."""
    from oracle import pm_oracle as O
    G, dt = 128, 0.01
    pos, vel, mass = make(11, 2.0)
    xo, vo = pos, vel
    storage = np.arange(N)
    with pm.Simulation(G, L, pos, vel, mass, fused_gradient=True) as sim:
        for step in range(6):
            keys, perm = sim.cell_keys()
            inc, movers = sim.last_binning()
            k_storage = O.cell_keys(sim.positions(), G, L)[storage]
            order = O.sort_order(k_storage)
            storage = storage[order]
            assert np.array_equal(perm, storage), step
            assert np.array_equal(keys, k_storage[order]), step
            if step > 0:
                assert inc is True and 0 < movers < N // 8, (step, inc, movers)
            sim.step(dt)
            sim.synchronize()
            xo, vo = O.pm_step(xo, vo, mass, G, L, dt)
        dx = np.abs(sim.positions().astype(np.float64) - xo)
        assert np.minimum(dx, L - dx).max() <= 1e-5 * L
        assert np.abs(sim.velocities() - vo).max() <= 1e-5 * np.abs(vo).max()


def test_nearly_everyone_moves():
    """The drift estimates its movers from a sample of its workgroups of 256 slots (of this set: every 8th).  Here exactly those slots hold particles
    at rest and all others move a cell or more: the estimate says `none`, the binning starts from the old order, finds
    7 of 8 particles moved -- and still arrives at the order the sort of everything gives."""
    G = 128
    pos, _, _ = make(5, 0.0)
    with pm.Simulation(G, L, pos, np.zeros((N, 3), np.float32), np.ones(N, np.float32)) as probe:
        probe.deposit()
        _, perm = probe.cell_keys()                      # perm[r]: the particle stored at slot r after the first binning
    slot = np.empty(N, np.int64)
    slot[perm] = np.arange(N)
    sampled = (slot // 256) % 8 == 0                # (2^21 particles: 8192 workgroups, every 8th makes 1024)
    vel = np.where(sampled[:, None], 0.0, 3.0 * L / G).astype(np.float32) * np.ones((1, 3), np.float32)
    mass = np.full(N, 1e-30, np.float32)                 # no forces to speak of: a particle at rest stays where it is
    rep = run_pair(G, pos, vel, mass, [1.0, 1.0])
    assert rep[0] == (True, int((~sampled).sum())), rep
    assert rep[1][0] is False and rep[1][1] > N // 2, rep                     # (the slots sampled now hold other particles: the estimate has caught up)


def test_staged_pipeline_and_wrapping_movers():
    """particles that cross the box's faces change to far-away keys; the staged pipeline (separate meshes) bins the same way"""
    rng = np.random.default_rng(9)
    pos = rng.random((N, 3)).astype(np.float32) * np.float32(L)
    pos[: N // 64] = (pos[: N // 64] * np.float32(0.004)) % np.float32(L)          # a corner clump next to three faces
    vel = rng.normal(0, 1.0, (N, 3)).astype(np.float32)
    vel[: N // 64] -= 3.0
    rep = run_pair(64, pos, vel, np.full(N, 1e-4, np.float32), [0.02] * 4, fused=False)
    assert all(r[0] for r in rep), rep


def test_slabs_with_arrivals_and_leavers():
    """Two slabs of a 128^3 mesh as threads on the one GPU, 2^21 particles each: particles change slabs every step (they
    leave with the key past all cells and arrive behind the slots in use), and the slabs that start from the order before
    end every step with the same particles, positions and velocities, bit for bit, as slabs that sort everything."""
    import threading
    from pm.distributed import SlabSimulation, ThreadComm, HipSlabEngine, slab_of
    G, world, n = 128, 2, 2 * N
    rng = np.random.default_rng(21)
    pos = (rng.random((n, 3)) * L).astype(np.float32)
    vel = rng.normal(0, 2.0, (n, 3)).astype(np.float32)
    mass = np.full(n, 1e-4, np.float32)
    ids = np.arange(n, dtype=np.uint32)
    owner = slab_of(pos, G, L, world)
    lock = threading.Lock()

    def run(full_sort):
        comms = ThreadComm.make(world)
        results, reports, errors = [None] * world, [None] * world, []

        def rank(r):
            try:
                mine = owner == r
                cap = int(mine.sum() * 1.2)
                with lock:
                    eng = HipSlabEngine(G, L, r * (G // world), G // world, cap, 0, full_sort=full_sort)
                    sim = SlabSimulation(G, L, pos[mine], vel[mine], mass[mine], ids[mine], device_index=0, capacity=cap, comm=comms[r], engine=eng)
                out, rep = [], []
                for _ in range(4):
                    sim.step(0.005)
                    out.append(sim.particles())
                    rep.append(eng.last_binning())
                sim.synchronize()
                sim.close()
                results[r], reports[r] = out, rep
            except BaseException as e:                  # a thread that dies must not leave the other at a barrier
                errors.append(e)
                comms[r]._s.barrier.abort()

        threads = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(600)
        assert not errors, errors
        return results, reports

    inc, rep_inc = run(False)
    full, rep_full = run(True)
    for r in range(world):
        assert all(not x[0] for x in rep_full[r]), rep_full[r]
        assert all(x[0] and x[1] > 0 for x in rep_inc[r]), rep_inc[r]
        moved = 0
        for (ia, xa, va), (ib, xb, vb) in zip(inc[r], full[r]):
            assert np.array_equal(ia, ib) and np.array_equal(xa, xb) and np.array_equal(va, vb)
            moved += int((owner[ia] != r).sum())
        assert moved > 0                                # particles did change slabs


def test_small_sets_when_the_environment_allows(monkeypatch):
    """PM_REBIN_MIN (read when a context is made) lowers the size from which a binning may start from the order before:
    a few thousand particles, whose estimate comes from the first 256 slots alone, against the sort of everything."""
    monkeypatch.setenv("PM_REBIN_MIN", "256")
    rng = np.random.default_rng(4)
    n = 6000
    pos = (rng.random((n, 3)) * L).astype(np.float32)
    vel = rng.normal(0, 1.0, (n, 3)).astype(np.float32)
    mass = np.full(n, 1e-3, np.float32)
    took = 0
    with pm.Simulation(32, L, pos, vel, mass, fused_gradient=True) as a, \
            pm.Simulation(32, L, pos, vel, mass, fused_gradient=True, full_sort=True) as b:
        for _ in range(12):
            a.step(0.05, 1), b.step(0.05, 1)
            a.synchronize(), b.synchronize()
            a.deposit(), b.deposit()
            ka, pa = a.cell_keys()
            kb, pb = b.cell_keys()
            assert np.array_equal(ka, kb) and np.array_equal(pa, pb)
            took += a.last_binning()[0]
        assert np.array_equal(a.positions(), b.positions()) and np.array_equal(a.velocities(), b.velocities())
    assert took > 0


def test_last_binning_reports_errors_as_negative_values():
    """0 and 1 are results of pm_last_binning: its errors are negated codes, which Simulation.last_binning raises on"""
    import ctypes
    lib = pm.load()
    movers = ctypes.c_int64(7)
    assert lib.pm_last_binning(None, ctypes.byref(movers)) == -1        # -PM_EINVAL
    with pm.Simulation(8, 100.0, np.zeros((2, 3), np.float32)) as sim:
        assert lib.pm_last_binning(sim.ctx, None) == -1
        assert b"NULL" in lib.pm_last_error(sim.ctx)
        assert sim.last_binning() == (False, -1)                        # nothing binned yet, no estimate yet
