"""pm_energies_noalloc (include/pm.h): the energy diagnostics for callers that cannot allocate.  Through the C ABI,
against pm_energies -- same kernels, so the same bits -- on two calls in a row, the second with a smaller input (the
library's per-GPU scratch array still holds the first call's partial sums then), and its size limit."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pm():
    import pm as _pm
    _pm.load()
    return _pm


def _set(n, L, seed):
    rng = np.random.default_rng(seed)
    pos = (rng.random((n, 3)) * L).astype(np.float32)
    vel = rng.normal(0, 300, (n, 3)).astype(np.float32)
    mass = (rng.random(n) + 0.5).astype(np.float32)
    return pos, vel, mass


def _both(sim, out):
    """(pm_energies, pm_energies_noalloc) on the engine's present state; `out` = two float64 on the device."""
    out.fill_(-1.0)
    import torch
    torch.cuda.synchronize()
    sim.energies_noalloc(out.data_ptr())
    sim.synchronize()
    got = tuple(out.cpu().tolist())
    return sim.energies(), got


def test_noalloc_energies_equal_pm_energies_on_two_calls_in_a_row(pm):
    import torch
    G, L, dt = 64, 10000, 0.005
    out = torch.empty(2, dtype=torch.float64, device="cuda:0")
    big, small = _set(300_000, L, 1), _set(700, L, 2)           # 1024 partial sums per energy, then 3
    with pm.Simulation(G, L, *big, fused_gradient=True) as sim:
        sim.step(dt, 2)                                         # a potential mesh; velocity records one reorder behind
        ref, got = _both(sim, out)
        assert got == ref and ref[0] > 0 and ref[1] != 0
        sim.set_particles(*small)                               # same context, same potential, fewer particles
        ref2, got2 = _both(sim, out)
        assert got2 == ref2 and ref2 != ref
        # a second context on the same GPU shares the array; float64 velocities take the other kernel instantiation
        with pm.Simulation(32, L, small[0].astype(np.float64), small[1].astype(np.float64), small[2]) as other:
            other.set_mesh(1, np.random.default_rng(5).normal(0, 1, (32, 32, 32)).astype(np.float32))
            ref3, got3 = _both(other, out)
            assert got3 == ref3
        ref4, got4 = _both(sim, out)
        assert got4 == ref4 == ref2
    # and the reference's own definition of the two sums (functions.py:352-361) on the small set
    from oracle import pm_oracle as O
    with pm.Simulation(32, L, *small) as sim:
        phi = np.random.default_rng(6).normal(0, 1, (32, 32, 32)).astype(np.float32)
        sim.set_mesh(1, phi)
        ref, got = _both(sim, out)
        assert got == ref
        assert abs(got[0] - O.compute_kinetic_energy(small[1], small[2])) <= 1e-6 * got[0]
        assert abs(got[1] - O.compute_potential_energy(small[0], small[2], phi, 32, L)) <= 1e-5 * abs(got[1]) + 1e-3


def test_noalloc_energies_arguments_are_checked(pm):
    import torch
    lib = pm.load()
    out = torch.zeros(2, dtype=torch.float64, device="cuda:0")
    assert lib.pm_energies_noalloc(None, out.data_ptr()) == 1           # PM_EINVAL
    with pm.Simulation(8, 100.0, capacity=4) as sim:
        with pytest.raises(pm.PMError, match="no particles"):
            sim.energies_noalloc(out.data_ptr())
        sim.set_particles(np.zeros((2, 3), np.float32))
        with pytest.raises(pm.PMError, match="NULL"):
            sim.energies_noalloc(0)
        # no solve has run and no potential was uploaded: both energy calls refuse instead of summing whatever the mesh holds
        with pytest.raises(pm.PMError, match="no potential"):
            sim.energies_noalloc(out.data_ptr())
        with pytest.raises(pm.PMError, match="no potential"):
            sim.energies()
        sim.set_mesh(1, np.ones((8, 8, 8), np.float32))
        sim.energies_noalloc(out.data_ptr())
        sim.synchronize()
        assert out.cpu().tolist() == [0.0, 1.0]                             # two unit masses at rest on a potential of ones


def test_noalloc_energies_refuse_more_particles_than_the_scratch_is_sized_for(pm):
    """PM_NOALLOC_MAX_PARTICLES = 2^28; the particles are drawn on the device (no host arrays of that size)."""
    import torch
    limit = 1 << 28
    out = torch.zeros(2, dtype=torch.float64, device="cuda:0")
    with pm.Simulation(16, 10000.0, capacity=limit + 256) as sim:
        sim.set_mesh(1, np.ones((16, 16, 16), np.float32))
        sim.generate("uniform", limit + 256, seed=3)
        with pytest.raises(pm.PMError, match="at most PM_NOALLOC_MAX_PARTICLES"):
            sim.energies_noalloc(out.data_ptr())
        sim.generate("uniform", limit, seed=3)
        ref, got = _both(sim, out)
        assert got == ref and abs(ref[1] - 0.5 * limit) <= 1e-9 * limit     # unit masses on a potential of ones
