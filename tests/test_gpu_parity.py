"""GPU parity: libpm.so (through its C ABI) against the oracle and the golden vectors.

Tolerances (float32 state):
  * cell keys and cell order: bit-exact;
  * np.gradient stencil on an identical potential: bit-exact;
  * CIC gather on an identical gradient: bit-exact (at most 2 ulp is the stated bar);
  * density: |gpu - ref| <= 2e-5*|ref| + 1e-6*max|ref|, and the GPU's error against a float64
    deposit must not exceed the reference's own (the reference rounds to float32 at every add,
    the GPU sums in float64 and rounds once);
  * potential on an identical density: 1e-5 * max|phi| (pocketfft C2C vs rocFFT R2C/C2R);
  * positions / velocities after K steps: 1e-5*L and 1e-5*max|v|.
"""
import numpy as np
import pytest

from oracle import pm_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pm():
    import pm as _pm
    _pm.load()
    return _pm


def ulp_diff(a, b):
    """Distance in float32 ulps (same-sign finite values)."""
    ia = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    ib = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


def assert_density(d_gpu, d_ref, d64):
    tol = 2e-5 * np.abs(d_ref) + 1e-6 * np.abs(d_ref).max()
    assert np.all(np.abs(d_gpu.astype(np.float64) - d_ref) <= tol)
    err_gpu = np.abs(d_gpu - d64).max()
    err_ref = np.abs(d_ref - d64).max()
    assert err_gpu <= err_ref + np.finfo(np.float32).eps * np.abs(d64).max()


@pytest.mark.parametrize("tag", ["64", "32"])
def test_edge_cases(pm, golden, tag):
    g = golden("edge_cases_g8.npz")
    G, L = int(g["G"]), int(g["L"])
    pos, mass = g["pos" + tag], g["mass" + tag]
    if tag == "32":
        assert pos.dtype == np.float32
    with pm.Simulation(G, L, pos, None, mass) as sim:
        keys, perm = sim.cell_keys()
        c = g["cell" + tag].astype(np.int64)
        ref_keys = (c[:, 0] * G + c[:, 1]) * G + c[:, 2]
        order = O.sort_order(ref_keys)
        assert np.array_equal(perm, order) and np.array_equal(keys, ref_keys[order])
        sim.deposit()
        assert_density(sim.density(), g["density" + tag], O.CIC_f64(pos, G, L, mass))
        for w, a in zip((2, 3, 4), (g["gx"], g["gy"], g["gz"])):
            sim.set_mesh(w, a)
        sim.compute_forces()
        assert np.array_equal(sim.forces(), g["forces" + tag])


def test_wrap_returns_box_size_like_numpy(pm, golden):
    g = golden("edge_cases_g8.npz")
    L = int(g["L"])
    x = g["wrap_in"].astype(np.float32)
    n = len(x) - len(x) % 3
    x3 = x[:n].reshape(-1, 3).copy()
    with pm.Simulation(8, L, x3, np.zeros_like(x3), None) as sim:
        for w in (2, 3, 4):
            sim.set_mesh(w, np.zeros((8, 8, 8), np.float32))
        sim.gather_kick_drift(0.005)
        assert np.array_equal(sim.positions().ravel(), g["wrap_out"][:n])


@pytest.fixture(params=["as shipped", "order from the order before, sort without look-back"])
def binning_ways(request, monkeypatch):
    """The golden vectors are small sets, which are binned by the look-back sort of everything.  The second way sends
    them through the code the large sets take: PM_REBIN_MIN (read by pm_create) lets a set of 256 particles put its cell
    order together from the order before, PM_SORT_CHAIN_TILES = 0 sends every sort through the counting kernel and scan."""
    if request.param != "as shipped":
        monkeypatch.setenv("PM_REBIN_MIN", "256")
        monkeypatch.setenv("PM_SORT_CHAIN_TILES", "0")
    return request.param


@pytest.mark.parametrize("name", ["uniform_n10000_g32.npz", "c1mini_g50.npz"])
def test_step0_stages(pm, golden, name, binning_ways):
    from pm import functions as F
    g = golden(name)
    G, L = int(g["G"]), int(g["L"])
    pos, mass = g["pos0"], g["mass"]
    with pm.Simulation(G, L, pos, g["vel0"], mass) as sim:
        keys, perm = sim.cell_keys()
        ref_keys = O.cell_keys(pos, G, L)
        order = O.sort_order(ref_keys)
        assert np.array_equal(perm, order) and np.array_equal(keys, ref_keys[order])
        sim.deposit()
        assert_density(sim.density(), g["s0_density"], O.CIC_f64(pos, G, L, mass))
    # each stage on the reference's own input, through the reference-named functions
    assert_density(F.CIC(pos, G, L, mass), g["s0_density"], O.CIC_f64(pos, G, L, mass))
    pot = F.compute_potential(g["s0_density"], G)
    assert pot.dtype == np.float32
    assert np.abs(pot - g["s0_potential"]).max() <= 1e-5 * np.abs(g["s0_potential"]).max()
    gx, gy, gz = F.gradient(g["s0_potential"])
    assert np.array_equal(gx, g["s0_gx"]) and np.array_equal(gy, g["s0_gy"]) and np.array_equal(gz, g["s0_gz"])
    f = F.force_CIC((g["s0_gx"], g["s0_gy"], g["s0_gz"]), pos, G, L)
    assert f.dtype == np.float32 and f.shape == (len(pos), 3)
    assert ulp_diff(f, g["s0_forces"]).max() <= 2
    assert np.array_equal(f, g["s0_forces"])
    x1, v1 = F.kick_drift((g["s0_gx"], g["s0_gy"], g["s0_gz"]), pos, g["vel0"], G, L, float(g["dt"]))
    assert x1.dtype == np.float32 and v1.dtype == g["vel0"].dtype
    assert np.array_equal(x1, g["x1"]) and np.array_equal(v1, g["v1"])
    F.release()


@pytest.mark.parametrize("name", ["uniform_n10000_g32.npz", "c1mini_g50.npz"])
@pytest.mark.parametrize("fused", [False, True])
def test_k_steps(pm, golden, name, fused, binning_ways):
    g = golden(name)
    G, L, dt, K = int(g["G"]), int(g["L"]), float(g["dt"]), int(g["K"])
    with pm.Simulation(G, L, g["pos0"], g["vel0"], g["mass"], fused_gradient=fused) as sim:
        for step in range(1, K + 1):
            sim.step(dt)
            if f"x{step}" in g:
                x, v = sim.positions(), sim.velocities()
                assert x.dtype == np.float32 and v.dtype == g["vel0"].dtype
                dx = np.abs(x.astype(np.float64) - g[f"x{step}"])
                dx = np.minimum(dx, L - dx)                      # periodic distance
                assert dx.max() <= 1e-5 * L, (step, dx.max())
                vmax = np.abs(g[f"v{step}"]).max()
                assert np.abs(v - g[f"v{step}"]).max() <= 1e-5 * vmax, (step, np.abs(v - g[f"v{step}"]).max(), vmax)


def test_fused_gradient_is_bit_identical(pm, golden):
    g = golden("uniform_n10000_g32.npz")
    G, L, dt = int(g["G"]), int(g["L"]), float(g["dt"])
    out = []
    for fused in (False, True):
        with pm.Simulation(G, L, g["pos0"], g["vel0"], g["mass"], fused_gradient=fused) as sim:
            sim.step(dt, 5)
            out.append((sim.positions(), sim.velocities()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_fused_step_512(pm):
    """G = 512: the fused step (deposit emits z-transformed rows, gradient inside the gather) against
    the stage-by-stage step, clustered input with heavy cells and heavy rows."""
    n, G, L, dt = 400_000, 512, 10000, 0.005
    rng = np.random.default_rng(9)
    pos = (rng.random((n, 3)) * L).astype(np.float32)
    pos[: n // 2] = (5000 + rng.normal(0, 40, (n // 2, 3))).astype(np.float32)
    vel = rng.normal(0, 2000, (n, 3)).astype(np.float32)
    mass = (rng.random(n) + 0.5).astype(np.float32) * 100
    out = []
    for fused in (False, True):
        with pm.Simulation(G, L, pos, vel, mass, fused_gradient=fused) as sim:
            sim.step(dt, 3)
            out.append((sim.positions(), sim.velocities(), sim.energies(), sim.potential()))
    dx = np.abs(out[0][0].astype(np.float64) - out[1][0])
    assert np.minimum(dx, L - dx).max() <= 1e-6 * L
    assert np.abs(out[0][1] - out[1][1]).max() <= 1e-6 * np.abs(out[0][1]).max()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])     # the same bits, in fact
    # The fused step transforms only the potential rows that its gather reads (most of this mesh is empty); asked for, the
    # energies and the mesh are those of the whole potential all the same.
    assert out[0][2] == out[1][2] and np.array_equal(out[0][3], out[1][3])
    # The main loop in one call samples the energies after steps 0 and 2: those steps transform every row at once, step 1 only
    # the rows its gather reads.  Step by step with the energies asked for afterwards (the rest of the rows on demand) gives
    # the same numbers and the same particles.
    with pm.Simulation(G, L, pos, vel, mass, fused_gradient=True) as sim:
        traj, energies = sim.run(3, dt, trajectory=True, energy_every=2)
        assert np.array_equal(traj[2], out[1][0]) and np.array_equal(sim.velocities(), out[1][1])
    with pm.Simulation(G, L, pos, vel, mass, fused_gradient=True) as sim:
        by_hand = []
        for s in range(3):
            sim.step(dt)
            if s % 2 == 0:
                by_hand.append(sim.energies())
    assert [tuple(e[:2]) for e in energies] == by_hand and by_hand[1] == out[1][2]


def test_cell_order_over_steps(pm, golden, binning_ways):
    """Cell order = stable sort by key of the storage order, which is the previous cell order."""
    g = golden("uniform_n10000_g32.npz")
    G, L, dt = int(g["G"]), int(g["L"]), float(g["dt"])
    with pm.Simulation(G, L, g["pos0"], g["vel0"], g["mass"]) as sim:
        storage = np.arange(len(g["pos0"]))             # original index of each storage slot
        for step in range(6):
            x = sim.positions()                         # original order
            keys, perm = sim.cell_keys()
            k_storage = O.cell_keys(x, G, L)[storage]
            order = O.sort_order(k_storage)
            storage = storage[order]
            assert np.array_equal(perm, storage), step
            assert np.array_equal(keys, k_storage[order]), step
            sim.step(dt)
            sim.synchronize()                           # (the drift's estimate of its movers has reached the host)


def test_results_do_not_depend_on_input_order(pm, golden):
    """Density is summed in cell order, forces are per particle: a shuffled input gives the same
    trajectories up to the float64 summation order inside a cell."""
    g = golden("uniform_n10000_g32.npz")
    G, L, dt = int(g["G"]), int(g["L"]), float(g["dt"])
    rng = np.random.default_rng(3)
    p = rng.permutation(len(g["pos0"]))
    with pm.Simulation(G, L, g["pos0"], g["vel0"], g["mass"]) as a, \
            pm.Simulation(G, L, g["pos0"][p], g["vel0"][p], g["mass"][p]) as b:
        a.step(dt, 3)
        b.step(dt, 3)
        assert np.abs(a.positions()[p] - b.positions()).max() <= 1e-6 * L


def _uniform(n, L, seed=0):
    rng = np.random.default_rng(seed)
    pos = (rng.random((n, 3)) * L).astype(np.float32)
    vel = rng.normal(0, 10, (n, 3)).astype(np.float32)
    return pos, vel, np.ones(n, np.float32)


def test_full_size_properties(pm):
    """N = 1M on 128^3 (the correctness configuration): size-independent properties."""
    n, G, L, dt = 1_000_000, 128, 10000, 0.005
    pos, vel, mass = _uniform(n, L)
    with pm.Simulation(G, L, pos, vel, mass) as sim:
        keys, perm = sim.cell_keys()
        assert np.all(np.diff(keys) >= 0) and np.array_equal(np.sort(perm), np.arange(n))
        ref_keys = O.cell_keys(pos, G, L)
        assert np.array_equal(keys, ref_keys[perm])
        same = np.diff(keys) == 0
        assert np.all(np.diff(perm)[same] > 0)                  # ties keep the original order
        sim.deposit()
        d = sim.density()
        assert abs(d.astype(np.float64).sum() - n) <= 1e-6 * n  # sum(density) == sum(mass), no /cell volume
        assert_density(d, O.CIC(pos, G, L, mass), O.CIC_f64(pos, G, L, mass))
        # gather of a constant gradient returns the constant (the eight weights sum to 1)
        for w, c in zip((2, 3, 4), (1.5, -2.0, 0.25)):
            sim.set_mesh(w, np.full((G, G, G), c, np.float32))
        sim.compute_forces()
        f = sim.forces()
        assert np.abs(f - np.array([1.5, -2.0, 0.25], np.float32)).max() <= 1e-6
        # one full step against the oracle
        sim.set_particles(pos, vel, mass)
        sim.step(dt)
        xo, vo = O.pm_step(pos, vel, mass, G, L, dt)
        dx = np.abs(sim.positions().astype(np.float64) - xo)
        assert np.minimum(dx, L - dx).max() <= 1e-5 * L
        assert np.abs(sim.velocities() - vo).max() <= 1e-5 * np.abs(vo).max()


def test_correctness_configuration_100_steps(pm):
    """Uniform random 1M particles on 128^3, 100 steps of main.py:50-60: the GPU trajectory against the oracle's,
    1e-5 L on positions and 1e-5 max|v| on velocities at the end (and at step 10, where nothing has had time to grow)."""
    n, G, L, dt = 1_000_000, 128, 10000, 0.005
    pos, vel, mass = _uniform(n, L)
    xo, vo = pos, vel
    with pm.Simulation(G, L, pos, vel, mass, fused_gradient=True) as sim:
        for k in range(1, 101):
            xo, vo = O.pm_step(xo, vo, mass, G, L, dt)
            if k in (10, 100):
                sim.step(dt, 10 if k == 10 else 90)
                dx = np.abs(sim.positions().astype(np.float64) - xo)
                assert np.minimum(dx, L - dx).max() <= 1e-5 * L, k
                assert np.abs(sim.velocities() - vo).max() <= 1e-5 * np.abs(vo).max(), k


def test_headline_size_properties(pm):
    """The headline configuration itself (two-galaxy collision, N = 16,777,216 on 512^3) through properties that do
    not need the oracle at that size: mass conservation of the deposit, sortedness and completeness of the cell order,
    the same bits from the fused and the staged pipeline and from two runs, no net force on the system (Newton's third
    law holds for CIC deposit + CIC gather of a centred-difference gradient up to the one-sided faces, which the
    galaxies do not touch), and the oracle on a 200k-particle sample of the same positions for the gathered forces."""
    import bench
    n, G, L, dt = 16_777_216, 512, 10000, 0.005
    pos, vel, mass = bench.make_workload("two_galaxy", n, L, 1052)
    with pm.Simulation(G, L, pos, vel, mass, fused_gradient=False) as sim:
        keys, perm = sim.cell_keys()
        assert np.all(np.diff(keys) >= 0)
        assert np.array_equal(np.bincount(perm, minlength=n), np.ones(n, np.int64))
        sample = np.random.default_rng(0).choice(n, 200_000, replace=False)
        assert np.array_equal(keys[np.argsort(perm)[sample]], O.cell_keys(pos[sample], G, L))
        sim.deposit()
        d = sim.density()
        assert abs(d.astype(np.float64).sum() - n) <= 1e-6 * n
        sim.solve_potential()
        sim.gradient()
        sim.compute_forces()
        f = sim.forces()
        grad = [sim.mesh(w) for w in (2, 3, 4)]
        fo = O.force_CIC(grad, pos[sample], G, L)
        assert np.array_equal(f[sample], fo)                            # same gradient meshes: bit for bit
        net = f.astype(np.float64).sum(axis=0)
        assert np.abs(net).max() <= 1e-5 * np.abs(f.astype(np.float64)).sum(axis=0).max()
        sim.set_particles(pos, vel, mass)
        sim.step(dt, 2)
        x_staged, v_staged = sim.positions(), sim.velocities()
    for _ in range(2):
        with pm.Simulation(G, L, pos, vel, mass, fused_gradient=True) as sim:
            sim.step(dt, 2)
            assert np.array_equal(sim.positions(), x_staged) and np.array_equal(sim.velocities(), v_staged)
    # and the oracle itself at this size (about 5 s of CPU per step): the whole cell order bit for bit, the density
    # within the deposit's bounds, two full steps within the K-step tolerances
    ko = O.cell_keys(pos, G, L)
    order = O.sort_order(ko)
    assert np.array_equal(keys, ko[order]) and np.array_equal(perm, order)
    assert_density(d, O.CIC(pos, G, L, mass), O.CIC_f64(pos, G, L, mass))
    xo, vo = pos, vel
    for _ in range(2):
        xo, vo = O.pm_step(xo, vo, mass, G, L, dt)
    dx = np.abs(x_staged.astype(np.float64) - xo)
    assert np.minimum(dx, L - dx).max() <= 1e-5 * L
    assert np.abs(v_staged - vo).max() <= 1e-5 * np.abs(vo).max()


def test_headline_size_uniform_set_against_oracle(pm):
    """The uniform particle set of the headline size (N = 16,777,216 on 512^3; 0.8 % of its particles change their cell
    per step, so its binnings start from the order before): after every one of three steps the whole cell order -- keys
    and permutation -- bit for bit against the oracle's keys and stable sort of the storage order, and the trajectory
    within 1e-5 L and 1e-5 max|v| of the oracle's."""
    import bench
    n, G, L, dt = 16_777_216, 512, 10000, 0.005
    pos, vel, mass = bench.make_workload("uniform", n, L, 1052)
    xo, vo = pos, vel
    storage = np.arange(n)
    took = []
    with pm.Simulation(G, L, pos, vel, mass, fused_gradient=True) as sim:
        for step in range(4):
            keys, perm = sim.cell_keys()                # (bins the particles as they are now)
            took.append(sim.last_binning()[0])
            k_storage = O.cell_keys(xo, G, L)[storage]
            order = O.sort_order(k_storage)
            storage = storage[order]
            assert np.array_equal(perm, storage), step
            assert np.array_equal(keys, k_storage[order]), step
            if step == 3:
                break
            sim.step(dt)
            sim.synchronize()
            xo_next, vo = O.pm_step(xo, vo, mass, G, L, dt)
            x = sim.positions()
            dx = np.abs(x.astype(np.float64) - xo_next)
            assert np.minimum(dx, L - dx).max() <= 1e-5 * L, step
            assert np.abs(sim.velocities() - vo).max() <= 1e-5 * np.abs(vo).max(), step
            # the next comparison of cell orders is about the binning, not about the last bits of the solve: the oracle
            # goes on from the engine's own positions (float32 both, so the keys are computed from the same numbers)
            xo, vo = x, sim.velocities()
    assert took == [False, True, True, True], took


def test_single_fourier_mode(pm):
    """A single density mode delta gives delta / k^2 with k in cycles per sample (functions.py:75-85)."""
    from pm import functions as F
    G = 64
    i = np.arange(G)
    for ax, m in ((0, 3), (1, 5), (2, 1)):
        shape = [1, 1, 1]
        shape[ax] = G
        dens = np.broadcast_to(np.cos(2 * np.pi * m * i / G).reshape(shape), (G, G, G)).astype(np.float32)
        pot = F.compute_potential(dens, G)
        want = dens / np.float32((m / G) ** 2)
        assert np.abs(pot - want).max() <= 2e-5 * np.abs(want).max()
    F.release()


def test_poisson_512_column_fft(pm):
    """G = 512 takes the hand-written column-FFT path: against the oracle (scipy.fft, full complex
    transform), against the rocFFT-only path, and on single Fourier modes."""
    G = 512
    rng = np.random.default_rng(2)
    dens = rng.random((G, G, G), dtype=np.float32)
    dens[100:110, 200:210, 300:310] += 50.0
    pots = []
    for rocfft_only in (False, True):
        with pm.Simulation(G, 1.0, capacity=1, rocfft_only=rocfft_only) as sim:
            sim.set_mesh(0, dens)
            sim.solve_potential()
            pots.append(sim.potential())
    ref = O.compute_potential(dens, G)
    scale = np.abs(ref).max()
    assert np.abs(pots[0] - ref).max() <= 1e-5 * scale
    assert np.abs(pots[1] - ref).max() <= 1e-5 * scale
    assert np.abs(pots[0] - pots[1]).max() <= 1e-5 * scale
    i = np.arange(G)
    with pm.Simulation(G, 1.0, capacity=1) as sim:
        for ax, m in ((0, 3), (1, 200), (2, 256), (0, 511)):
            shape = [1, 1, 1]
            shape[ax] = G
            mode = np.broadcast_to(np.cos(2 * np.pi * m * i / G).reshape(shape), (G, G, G)).astype(np.float32)
            sim.set_mesh(0, mode)
            sim.solve_potential()
            f = min(m, G - m) / G
            want = mode / np.float32(f * f)
            assert np.abs(sim.potential() - want).max() <= 2e-5 * np.abs(want).max(), (ax, m)


@pytest.mark.parametrize("G", [128, 256])
def test_poisson_small_hand_written_transforms(pm, G):
    """G = 128 and 256 (column passes 2 x 8 x 8 and 4 x 8 x 8; 64- and 128-point row transforms): the solve against
    the oracle (scipy's transforms, the reference's own arithmetic) and against the rocFFT-only path; and a fused step
    (deposit emits z-transformed rows) with the same bits as the staged one."""
    rng = np.random.default_rng(G)
    dens = rng.random((G, G, G), dtype=np.float32) ** 6
    dens[10:14, G - 9:G - 5, 30:34] += 40.0
    ref = O.compute_potential(dens, G)
    pots = []
    for rocfft_only in (False, True):
        with pm.Simulation(G, 1.0, capacity=1, rocfft_only=rocfft_only) as sim:
            sim.set_mesh(0, dens)
            sim.solve_potential()
            pots.append(sim.potential())
    scale = np.abs(ref).max()
    assert np.abs(pots[0] - ref).max() <= 1e-5 * scale and np.abs(pots[1] - ref).max() <= 1e-5 * scale
    n, L, dt = 300_000, 10000, 0.005
    pos = (rng.random((n, 3)) * L).astype(np.float32)
    pos[: n // 2] = (5000 + rng.normal(0, 150, (n // 2, 3))).astype(np.float32)
    vel = rng.normal(0, 2000, (n, 3)).astype(np.float32)
    mass = (rng.random(n) + 0.5).astype(np.float32) * 100
    out = []
    for fused in (False, True):
        with pm.Simulation(G, L, pos, vel, mass, fused_gradient=fused) as sim:
            sim.step(dt, 3)
            out.append((sim.positions(), sim.velocities()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    xo, vo = pos, vel
    for _ in range(3):
        xo, vo = O.pm_step(xo, vo, mass, G, L, dt)
    dx = np.abs(out[1][0].astype(np.float64) - xo)
    assert np.minimum(dx, L - dx).max() <= 1e-5 * L
    assert np.abs(out[1][1] - vo).max() <= 1e-5 * np.abs(vo).max()


def test_poisson_1024_column_fft(pm):
    """G = 1024 (radix 16, 8, 8 columns; 512-point rows): against the rocFFT-only path, which the small-mesh tests pin
    to the oracle, and on single Fourier modes with their analytic answer."""
    G = 1024
    rng = np.random.default_rng(4)
    dens = rng.random((G, G, G), dtype=np.float32)
    dens[100:110, 900:910, 300:310] += 50.0
    pots = []
    for rocfft_only in (False, True):
        with pm.Simulation(G, 1.0, capacity=1, rocfft_only=rocfft_only) as sim:
            sim.set_mesh(0, dens)
            sim.solve_potential()
            pots.append(sim.potential())
    scale = np.abs(pots[1]).max()
    assert np.abs(pots[0] - pots[1]).max() <= 1e-5 * scale
    del pots, dens
    i = np.arange(G)
    with pm.Simulation(G, 1.0, capacity=1) as sim:
        for ax, m in ((0, 3), (1, 700), (2, 512), (0, 1023), (2, 37)):
            shape = [1, 1, 1]
            shape[ax] = G
            mode = np.broadcast_to(np.cos(2 * np.pi * m * i / G).reshape(shape), (G, G, G)).astype(np.float32)
            sim.set_mesh(0, mode)
            sim.solve_potential()
            f = min(m, G - m) / G
            want = mode / np.float32(f * f)
            assert np.abs(sim.potential() - want).max() <= 2e-5 * np.abs(want).max(), (ax, m)


def test_fused_step_1024(pm):
    """G = 1024: the fused step against the stage-by-stage step and against the rocFFT-only pipeline."""
    n, G, L, dt = 400_000, 1024, 10000, 0.005
    rng = np.random.default_rng(9)
    pos = (rng.random((n, 3)) * L).astype(np.float32)
    pos[: n // 2] = (5000 + rng.normal(0, 40, (n // 2, 3))).astype(np.float32)
    vel = rng.normal(0, 2000, (n, 3)).astype(np.float32)
    mass = (rng.random(n) + 0.5).astype(np.float32) * 100
    out = []
    for fused, rocfft_only in ((False, False), (True, False), (False, True)):
        with pm.Simulation(G, L, pos, vel, mass, fused_gradient=fused, rocfft_only=rocfft_only) as sim:
            sim.step(dt, 2)
            out.append((sim.positions(), sim.velocities()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    # two float32 implementations of a 1024^3 transform agree to ~1e-6 of max|phi|; the difference quotient of the
    # gradient and the dense clump's large forces carry that into the velocities at a few 1e-5 of max|v|
    dx = np.abs(out[0][0].astype(np.float64) - out[2][0])
    assert np.minimum(dx, L - dx).max() <= 1e-5 * L
    assert np.abs(out[0][1] - out[2][1]).max() <= 5e-5 * np.abs(out[2][1]).max()


# ---- the clustered-disk configuration itself: N = 67,108,864 on 1024^3 -------------------------------------

@pytest.fixture(scope="module")
def disk64m(pm):
    """The exponential disk of 64M particles (drawn on the device, as bench.py --device-ic draws it) as host arrays."""
    n, G, L = 67_108_864, 1024, 10000
    with pm.Simulation(G, L, capacity=n) as sim:
        sim.generate("disk", n, seed=42)
        pos, vel = sim.positions(), sim.velocities()
    return n, G, L, pos, vel, np.ones(n, np.float32)


def test_disk_1024_cell_order_and_density(pm, disk64m):
    """Whole cell order bit for bit against the oracle's keys and stable sort, mass conservation, density within the
    deposit's bounds against the oracle's float32 and float64 deposits."""
    n, G, L, pos, vel, mass = disk64m
    with pm.Simulation(G, L, pos, vel, mass) as sim:
        keys, perm = sim.cell_keys()
        ko = O.cell_keys(pos, G, L)
        order = O.sort_order(ko)
        assert np.array_equal(perm, order) and np.array_equal(keys, ko[order])
        # at least 30 % of the particles inside 0.1 % of the cells: the contention this configuration is about
        edges = np.flatnonzero(np.diff(keys)) + 1
        counts = np.sort(np.diff(np.concatenate(([0], edges, [n]))))[::-1]
        assert counts[: G ** 3 // 1000].sum() >= 0.3 * n
        del keys, perm, order, ko, edges, counts
        sim.deposit()
        d = sim.density()
    assert abs(d.astype(np.float64).sum() - n) <= 1e-6 * n
    d64 = O.CIC_f64(pos, G, L, mass)
    d32 = O.CIC(pos, G, L, mass)
    tol = 2e-5 * np.abs(d32) + 1e-6 * np.abs(d32).max()
    assert np.all(np.abs(d - d32) <= tol)
    assert np.abs(d - d64).max() <= np.abs(d32 - d64).max() + np.finfo(np.float32).eps * np.abs(d64).max()


def test_disk_1024_forces_and_fused_step(pm, disk64m):
    """force_CIC bit for bit against the oracle on a 200k-particle sample (same gradient meshes); the potential against
    the density on single Fourier modes (phi_k = rho_k / k^2, functions.py:85); no net force; and one whole step with the
    same bits from the fused and the staged pipeline."""
    n, G, L, pos, vel, mass = disk64m
    dt = 0.005
    with pm.Simulation(G, L, pos, vel, mass, fused_gradient=False) as sim:
        sim.deposit()
        sim.solve_potential()
        # single modes by projection, sum(f * exp(-2 pi i (a x + b y + c z) / G)): along z for all modes at once as a
        # matrix product (float64, a group of planes at a time), then the two small remaining axes
        modes = ((1, 0, 0), (0, 3, 2), (5, 7, 1))
        x = np.arange(G)
        wz = np.concatenate([[np.cos(2 * np.pi * c * x / G), -np.sin(2 * np.pi * c * x / G)] for _, _, c in modes]).T

        def project(f):
            t = np.concatenate([f[i:i + 64].reshape(-1, G).astype(np.float64) @ wz for i in range(0, G, 64)]).reshape(G, G, len(modes), 2)
            t = t[..., 0] + 1j * t[..., 1]
            return [np.exp(-2j * np.pi * a * x / G) @ (t[:, :, q].T * np.exp(-2j * np.pi * b * x / G)[:, None]).sum(axis=0)
                    for q, (a, b, _) in enumerate(modes)]
        rho, phi = project(sim.density()), project(sim.potential())
        for (a, b, c), rho_k, phi_k in zip(modes, rho, phi):
            k2 = sum((min(q, G - q) / G) ** 2 for q in (a, b, c))
            assert abs(phi_k - rho_k / k2) <= 2e-5 * abs(rho_k / k2), (a, b, c)
        sim.gradient()
        sim.compute_forces()
        f = sim.forces()
        sample = np.random.default_rng(0).choice(n, 200_000, replace=False)
        fo = O.force_CIC([sim.mesh(w) for w in (2, 3, 4)], pos[sample], G, L)
        assert np.array_equal(f[sample], fo)
        net = f.astype(np.float64).sum(axis=0)
        assert np.abs(net).max() <= 1e-5 * np.abs(f.astype(np.float64)).sum(axis=0).max()
        del f, fo
        sim.set_particles(pos, vel, mass)
        sim.step(dt)
        x_staged, v_staged = sim.positions(), sim.velocities()
    with pm.Simulation(G, L, pos, vel, mass, fused_gradient=True) as sim:
        sim.step(dt)
        assert np.array_equal(sim.positions(), x_staged) and np.array_equal(sim.velocities(), v_staged)
    # the particles moved as main.py:58-60 says they do: every one by its velocity times dt (float32), inside the box
    dx = np.abs(x_staged.astype(np.float64) - (pos.astype(np.float64) + v_staged.astype(np.float64) * dt) % L)
    assert np.minimum(dx, L - dx).max() <= 1e-6 * L and x_staged.min() >= 0 and x_staged.max() <= L


def test_clustered_hot_cell(pm):
    """Half of the particles in one cell: the wave-cooperative run path of the deposit."""
    n, G, L = 200_000, 100, 10000
    rng = np.random.default_rng(5)
    pos = (rng.random((n, 3)) * L).astype(np.float32)
    pos[: n // 2] = (5000 + rng.random((n // 2, 3)) * 90).astype(np.float32)
    mass = (rng.random(n) + 0.5).astype(np.float32)
    with pm.Simulation(G, L, pos, None, mass) as sim:
        sim.deposit()
        assert_density(sim.density(), O.CIC(pos, G, L, mass), O.CIC_f64(pos, G, L, mass))


@pytest.mark.parametrize("G,n", [(32, 150_000), (16, 200_000), (8, 300_000), (6, 70_001)])
def test_deposit_walks(pm, G, n):
    """The three walks of the deposit -- one particle per lane (few particles per row), four per lane (hundreds to
    thousands), rows cut into units (more) -- on meshes small enough that every row, including the wrapped ones at
    i = 0 and j = 0, takes the walk in question; uneven masses, runs that start at any alignment."""
    L = 1000
    rng = np.random.default_rng(G)
    pos = (rng.random((n, 3)) * L).astype(np.float32)
    pos[: n // 3, 2] = (rng.random(n // 3) * (L / G) + 3 * (L / G)).astype(np.float32)     # a third of them in the cells k = 3
    mass = (rng.random(n) * 2 + 0.01).astype(np.float32)
    with pm.Simulation(G, L, pos, None, mass) as sim:
        sim.deposit()
        assert_density(sim.density(), O.CIC(pos, G, L, mass), O.CIC_f64(pos, G, L, mass))


@pytest.mark.parametrize("units", ["0", "1"])
def test_deposit_with_and_without_units(pm, monkeypatch, units):
    """A deposit plans units for its crowded rows only while recent deposits counted some; otherwise the row kernel walks
    every row itself and is the deposit's only kernel beside the plan.  Either way is right for any set: a set with a dense
    clump and the uniform golden set, each with the units forced off and forced on; and the switch itself -- a context
    that starts with the units on, finds none on a uniform set, and goes on without them."""
    monkeypatch.setenv("PM_DEPOSIT_UNITS", units)
    n, G, L = 300_000, 64, 10000
    rng = np.random.default_rng(12)
    pos = (rng.random((n, 3)) * L).astype(np.float32)
    pos[: n // 3] = (2500 + rng.normal(0, 60, (n // 3, 3))).astype(np.float32)
    mass = (rng.random(n) + 0.5).astype(np.float32)
    with pm.Simulation(G, L, pos, None, mass) as sim:
        for _ in range(3):                              # (the counters of a deposit are zeroed by the one before)
            sim.deposit()
            assert_density(sim.density(), O.CIC(pos, G, L, mass), O.CIC_f64(pos, G, L, mass))
    monkeypatch.delenv("PM_DEPOSIT_UNITS")
    upos = (rng.random((n, 3)) * L).astype(np.float32)
    with pm.Simulation(G, L, upos, None, mass) as sim:
        for _ in range(4):                              # on from the start, off once the count of none has come back
            sim.deposit()
            sim.synchronize()
            assert_density(sim.density(), O.CIC(upos, G, L, mass), O.CIC_f64(upos, G, L, mass))
        sim.set_particles(pos, None, mass)              # a new set: on again
        for _ in range(3):
            sim.deposit()
            sim.synchronize()
            assert_density(sim.density(), O.CIC(pos, G, L, mass), O.CIC_f64(pos, G, L, mass))


def test_steps_repeat_bit_for_bit_while_crowded_rows_appear(pm):
    """A set that starts spread out (no crowded row: after two deposits the units are off) and falls into a clump within a
    few steps (crowded rows: the row kernel walks them itself, then the units come on).  Which deposit goes which way is
    decided by the count of the deposit before the last, never by how far the host happens to be ahead of the device: two
    runs give the same bits at every step, hurried or not."""
    n, G, L, dt = 300_000, 64, 10000.0, 0.005
    rng = np.random.default_rng(21)
    pos = (rng.random((n, 3)) * L).astype(np.float32)
    target = (5000 + rng.normal(0, 30, (n, 3))).astype(np.float32)
    vel = ((target - pos) / (8 * dt)).astype(np.float32)               # there after eight steps: crowded rows from the seventh on
    mass = np.full(n, 1e-9, np.float32)                                 # (no force worth the name: they arrive)
    runs = []
    for hurried in (True, False):
        with pm.Simulation(G, L, pos, vel, mass, fused_gradient=True) as sim:
            states = []
            for s in range(10):
                sim.step(dt)
                if not hurried:
                    sim.synchronize()
                if hurried and s < 9:
                    continue
                states.append(sim.positions())
            sim.deposit()
            runs.append((states, sim.density()))
    assert np.array_equal(runs[0][0][-1], runs[1][0][-1]) and np.array_equal(runs[0][1], runs[1][1])
    x = runs[1][0][-1].astype(np.float64)
    assert np.abs(x - 5000).max() < 2000 and len(runs[1][0]) == 10     # the clump formed; every step of the slow run was read
    assert_density(runs[1][1], O.CIC(runs[1][0][-1], G, int(L), mass), O.CIC_f64(runs[1][0][-1], G, int(L), mass))


def test_deposit_is_bitwise_reproducible(pm):
    """Same input, fresh engines, clustered and sparse rows alike: identical bits every time."""
    n, G, L = 300_000, 64, 10000
    rng = np.random.default_rng(11)
    pos = (rng.random((n, 3)) * L).astype(np.float32)
    pos[: n // 3] = (2500 + rng.normal(0, 60, (n // 3, 3))).astype(np.float32)     # a dense clump: heavy rows
    mass = (rng.random(n) + 0.5).astype(np.float32)
    out = []
    for _ in range(3):
        with pm.Simulation(G, L, pos, None, mass) as sim:
            sim.deposit()
            out.append(sim.density())
    assert np.array_equal(out[0], out[1]) and np.array_equal(out[0], out[2])
    assert_density(out[0], O.CIC(pos, G, L, mass), O.CIC_f64(pos, G, L, mass))


def test_energies(pm, golden):
    """KE / PE every 10th step as main.py:64-67: post-drift positions, pre-drift potential."""
    for name, tol in (("uniform_n10000_g32.npz", 1e-5), ("c1mini_g50.npz", 1e-5)):
        g = golden(name)
        G, L, dt, K = int(g["G"]), int(g["L"]), float(g["dt"]), int(g["K"])
        with pm.Simulation(G, L, g["pos0"], g["vel0"], g["mass"]) as sim:
            for step in range(K):
                sim.step(dt)
                if f"s{step}_KE" in g:
                    ke, pe = sim.energies()
                    assert abs(ke - g[f"s{step}_KE"]) <= tol * abs(g[f"s{step}_KE"]), (name, step)
                    assert abs(pe - g[f"s{step}_PE"]) <= tol * abs(g[f"s{step}_PE"]), (name, step)
    # the stand-alone functions on the reference's own inputs
    from pm import functions as F
    g = golden("c1mini_g50.npz")
    ke = F.compute_kinetic_energy(g["vel0"], g["mass"])
    assert abs(ke - O.compute_kinetic_energy(g["vel0"], g["mass"])) <= 1e-12 * abs(ke)      # float64 velocities
    x32 = g["x1"]
    pe = F.compute_potential_energy(x32, g["mass"], g["s0_potential"], int(g["G"]), int(g["L"]))
    assert abs(pe - O.compute_potential_energy(x32, g["mass"], g["s0_potential"], int(g["G"]), int(g["L"]))) <= 1e-5 * abs(pe)
    F.release()


@pytest.mark.parametrize("pinned", [True, False])
def test_run_is_the_main_loop(pm, golden, pinned):
    """Simulation.run (pm_run) = main.py:44-67: trajectory[s] is positions.copy() after step s, energies every
    energy_every-th step.  Bitwise equal to stepping and downloading by hand; the golden vectors pin the values."""
    for name in ("uniform_n10000_g32.npz", "c1mini_g50.npz"):
        g = golden(name)
        G, L, dt = int(g["G"]), int(g["L"]), float(g["dt"])
        K = 12
        with pm.Simulation(G, L, g["pos0"], g["vel0"], g["mass"]) as sim:
            traj, en = sim.run(K, dt, trajectory=True, energy_every=5, pinned=pinned)
            v_end = sim.velocities()
        assert traj.shape == (K, len(g["mass"]), 3) and traj.dtype == np.float32
        assert en.shape == (3, 3) and np.array_equal(en[:, 2], en[:, 0] + en[:, 1])
        with pm.Simulation(G, L, g["pos0"], g["vel0"], g["mass"]) as ref:
            for s in range(K):
                ref.step(dt)
                assert np.array_equal(traj[s], ref.positions()), (name, s)
                if s % 5 == 0:
                    assert en[s // 5, :2].tolist() == list(ref.energies()), (name, s)
            assert np.array_equal(v_end, ref.velocities())
        for s in (1, 5, 10):
            assert np.abs(traj[s - 1].astype(np.float64) - g[f"x{s}"]).max() <= 1e-5 * L, (name, s)
        assert abs(en[0, 0] - g["s0_KE"]) <= 1e-5 * abs(g["s0_KE"]) and abs(en[0, 1] - g["s0_PE"]) <= 1e-5 * abs(g["s0_PE"])
    # views keep the page-locked block alive; no trajectory, no energies, no steps
    g = golden("uniform_n10000_g32.npz")
    with pm.Simulation(int(g["G"]), int(g["L"]), g["pos0"], g["vel0"], g["mass"]) as sim:
        last = sim.run(3, float(g["dt"]), energy_every=0, pinned=pinned)[0][-1]
        none, en = sim.run(2, float(g["dt"]), trajectory=False, energy_every=1)
        assert none is None and en.shape == (2, 3)
        traj, en = sim.run(0, float(g["dt"]))
        assert traj.shape == (0, 10000, 3) and en.shape == (0, 3)
    import gc
    gc.collect()
    assert np.isfinite(last).all() and last.shape == (10000, 3)


def test_ngp_variants(pm, golden):
    from pm import functions as F
    g = golden("uniform_n10000_g32.npz")
    G, L = int(g["G"]), int(g["L"])
    d = F.NGP(g["pos0"], G, L, g["mass"])
    assert d.dtype == np.float64 and np.array_equal(d, g["ngp_density"])
    f = F.force_NGP((g["s0_gx"], g["s0_gy"], g["s0_gz"]), g["pos0"], G, L)
    assert f.dtype == np.float32 and np.array_equal(f, g["ngp_forces"])
    # a clump: long runs of one cell
    rng = np.random.default_rng(4)
    pos = (rng.random((50000, 3)) * L).astype(np.float32)
    pos[:30000] = (7000 + rng.random((30000, 3)) * 100).astype(np.float32)
    mass = (rng.random(50000) + 0.5).astype(np.float32)
    assert np.allclose(F.NGP(pos, G, L, mass), O.NGP(pos, G, L, mass), rtol=1e-12, atol=0)
    # the reference raises IndexError when the float32 wrap returns exactly box_size
    bad = np.array([[-1e-9, 5.0, 5.0]], np.float32)
    with pytest.raises(IndexError):
        O.NGP(bad, G, L, np.ones(1, np.float32))
    with pytest.raises(IndexError):
        F.NGP(bad, G, L, np.ones(1, np.float32))
    F.release()


@pytest.mark.parametrize("tag", ["", "d_"])
def test_ngp_branch_resident(pm, golden, tag):
    """Simulation(interpolation="NGP"): main.py:50-60 with CIC_interpolation = False, every mesh and the forces in
    float64.  Tolerances: density 1e-12 relative (cells of more than 64 particles are summed in another order than
    np.add.at), potential / gradient / forces 1e-12 of their largest value (rocFFT vs pocketfft in float64), float32
    positions 1e-5 L, velocities 1e-5 max|v| after 12 steps."""
    from pm import _lib as Lb
    g = golden("ngp_n10000_g32.npz")
    G, L, dt, K = int(g["G"]), int(g["L"]), float(g["dt"]), int(g["K"])
    x0, v0 = g["pos0"], g["vel0"]
    if tag:
        x0, v0 = x0.astype(np.float64), v0.astype(np.float64)
    with pm.Simulation(G, L, x0, v0, g["mass"], interpolation="NGP") as sim:
        sim.deposit_ngp()
        assert np.allclose(sim.density_f64(), g[tag + "s0_density"], rtol=1e-12, atol=0)
        sim.solve_potential_f64()
        phi = sim.mesh_f64(Lb.MESH_POTENTIAL)
        assert np.abs(phi - g[tag + "s0_potential"]).max() <= 1e-12 * np.abs(g[tag + "s0_potential"]).max()
        sim.gradient_f64()
        for w, name in zip((Lb.MESH_GX, Lb.MESH_GY, Lb.MESH_GZ), ("gx", "gy", "gz")):
            ref = g[tag + "s0_" + name]
            assert np.abs(sim.mesh_f64(w) - ref).max() <= 1e-12 * np.abs(ref).max()
        sim.compute_forces_ngp_f64()
        f = sim.forces_f64()
        assert f.dtype == np.float64 and np.abs(f - g[tag + "s0_forces"]).max() <= 1e-12 * np.abs(g[tag + "s0_forces"]).max()
        for step in range(K):
            sim.step(dt)
            if step == 0:
                ke, pe = sim.energies()
                assert abs(ke - g[tag + "s0_KE"]) <= 1e-6 * abs(g[tag + "s0_KE"])
                assert abs(pe - g[tag + "s0_PE"]) <= 1e-6 * abs(g[tag + "s0_PE"])
            if step + 1 in (1, 5, K):
                x, v = sim.positions(), sim.velocities()
                xr, vr = g[f"{tag}x{step + 1}"], g[f"{tag}v{step + 1}"]
                assert x.dtype == np.float32 and v.dtype == vr.dtype
                dx = np.abs(x.astype(np.float64) - xr)
                assert np.minimum(dx, L - dx).max() <= 1e-5 * L, (tag, step)
                assert np.abs(v - vr).max() <= 1e-5 * np.abs(vr).max(), (tag, step)
    # the trajectory / energy loop runs the same branch
    with pm.Simulation(G, L, x0, v0, g["mass"], interpolation="NGP") as sim:
        traj, en = sim.run(5, dt, energy_every=10)
        dx = np.abs(traj[4].astype(np.float64) - g[tag + "x5"])
        assert np.minimum(dx, L - dx).max() <= 1e-5 * L
        assert abs(en[0, 0] - g[tag + "s0_KE"]) <= 1e-6 * abs(g[tag + "s0_KE"])
    # and the stand-alone functions keep the reference's dtypes
    from pm import functions as F
    d = F.NGP(g["pos0"], G, L, g["mass"])
    phi = F.compute_potential(d, G)
    grad = F.gradient(phi)
    f = F.force_NGP(grad, g["pos0"], G, L)
    assert phi.dtype == np.float64 and grad[0].dtype == np.float64 and f.dtype == np.float64
    assert np.abs(f - g["s0_forces"]).max() <= 1e-12 * np.abs(g["s0_forces"]).max()
    pe = F.compute_potential_energy(g["x1"], g["mass"], phi, G, L)
    assert abs(pe - g["s0_PE"]) <= 1e-9 * abs(g["s0_PE"])
    F.release()
    with pytest.raises(ValueError):
        pm.Simulation(G, L, interpolation="TSC")
    with pm.Simulation(G, L, g["pos0"], g["vel0"], g["mass"], interpolation="NGP") as sim:
        with pytest.raises(pm.PMError):
            sim.energies()                  # no potential before the first solve
        with pytest.raises(pm.PMError):
            sim.gradient_f64()
        with pytest.raises(pm.PMError):
            sim.forces_f64()


def test_random_small_configurations(pm):
    """Forty random small set-ups against the oracle, two steps each: any grid size from 2 up (odd ones too), any box
    size, from no particle to a few thousand, positions far outside the box and exactly on cell and box edges, all in
    one cell, float32 and float64 particles in every combination, fused and staged."""
    rng = np.random.default_rng(2024)
    for case in range(40):
        G = int(rng.choice([2, 3, 4, 5, 7, 8, 9, 16, 17, 24, 31, 32, 33, 40]))
        L = int(rng.choice([1, 7, 100, 10000, 123457]))
        n = int(rng.choice([0, 1, 2, 63, 64, 65, 257, 1000, 4099]))
        kind = case % 4
        if kind == 0:
            pos = rng.random((n, 3)) * L
        elif kind == 1:
            pos = (rng.random((n, 3)) - 0.5) * 20 * L                      # up to ten boxes away, both signs
        elif kind == 2:
            pos = rng.integers(0, G + 1, (n, 3)) * (L / G)                  # on cell edges, and on the box edge itself
        else:
            pos = np.full((n, 3), 0.37 * L) + rng.random((n, 3)) * (0.5 * L / G)   # (nearly) one cell
        vel = rng.normal(0, 0.3 * L, (n, 3))
        mass = (rng.random(n) + 0.5).astype(np.float32)
        pdt, vdt = [(np.float32, np.float32), (np.float64, np.float64), (np.float32, np.float64), (np.float64, np.float32)][(case // 4) % 4]
        pos, vel = pos.astype(pdt), vel.astype(vdt)
        dt = 0.01
        fused = bool(case % 2)
        tag = (case, G, L, n, kind, pdt.__name__, vdt.__name__, fused)
        with pm.Simulation(G, L, pos, vel, mass, fused_gradient=fused) as sim:
            if n:
                keys, perm = sim.cell_keys()
                ko = O.cell_keys(pos, G, L)
                order = O.sort_order(ko)
                assert np.array_equal(keys, ko[order]) and np.array_equal(perm, order), tag
            sim.deposit()
            d = sim.density()
            if n:
                assert_density(d, O.CIC(pos, G, L, mass), O.CIC_f64(pos, G, L, mass))
            else:
                assert not d.any(), tag
            sim.set_particles(pos, vel, mass)
            sim.step(dt, 2)
            x, v = sim.positions(), sim.velocities()
        xo, vo = pos, vel
        for _ in range(2):
            xo, vo = O.pm_step(xo, vo, mass, G, L, dt)
        assert x.dtype == np.float32 and v.dtype == vdt and x.shape == (n, 3), tag
        if n:
            dx = np.abs(x.astype(np.float64) - xo)
            assert np.minimum(dx, L - dx).max() <= 1e-5 * L, tag
            assert np.abs(v - vo).max() <= 1e-5 * max(np.abs(vo).max(), 1e-30), tag


def test_random_small_configurations_ngp(pm):
    """The same for the nearest-grid-point branch (float64 meshes), against the oracle's NumPy restatement."""
    rng = np.random.default_rng(77)
    for case in range(16):
        G = int(rng.choice([2, 3, 5, 8, 16, 17, 32]))
        L = int(rng.choice([7, 100, 10000]))
        n = int(rng.choice([1, 2, 65, 257, 3000]))
        pos = rng.random((n, 3)) * L if case % 3 else (rng.random((n, 3)) - 0.5) * 6 * L
        if case % 3 == 2:
            pos = np.full((n, 3), 0.61 * L) + rng.random((n, 3)) * (0.5 * L / G)
        vel = rng.normal(0, 0.3 * L, (n, 3))
        mass = (rng.random(n) + 0.5).astype(np.float32)
        pdt, vdt = [(np.float32, np.float32), (np.float64, np.float64), (np.float32, np.float64), (np.float64, np.float32)][case % 4]
        pos, vel = pos.astype(pdt), vel.astype(vdt)
        tag = (case, G, L, n, pdt.__name__, vdt.__name__)
        try:
            xo, vo = pos, vel
            for _ in range(2):
                xo, vo = O.pm_step_ngp(xo, vo, mass, G, L, 0.01)
        except IndexError:
            continue                                    # a position wrapped to exactly L: the reference stops there
        with pm.Simulation(G, L, pos, vel, mass, interpolation="NGP") as sim:
            sim.step(0.01, 2)
            x, v = sim.positions(), sim.velocities()
        assert x.dtype == np.float32 and v.dtype == vo.dtype, tag
        dx = np.abs(x.astype(np.float64) - xo)
        assert np.minimum(dx, L - dx).max() <= 1e-5 * L, tag
        assert np.abs(v - vo).max() <= 1e-5 * max(np.abs(vo).max(), 1e-30), tag


def test_main_shaped_driver_runs_both_ways():
    """examples/run_collision.py = main.py with the reference's constants and inputs: the loop body call by call through
    pm.functions (reference names, host arrays) and the resident loop give the same energies."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for mode in ([], ["--granular"], ["--ngp"], ["--ngp", "--granular"]):
        r = subprocess.run([sys.executable, os.path.join(root, "examples", "run_collision.py"), "--steps", "12"] + mode,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        m = re.search(r"12 frames of 10000 particles, total energy (\S+) -> (\S+)", r.stdout)
        assert m, r.stdout
        out[tuple(mode)] = (float(m.group(1)), float(m.group(2)))
    for a, b in (((), ("--granular",)), (("--ngp",), ("--ngp", "--granular"))):
        for u, v in zip(out[a], out[b]):
            assert abs(u - v) <= 1e-5 * abs(v), (a, out)


def test_shipped_grid_size_against_golden(pm, golden):
    """G = 100, the grid size main.py:5 ships (not a power of two: rocFFT's plans, not the hand-written transforms),
    one step on the shipped run's dtypes (float64 positions and velocities): cell indices bit for bit, density within
    the deposit's bounds, potential 1e-5 max|phi|, the gather on the golden density's own field, x1 / v1, energies."""
    base, g = golden("c1mini_g50.npz"), golden("c1mini_g100_step1.npz")
    G, L, dt = int(g["G"]), int(base["L"]), float(base["dt"])
    pos, vel, mass = base["pos0"], base["vel0"], base["mass"]
    assert G == 100 and pos.dtype == np.float64 and vel.dtype == np.float64
    c = g["s0_cell"].astype(np.int64)
    ref_keys = (c[:, 0] * G + c[:, 1]) * G + c[:, 2]
    order = O.sort_order(ref_keys)
    for fused in (False, True):
        with pm.Simulation(G, L, pos, vel, mass, fused_gradient=fused) as sim:
            keys, perm = sim.cell_keys()
            assert np.array_equal(perm, order) and np.array_equal(keys, ref_keys[order])
            if not fused:
                sim.deposit()
                assert_density(sim.density(), g["s0_density"], O.CIC_f64(pos, G, L, mass))
                sim.set_mesh(0, g["s0_density"])                # the solve and the gather on the reference's own density
                sim.solve_potential()
                phi = sim.potential()
                assert np.abs(phi[::7, ::7, ::7] - g["s0_potential_sample"]).max() <= 1e-5 * np.abs(g["s0_potential_sample"]).max()
                sim.gradient()
                sim.compute_forces()
                f = sim.forces()
                assert np.abs(f - g["s0_forces"]).max() <= 1e-4 * np.abs(g["s0_forces"]).max()
                sim.set_particles(pos, vel, mass)
            sim.step(dt)
            x, v = sim.positions(), sim.velocities()
            assert x.dtype == np.float32 and v.dtype == np.float64
            dx = np.abs(x.astype(np.float64) - g["x1"])
            assert np.minimum(dx, L - dx).max() <= 1e-5 * L
            assert np.abs(v - g["v1"]).max() <= 1e-5 * np.abs(g["v1"]).max()
            ke, pe = sim.energies()
            assert abs(ke - float(g["s0_KE"])) <= 1e-6 * abs(float(g["s0_KE"]))
            assert abs(pe - float(g["s0_PE"])) <= 1e-4 * abs(float(g["s0_PE"]))


def test_star_import_is_the_only_change_a_main_py_needs(monkeypatch):
    """examples/main_star_import.py takes every name it uses from `from pm.functions import *`, as main.py:1 takes them
    from the reference's module, and runs main.py:20-69 statement for statement.  Its state after three steps against
    the oracle's loop on the same files."""
    import os
    import runpy
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.setenv("STEPS", "3")
    monkeypatch.setattr(sys, "argv", ["main_star_import.py"])
    ns = runpy.run_path(os.path.join(root, "examples", "main_star_import.py"), run_name="__main__")
    G, L, dt = ns["grid_size"], ns["box_size"], ns["dt"]
    assert G == 100 and len(ns["trajectory"]) == 3 and len(ns["energies"]) == 1
    x, v, m, lab = O.import_galaxy(*ns["galaxy_files"], separation=400.0, direction=(0.67, 0.33, 0), velocity=(-3000, 300, 0))
    x += 0.5 * np.asarray(L, dtype=float) - x.mean(axis=0)
    for _ in range(3):
        x, v = O.pm_step(x, v, m, G, L, dt)
    assert ns["positions"].dtype == np.float32 and ns["velocities"].dtype == np.float64
    dx = np.abs(ns["positions"].astype(np.float64) - x)
    assert np.minimum(dx, L - dx).max() <= 1e-5 * L
    assert np.abs(ns["velocities"] - v).max() <= 1e-5 * np.abs(v).max()


def test_forces_do_not_survive_a_reorder(pm, golden):
    """pm_forces on a set that is not in cell order yet, then a call that bins it (pm_get_cell_keys, pm_deposit): the
    stored forces belong to the storage slots of before the reorder, so pm_get_forces must refuse rather than hand
    them out against the wrong particles; computed again after the reorder they are the direct result."""
    g = golden("uniform_n10000_g32.npz")
    G, L = int(g["G"]), int(g["L"])
    with pm.Simulation(G, L, g["pos0"], g["vel0"], g["mass"]) as sim:
        for w, a in zip((2, 3, 4), (g["s0_gx"], g["s0_gy"], g["s0_gz"])):
            sim.set_mesh(w, a)
        sim.compute_forces()
        assert np.array_equal(sim.forces(), g["s0_forces"])
        sim.cell_keys()                                     # bins: the storage order changes
        with pytest.raises(pm.PMError):
            sim.forces()
        sim.compute_forces()
        assert np.array_equal(sim.forces(), g["s0_forces"])
        sim.set_particles(g["pos0"], g["vel0"], g["mass"])
        sim.compute_forces()
        sim.deposit()
        with pytest.raises(pm.PMError):
            sim.forces()


def test_errors(pm):
    lib = pm.load()
    assert not lib.pm_create(1, 1.0, 10, 0, 0)
    assert b"grid_size" in lib.pm_last_error(None)
    assert not lib.pm_create(8, -1.0, 10, 0, 0)
    with pm.Simulation(8, 100.0, capacity=4) as sim:
        with pytest.raises(pm.PMError):
            sim.deposit()                                   # no particles yet
        with pytest.raises(pm.PMError):
            sim.set_particles(np.zeros((5, 3), np.float32)) # beyond capacity
        sim.set_particles(np.zeros((0, 3), np.float32))     # empty set is fine
        sim.deposit()
        assert not sim.density().any()
        with pytest.raises(pm.PMError):
            sim.forces()                                    # pm_forces not called


def test_meshes_a_stage_has_not_written_cannot_be_read(pm):
    """A fused step keeps neither the density nor the gradient meshes: reading them, or running a stage on them, is an
    error, not stale data; pm_set_mesh and the granular stages make them readable again."""
    rng = np.random.default_rng(4)
    pos = (rng.random((5000, 3)) * 100).astype(np.float32)
    with pm.Simulation(128, 100.0, pos, fused_gradient=True) as sim:      # hand-written transforms: no density mesh either
        for stage in (sim.density, sim.potential, sim.potential_gradient, sim.solve_potential, sim.gradient, sim.compute_forces):
            with pytest.raises(pm.PMError):
                stage()
        sim.step(0.01)
        assert np.isfinite(sim.potential()).all()
        sim.energies()
        for stage in (sim.density, sim.potential_gradient, sim.solve_potential, sim.compute_forces):
            with pytest.raises(pm.PMError):
                stage()
        sim.deposit(); sim.solve_potential(); sim.gradient(); sim.compute_forces()
        assert sim.density().sum() > 0 and len(sim.potential_gradient()) == 3 and sim.forces().shape == (5000, 3)
    with pm.Simulation(16, 100.0, capacity=4) as sim:
        sim.set_mesh(0, np.ones((16, 16, 16), np.float32))             # PM_MESH_DENSITY
        sim.solve_potential()
        assert sim.potential().shape == (16, 16, 16)


# ---- optional softening of the solve (pm_set_softening; the reference has none) -----------------

def _blob_density(G, seed):
    rng = np.random.default_rng(seed)
    n = 20000
    pos = (np.float32(0.5 * 10000) + rng.normal(0, 600, (n, 3))).astype(np.float32) % np.float32(10000)
    return pos, O.CIC(pos, G, 10000, np.ones(n, np.float32))


@pytest.mark.parametrize("cells", [0.75, 2.0])
def test_softened_potential_against_oracle(pm, cells):
    G, L = 32, 10000
    pos, dens = _blob_density(G, 3)
    length = cells * L / G
    ref = O.compute_potential(dens, G, soft=O.softening_coefficient(length, G, L))
    bare = O.compute_potential(dens, G)
    assert np.abs(ref - bare).max() > 0.05 * np.abs(bare).max()          # the softening does something
    with pm.Simulation(G, L, pos, None, None) as sim:
        sim.set_softening(length)
        sim.set_mesh(0, dens)
        sim.solve_potential()
        phi = sim.potential()
        assert np.abs(phi - ref).max() <= 1e-5 * np.abs(ref).max()
        # length 0 restores the reference's solve bit for bit
        sim.set_softening(0.0)
        sim.solve_potential()
        again = sim.potential()
    with pm.Simulation(G, L, pos, None, None) as sim:
        sim.set_mesh(0, dens)
        sim.solve_potential()
        assert np.array_equal(sim.potential(), again)
    with pytest.raises(pm.PMError, match="softening"):
        with pm.Simulation(G, L, pos, None, None) as sim:
            sim.set_softening(-1.0)


def test_softened_potential_hand_written_transforms_match_rocfft(pm):
    # G = 512: the fused forward-x / Green / inverse-x pass against rocFFT's 3-D plans around the plain multiply, and
    # the fused step (spectrum rows straight from the deposit) against the staged one
    G, L, n = 512, 10000, 200_000
    rng = np.random.default_rng(11)
    pos = (np.float32(5000) + rng.normal(0, 400, (n, 3))).astype(np.float32) % np.float32(L)
    vel = rng.normal(0, 10, (n, 3)).astype(np.float32)
    length = 1.5 * L / G
    out = []
    for rocfft_only in (False, True):
        with pm.Simulation(G, L, pos, vel, None, rocfft_only=rocfft_only) as sim:
            sim.set_softening(length)
            sim.deposit()
            sim.solve_potential()
            out.append(sim.potential())
    assert np.abs(out[0] - out[1]).max() <= 1e-5 * np.abs(out[1]).max()
    with pm.Simulation(G, L, pos, vel, None) as sim:
        sim.deposit()
        sim.solve_potential()
        bare = sim.potential()
    assert np.abs(out[0] - bare).max() > 0.002 * np.abs(bare).max()     # a blob 20 cells wide: 0.4 % at its centre
    res = []
    for fused in (True, False):
        with pm.Simulation(G, L, pos, vel, None, fused_gradient=fused) as sim:
            sim.set_softening(length)
            sim.step(0.005, 2)
            res.append((sim.positions(), sim.velocities()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


def test_softened_ngp_branch_against_oracle(pm):
    G, L = 32, 10000
    pos, _ = _blob_density(G, 5)
    dens = O.NGP(pos, G, L, np.ones(len(pos), np.float32))
    length = 1.25 * L / G
    ref = O.compute_potential(dens, G, soft=O.softening_coefficient(length, G, L))
    assert ref.dtype == np.float64
    with pm.Simulation(G, L, pos, None, None, interpolation="NGP") as sim:
        sim.set_softening(length)
        sim.set_mesh_f64(0, dens)
        sim.solve_potential_f64()
        assert np.abs(sim.mesh_f64(1) - ref).max() <= 1e-12 * np.abs(ref).max()
