"""bench.py prints ONE JSON line with the fields its contract names; here on a small configuration (2^20 particles on a
128^3 mesh, a few steps, no CPU baseline), run as a child process the way the driver runs it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line():
    d = run_bench("--n", "1048576", "--grid", "128", "--workload", "uniform", "--steps", "4", "--warmup", "2", "--cpu-baseline", "skip", "--repeats", "1")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["unit"] == "particle-updates/s" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1048576 * 4 / (d["ms_per_step"] * 4e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0.0 < r["frac"] < 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["ms_per_launch"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    assert r["kernel"] == r["ranking"][0]["kernel"]                      # the longest priced launch is the one named
    assert all(a["ms_per_launch"] >= b["ms_per_launch"] for a, b in zip(r["ranking"], r["ranking"][1:]))
    assert set(d["stage_ms"]) >= {"bin", "deposit", "fft_fwd", "green", "fft_inv", "gather"}
    assert "from_order_before" in d["last_binning"]


def check_slab_line(d, n, steps, world):
    assert d["n_gpus"] == world and d["scaling"] == "strong" and d["steps"] == steps
    c = d["config"]
    assert c["collective_world_size"] == world and (c["backend"] in ("gloo", "nccl") if world > 1 else c["backend"] is None)
    assert sum(c["particles_per_rank_at_start"]) == n and sum(c["particles_per_rank_at_end"]) == n
    assert set(d["phases_ms"]) >= {"deposit", "transpose", "solve_x", "transpose_back", "gather_kick_drift", "migration"}
    assert abs(d["value"] - n * steps / (d["ms_per_step"] * steps * 1e-3)) <= 1e-6 * d["value"]
    assert len(d["kernel_ms_per_rank"]) == world and len(d["local_work_ms_per_rank"]) == world
    assert all(k["gather"] > 0 and k["deposit_rows"] > 0 and k["col_x_green"] > 0 for k in d["kernel_ms_per_rank"])
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0.0 < r["frac"] < 1.0 and 0 <= r["rank"] < world
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["ms_per_launch"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    assert d["step_roofline"]["peak"] == 8000.0 * world


def test_one_slab_line():
    d = run_bench("--gpus", "1", "--parallelism", "slab", "--n", "1048576", "--grid", "128", "--workload", "uniform", "--steps", "3", "--warmup", "2",
                  "--cpu-baseline", "skip")
    check_slab_line(d, 1048576, 3, 1)
    assert d["config"]["particle_ownership"] == "by slab"


def test_two_rank_line_with_windows_and_device_drawn_particles():
    """Two ranks on the one GPU of the test box, exchanging through the host (gloo): a rehearsal of the line a multi-GPU run
    prints -- the clustered set is held by windows of equal particle count, and every rank draws its share on the device."""
    d = run_bench("--gpus", "2", "--backend", "gloo", "--n", "2097152", "--grid", "128", "--workload", "two_galaxy", "--steps", "3", "--warmup", "2",
                  "--no-uniform", "--device-ic", "--balance", "on", "--balance-every", "2")
    check_slab_line(d, 2097152, 3, 2)
    c = d["config"]
    assert c["particle_ownership"] == "windows of equal particle count" and c["backend"] == "gloo"
    assert c["window_cuts_at_start"][0] == 0 and c["window_cuts_at_start"][-1] == 128
    assert max(c["particles_per_rank_at_start"]) <= 1.2 * 2097152 / 2
    assert "drawn on the device" in d["data"]
