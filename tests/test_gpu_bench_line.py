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


def test_one_slab_line():
    d = run_bench("--gpus", "1", "--parallelism", "slab", "--n", "1048576", "--grid", "128", "--workload", "uniform", "--steps", "3", "--warmup", "2")
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["config"]["collective_world_size"] == 1
    assert set(d["phases_ms"]) >= {"deposit", "transpose", "solve_x", "transpose_back", "gather_kick_drift", "migration"}
    assert abs(d["value"] - 1048576 * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-6 * d["value"]
