#!/bin/bash
# The three single-GPU reporting configurations without the profiler: headline (with its uniform set) and 1M / 128^3.
mkdir -p gpurun_out
timeout -k 10 200 python bench.py --steps 40 --warmup 3 --cpu-baseline skip --repeats 1 > gpurun_out/three_headline.log 2>&1 || exit 1
timeout -k 10 100 python bench.py --n 1000000 --grid 128 --workload uniform --steps 400 --warmup 10 --cpu-baseline skip --repeats 1 > gpurun_out/three_c2.log 2>&1 || exit 1
python - <<'PY'
import json
for f in ("three_headline", "three_c2"):
    d = json.loads([l for l in open(f"gpurun_out/{f}.log") if l.startswith("{")][-1])
    print(f, round(d["ms_per_step"], 4), {k: round(v, 3) for k, v in d["stage_ms"].items()}, "uniform:", (d.get("uniform_set") or {}).get("ms_per_step"))
PY
