#!/bin/bash
# Runs bench.py once per library variant under nbody-simulation_amd/lib/var (tuning builds), twice over; prints ms per
# step and the stage times of every run.
mkdir -p gpurun_out
: > gpurun_out/variants.log
for rep in 1 2; do
for so in nbody-simulation_amd/lib/var/*.so; do
  echo "== $so" >> gpurun_out/variants.log
  PM_LIBRARY=$PWD/$so timeout -k 10 150 python bench.py --steps 40 --warmup 3 --cpu-baseline skip --no-uniform --repeats 0 >> gpurun_out/variants.log 2>&1 || exit 1
done
done
python - <<'PY'
import json
name = None
for line in open("gpurun_out/variants.log"):
    if line.startswith("== "):
        name = line.split("/")[-1].strip()
    elif line.startswith("{"):
        d = json.loads(line)
        print(f"{name:16s} {d['ms_per_step']:.4f}  " + " ".join(f"{k}={v:.3f}" for k, v in d["stage_ms"].items() if v))
PY
