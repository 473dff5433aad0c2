#!/bin/bash
# The 1M / 128^3 correctness configuration with its sorts going either way (look-back chain: 6 launches a sort; counting
# kernel and scan: 12), and two sizes around it: where the bound between the two belongs.
mkdir -p gpurun_out
: > gpurun_out/c2_ways.log
for n in 1000000 2097152 4194304; do
for tiles in 0 100000; do
  echo "== n=$n PM_SORT_CHAIN_TILES=$tiles" >> gpurun_out/c2_ways.log
  PM_SORT_CHAIN_TILES=$tiles timeout -k 10 150 python bench.py --n $n --grid 128 --workload uniform --steps 200 --warmup 5 --cpu-baseline skip --repeats 1 >> gpurun_out/c2_ways.log 2>&1 || exit 1
done
done
python - <<'PY'
import json
name = None
for line in open("gpurun_out/c2_ways.log"):
    if line.startswith("== "):
        name = line[3:].strip()
    elif line.startswith("{"):
        d = json.loads(line)
        print(f"{name:40s} {d['ms_per_step']:.4f} {d['ms_per_step_blocks']}  bin={d['stage_ms']['bin']:.4f}")
PY
