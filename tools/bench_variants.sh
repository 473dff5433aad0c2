#!/bin/bash
# Runs bench.py once per library variant under nbody-simulation_amd/lib/var (tuning builds); whole-step time only
# matters here (the per-stage timers run the column passes ungrouped).
mkdir -p gpurun_out
: > gpurun_out/variants.log
for rep in 1 2; do
for so in nbody-simulation_amd/lib/libpm.so nbody-simulation_amd/lib/var/*.so; do
  echo "== $so" >> gpurun_out/variants.log
  PM_LIBRARY=$PWD/$so timeout -k 10 150 python bench.py --steps 40 --warmup 3 --cpu-baseline skip --no-uniform >> gpurun_out/variants.log 2>&1 || exit 1
done
done
