#!/bin/bash
# Runs bench.py once per library variant under nbody-simulation_amd/lib/var (tuning builds).
mkdir -p gpurun_out
: > gpurun_out/variants.log
for so in nbody-simulation_amd/lib/libpm.so nbody-simulation_amd/lib/var/*.so; do
  echo "== $so" >> gpurun_out/variants.log
  PM_LIBRARY=$PWD/$so timeout -k 10 150 python bench.py --steps 20 --warmup 3 --cpu-baseline skip >> gpurun_out/variants.log 2>&1 || exit 1
done
