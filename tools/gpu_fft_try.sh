#!/bin/bash
# parity of the Poisson solve per library variant, then the 1024^3 disk and the headline with each
mkdir -p gpurun_out
: > gpurun_out/fft_try.log
for so in nbody-simulation_amd/lib/libpm.so nbody-simulation_amd/lib/var/*.so; do
  echo "== $so tests" >> gpurun_out/fft_try.log
  PM_LIBRARY=$PWD/$so timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_slab.py -x -q --timeout 120 --timeout-method thread -k "poisson or fused_step or g512" >> gpurun_out/fft_try.log 2>&1 || { tail -20 gpurun_out/fft_try.log; exit 1; }
  echo "== $so 1024" >> gpurun_out/fft_try.log
  PM_LIBRARY=$PWD/$so timeout -k 10 200 python bench.py --workload disk --particles 134217728 --grid 1024 --steps 8 --warmup 2 --cpu-baseline skip >> gpurun_out/fft_try.log 2>&1 || exit 1
  echo "== $so 512" >> gpurun_out/fft_try.log
  PM_LIBRARY=$PWD/$so timeout -k 10 150 python bench.py --steps 20 --warmup 3 --cpu-baseline skip >> gpurun_out/fft_try.log 2>&1 || exit 1
done
