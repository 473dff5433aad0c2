#!/bin/bash
# rocprofv3 kernel trace of a short bench run per library variant (lib/libpm.so and lib/var/*.so): per-kernel averages
# for tuning one kernel.  Results: gpurun_out/pv_<name>/r_results.db
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for so in $R/nbody-simulation_amd/lib/libpm.so $R/nbody-simulation_amd/lib/var/*.so; do
  name=$(basename $so .so)
  rm -rf $R/gpurun_out/pv_$name
  PM_LIBRARY=$so timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pv_$name -o r -- python $R/bench.py --steps 10 --warmup 2 --cpu-baseline skip --no-uniform > $R/gpurun_out/pv_$name.log 2>&1 || exit 1
done
