#!/bin/bash
# Kernel trace of the headline workload at several particle counts (how the binning passes scale with the number of tiles).
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for n in ${SIZES:-2097152 4194304 8388608}; do
  rm -rf $R/gpurun_out/prof_$n
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$n -o r -- python $R/bench.py --n $n --steps 10 --warmup 2 --cpu-baseline skip --no-uniform --no-slab-probe > $R/gpurun_out/b_$n.log 2>&1 || exit 1
  echo "== n=$n"; python $R/tools/prof_db.py $R/gpurun_out/prof_$n/r_results.db 14 | cut -c1-60,90-130 | grep -i "radix\|digit\|reorder\|row_start\|gather"
done
