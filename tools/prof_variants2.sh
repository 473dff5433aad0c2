#!/bin/bash
# Kernel trace of bench.py --steps 10 for the default library and every tuning build under lib/var: the binning kernels' averages.
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for so in $R/nbody-simulation_amd/lib/libpm.so $R/nbody-simulation_amd/lib/var/*.so; do
  tag=$(basename $so .so)
  rm -rf $R/gpurun_out/pv_$tag
  PM_LIBRARY=$so timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pv_$tag -o r -- python $R/bench.py --steps 10 --warmup 2 --cpu-baseline skip --no-uniform --no-slab-probe ${BENCH_ARGS} > $R/gpurun_out/pv_$tag.log 2>&1 || exit 1
  echo "== $tag  $(tail -1 $R/gpurun_out/pv_$tag.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['stage_ms'].items()})")"
  python $R/tools/prof_db.py $R/gpurun_out/pv_$tag/r_results.db 16 | cut -c1-70,90-130 | grep -i "${KERNELS:-radix\|digit\|reorder\|row_start}"
done
