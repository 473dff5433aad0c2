#!/bin/bash
# For every tuning build under lib/var/*.so: two steps of the headline workload and the library's own per-kernel timers.
# (Variants that break the results are fine here: only durations are read.)
R=$PWD
mkdir -p $R/gpurun_out
: > $R/gpurun_out/kernel_variants.log
for so in $R/nbody-simulation_amd/lib/libpm.so $R/nbody-simulation_amd/lib/var/*.so; do
  tag=$(basename $so .so)
  echo "== $tag" >> $R/gpurun_out/kernel_variants.log
  PM_LIBRARY=$so timeout -k 10 120 python $R/bench.py --steps ${STEPS:-2} --warmup 1 --cpu-baseline skip --no-uniform --repeats 0 ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['stage_ms'].items()}, d.get('kernel_ms'))" >> $R/gpurun_out/kernel_variants.log || echo failed >> $R/gpurun_out/kernel_variants.log
done
cat $R/gpurun_out/kernel_variants.log
