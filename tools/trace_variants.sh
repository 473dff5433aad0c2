#!/bin/bash
# For every tuning build under lib/var/t_*.so (and trace.so): a short bench (ms per step, kernel times) and the deposit wave trace.
R=$PWD
mkdir -p $R/gpurun_out
: > $R/gpurun_out/trace_variants.log
for so in $R/nbody-simulation_amd/lib/var/*.so; do
  tag=$(basename $so .so)
  echo "== $tag" >> $R/gpurun_out/trace_variants.log
  PM_LIBRARY=$so timeout -k 10 120 python $R/bench.py --steps 20 --warmup 3 --cpu-baseline skip --no-uniform --repeats 2 ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step_median'],4), d['ms_per_step_blocks'], {k: round(v,3) for k,v in d['stage_ms'].items()}, d.get('kernel_ms'))" >> $R/gpurun_out/trace_variants.log || exit 1
  if [ -z "$NO_TRACE" ]; then PM_TAG=${tag}_ PM_LIBRARY=$so timeout -k 10 120 python $R/tools/deposit_trace.py ${WORKLOAD:-two_galaxy} > /dev/null || exit 1; fi
done
cat $R/gpurun_out/trace_variants.log
