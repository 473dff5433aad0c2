#!/bin/bash
# GPU parity tests, then the headline bench under rocprofv3 (kernel trace) and the uniform workload.
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t.log 2>&1
rc=$?
tail -3 gpurun_out/t.log
[ $rc -ne 0 ] && exit $rc
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r -- python $R/bench.py --steps 10 --warmup 2 --cpu-baseline skip --no-uniform > $R/gpurun_out/b.log 2>&1 || exit 1
cd $R
timeout -k 10 150 python bench.py --steps 20 --warmup 3 --cpu-baseline skip --no-uniform >> gpurun_out/b.log 2>&1 || exit 1
timeout -k 10 150 python bench.py --steps 20 --warmup 3 --cpu-baseline skip --no-uniform --workload uniform >> gpurun_out/b.log 2>&1 || exit 1
