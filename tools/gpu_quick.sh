#!/bin/bash
# Quick GPU check of a change to the particle kernels: the parity tests that exercise the binning, then the headline
# bench under rocprofv3 (kernel trace) with its summary.  Usage: tools/gpu_quick.sh [pytest -k expression]
mkdir -p gpurun_out
K=${1:-"edge_cases or step0 or k_steps or cell_order or full_size or fused_step or deposit or slab"}
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread -k "$K" > gpurun_out/t.log 2>&1
rc=$?
tail -4 gpurun_out/t.log
[ $rc -ne 0 ] && exit $rc
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r -- python $R/bench.py --steps 10 --warmup 2 --cpu-baseline skip --no-uniform --no-slab-probe > $R/gpurun_out/b.log 2>&1 || exit 1
cd $R
python tools/prof_summary.py gpurun_out/prof > gpurun_out/prof_summary.txt 2>&1
head -24 gpurun_out/prof_summary.txt
timeout -k 10 150 python bench.py --steps 20 --warmup 3 --cpu-baseline skip --no-slab-probe > gpurun_out/b2.log 2>&1 || exit 1
tail -1 gpurun_out/b2.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms'], d.get('uniform_set'))"
