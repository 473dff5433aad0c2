#!/bin/bash
# Quick GPU check of a change to the binning: the parity tests that see large sorts, then the headline bench under
# rocprofv3 (kernel trace).  Usage: tools/gpu_sort.sh [pytest -k expression]
mkdir -p gpurun_out
K=${1:-"headline_size or full_size or correctness_configuration or cell_order or few_movers"}
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread -k "$K" > gpurun_out/t.log 2>&1
rc=$?
tail -4 gpurun_out/t.log
[ $rc -ne 0 ] && exit $rc
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r -- python $R/bench.py --steps 10 --warmup 2 --cpu-baseline skip --no-uniform > $R/gpurun_out/b.log 2>&1 || exit 1
cd $R
python tools/prof_db.py gpurun_out/prof/r_results.db 24 > gpurun_out/prof_summary.txt 2>&1
cut -c1-150 gpurun_out/prof_summary.txt
timeout -k 10 150 python bench.py --steps 20 --warmup 3 --cpu-baseline skip > gpurun_out/b2.log 2>&1 || exit 1
tail -1 gpurun_out/b2.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms'], d.get('uniform_set'))"
