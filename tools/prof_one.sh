#!/bin/bash
# prof_one.sh TAG bench-args...: kernel trace of a short bench run, summary into gpurun_out/prof_TAG.txt
R=$PWD
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$tag
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o r -- python $R/bench.py --steps 5 --warmup 2 --cpu-baseline skip --no-uniform --repeats 0 "$@" > $R/gpurun_out/prof_$tag.log 2>&1 || exit 1
cd $R
python tools/prof_summary.py gpurun_out/prof_$tag > gpurun_out/prof_$tag.txt 2>&1
head -${LINES_SHOWN:-16} gpurun_out/prof_$tag.txt
