#!/bin/bash
# One rocprofv3 counter pass over a short bench run: tools/pmc.sh <tag> <counter> [<counter> ...]
# Writes gpurun_out/pmc_<tag>/ (csv).  Counter passes are run on their own, with the kernel trace only.
tag=$1; shift
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_$tag
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$tag -o r -- python $R/bench.py --steps 3 --warmup 1 --cpu-baseline skip --no-uniform --repeats 0 ${PMC_BENCH_ARGS} > $R/gpurun_out/pmc_$tag.log 2>&1
