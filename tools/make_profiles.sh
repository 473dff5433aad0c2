#!/bin/bash
# Runs on the GPU box: kernel trace + stats of `bench.py --steps 10`, then FETCH_SIZE and WRITE_SIZE counter
# passes (each in a run of its own).  Results land under gpurun_out/profiles_raw/.
R=$PWD
OUT=$R/gpurun_out/profiles_raw
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- python $R/bench.py --steps 10 --warmup 2 --cpu-baseline skip --no-uniform --repeats 0 > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o r -- python $R/bench.py --steps 3 --warmup 1 --cpu-baseline skip --no-uniform --repeats 0 > $OUT/fetch.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o r -- python $R/bench.py --steps 3 --warmup 1 --cpu-baseline skip --no-uniform --repeats 0 > $OUT/write.log 2>&1 || exit 1
# the same two counter passes on the uniform particle set of the same size (bench.py reports it beside the headline)
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_uniform -o r -- python $R/bench.py --steps 3 --warmup 1 --cpu-baseline skip --workload uniform --repeats 0 > $OUT/fetch_uniform.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_uniform -o r -- python $R/bench.py --steps 3 --warmup 1 --cpu-baseline skip --workload uniform --repeats 0 > $OUT/write_uniform.log 2>&1 || exit 1
