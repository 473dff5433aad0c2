#!/bin/bash
# Kernel trace + stats of the other measured configurations: the uniform set at the headline size, the 64M-particle
# disk on the 1024^3 mesh (drawn on the device), and the correctness configuration.  Results under gpurun_out/profiles_other/.
R=$PWD
OUT=$R/gpurun_out/profiles_other
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/uniform -o r -- python $R/bench.py --steps 10 --warmup 2 --cpu-baseline skip --no-uniform --repeats 0 --workload uniform > $OUT/uniform.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/disk1024 -o r -- python $R/bench.py --steps 10 --warmup 2 --cpu-baseline skip --device-ic --workload disk --n 67108864 --grid 1024 > $OUT/disk1024.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2 -o r -- python $R/bench.py --steps 100 --warmup 5 --cpu-baseline skip --workload uniform --n 1000000 --grid 128 > $OUT/c2.log 2>&1 || exit 1
