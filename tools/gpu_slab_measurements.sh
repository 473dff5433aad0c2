#!/bin/bash
# The slab-path measurements that one GPU can make: a rank's local work (nothing exchanged) for the headline set in eight
# slabs and in eight windows of equal load, one slab of the 8-GPU configuration, and the headline set as one slab through
# the multi-GPU path.  Results under gpurun_out/ (the last line of each log is the summary).
mkdir -p gpurun_out
timeout -k 10 300 python -u tools/slab_local_phases.py --grid 512 --slabs 8 --workload two_galaxy --n 16777216 --steps 5 > gpurun_out/lp_slabs.log 2>&1 || exit 1
timeout -k 10 300 python -u tools/slab_local_phases.py --grid 512 --slabs 8 --workload two_galaxy --n 16777216 --steps 5 --windows > gpurun_out/lp_windows.log 2>&1 || exit 1
timeout -k 10 300 python -u tools/slab_local_phases.py > gpurun_out/lp_2048.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --gpus 1 --parallelism slab --steps 20 --warmup 3 --cpu-baseline skip > gpurun_out/slab1.log 2>&1 || exit 1
for f in lp_slabs lp_windows lp_2048 slab1; do tail -1 gpurun_out/$f.log | cut -c1-400; done
