#!/bin/bash
# GPU tests of the device-side particle sets, then the default bench line (with the CPU baseline) and the 1024^3 disk drawn on the device.
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_generate.py -x -q --timeout 120 --timeout-method thread > gpurun_out/tg.log 2>&1
rc=$?
tail -5 gpurun_out/tg.log
[ $rc -ne 0 ] && exit $rc
( time timeout -k 10 400 python bench.py ) > gpurun_out/bench_default.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu-baseline skip --no-uniform --device-ic --workload disk --n 67108864 --grid 1024 > gpurun_out/bench_disk_device_ic.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --cpu-baseline skip --no-uniform --device-ic > gpurun_out/bench_device_ic.log 2>&1 || exit 1
tail -c 600 gpurun_out/bench_default.log
