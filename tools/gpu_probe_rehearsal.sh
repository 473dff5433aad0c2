#!/bin/bash
# Two ranks on the one GPU under the driver's launcher (bench.py --gpus 2 is the slab run),
# first with host-staged exchanges (gloo: the whole child path runs), then with nccl (two ranks on one GPU: RCCL refuses,
# and the line must carry that error rather than a timeout).
mkdir -p gpurun_out
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo > gpurun_out/probe_gloo.log 2>&1 || { tail -20 gpurun_out/probe_gloo.log; exit 1; }
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29641 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/probe_nccl.log 2>&1 || { tail -20 gpurun_out/probe_nccl.log; exit 1; }
grep -h '^{' gpurun_out/probe_gloo.log gpurun_out/probe_nccl.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['n_gpus'], d['ms_per_step'], json.dumps(d.get('slab_decomposed'))[:700])"
