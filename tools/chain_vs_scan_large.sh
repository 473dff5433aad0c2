#!/bin/bash
# Large sorts with 10-bit places (30 key bits: the 1024^3 mesh, a slab of the 2048^3 mesh) both ways: look-back chain against
# counting kernel + scan.
mkdir -p gpurun_out
for tiles in 0 1000000; do
  echo "== PM_SORT_CHAIN_TILES=$tiles"
  PM_SORT_CHAIN_TILES=$tiles timeout -k 10 200 python bench.py --steps 10 --warmup 2 --cpu-baseline skip --device-ic --workload disk --n 67108864 --grid 1024 --repeats 1 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('disk1024', round(d['ms_per_step'],3), d['stage_ms']['bin'], d['kernel_ms'])" || exit 1
  PM_SORT_CHAIN_TILES=$tiles timeout -k 10 200 python -u tools/slab_local_phases.py 2>/dev/null | tail -1 | cut -c1-330 || exit 1
done
