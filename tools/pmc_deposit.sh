#!/bin/bash
# SQ / LDS counter passes (one rocprofv3 run each) over a short run of the headline workload: what bounds the deposit kernels.
set -e
PMC_BENCH_ARGS="--no-slab-probe" bash tools/pmc.sh sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY
bash tools/pmc.sh sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
bash tools/pmc.sh sq3 SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_THREAD_CYCLES_VALU
bash tools/pmc.sh sq4 SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_WAIT_ANY
for t in sq1 sq2 sq3 sq4; do echo "== $t"; python tools/pmc_summary.py gpurun_out/pmc_$t | cut -c1-60,70-250 | grep -i "kernel\|deposit\|gather\|radix\|reorder\|tile_counts\|scan_chunks\|chunk_sums" ; done > gpurun_out/pmc_deposit.txt
cat gpurun_out/pmc_deposit.txt
