#!/bin/bash
# SQ / LDS counter passes (one rocprofv3 run each) over a short run of the headline workload: what bounds the transforms.
set -e
bash tools/pmc.sh fq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY
bash tools/pmc.sh fq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
bash tools/pmc.sh fq4 SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_WAIT_ANY
for t in fq1 fq2 fq4; do echo "== $t"; python tools/pmc_summary.py gpurun_out/pmc_$t | cut -c1-60,70-250 | grep -i "kernel\|col_fft\|c2r\|gather" ; done > gpurun_out/pmc_fft.txt
cat gpurun_out/pmc_fft.txt
