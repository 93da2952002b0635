#!/bin/bash
# PMC view of the byte-space tile kernel on C3 (non-ASCII) and C2 (ASCII) + the Latin-1 kernel
set -u
bash tools/pmc_pass.sh r03_bytes_c3 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU" -- tools/path_bench.py --workload C3 --iters 3 --paths bytes_mask > gpurun_out/r03_bytes_c3.log 2>&1
grep -A40 "k_tiles_main" gpurun_out/r03_bytes_c3_pmc_summary.txt | head -60
