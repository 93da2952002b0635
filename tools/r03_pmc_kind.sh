#!/bin/bash
set -u
bash tools/pmc_pass.sh r03_kind_c2 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE" -- tools/path_bench.py --workload C2 --iters 3 --paths kind_mask,mask > gpurun_out/r03_kind_c2.log 2>&1
grep -A24 "k_tiles_main" gpurun_out/r03_kind_c2_pmc_summary.txt | head -80
