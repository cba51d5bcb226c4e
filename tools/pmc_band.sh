#!/bin/bash
# Counter passes of tools/band_check.py (run on the GPU box from the repo root): instruction mix, activity, LDS / waits.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
export TTM_BAND_CHECK_FAST=1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU -d $R/gpurun_out/bmix1 --output-format csv -- python3 $R/tools/band_check.py 1000000 --no-oracle > $R/gpurun_out/bmix1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH -d $R/gpurun_out/bmix2 --output-format csv -- python3 $R/tools/band_check.py 1000000 --no-oracle > $R/gpurun_out/bmix2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY -d $R/gpurun_out/bmix3 --output-format csv -- python3 $R/tools/band_check.py 1000000 --no-oracle > $R/gpurun_out/bmix3.log 2>&1
cd $R && python3 tools/pmc_summary.py gpurun_out/bmix1 gpurun_out/bmix2 gpurun_out/bmix3 > gpurun_out/bmix.json
