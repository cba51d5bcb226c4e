#!/bin/bash
# Instruction mix by type of the map kernels (run on the GPU box from the repo root): three counter passes of a short bench.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline --no-optimize --no-other-configs"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 -d $R/gpurun_out/mix1 --output-format csv -- python3 $R/bench.py $ARGS "$@" > $R/gpurun_out/mix1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS -d $R/gpurun_out/mix2 --output-format csv -- python3 $R/bench.py $ARGS "$@" > $R/gpurun_out/mix2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM -d $R/gpurun_out/mix3 --output-format csv -- python3 $R/bench.py $ARGS "$@" > $R/gpurun_out/mix3.log 2>&1
cd $R && python3 tools/pmc_summary.py gpurun_out/mix1 gpurun_out/mix2 gpurun_out/mix3 > gpurun_out/mix.json
