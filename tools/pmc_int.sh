#!/bin/bash
# PMC passes over tools/time_int.py (generic and dense integrated kernels in one process): instruction mix and wait
# counters per kernel -> gpurun_out/pmc_int_<W>.json       (GPU box, repo root)     tools/pmc_int.sh C2a C5int
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for W in "$@"; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM -d $R/gpurun_out/pia_$W --output-format csv -- python3 $R/tools/time_int.py $W > $R/gpurun_out/pia_$W.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES -d $R/gpurun_out/pib_$W --output-format csv -- python3 $R/tools/time_int.py $W > $R/gpurun_out/pib_$W.log 2>&1
  (cd $R && python3 tools/pmc_summary.py gpurun_out/pia_$W gpurun_out/pib_$W > gpurun_out/pmc_int_$W.json)
  rm -rf $R/gpurun_out/pia_$W $R/gpurun_out/pib_$W
  echo "$W done"
done
