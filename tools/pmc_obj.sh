#!/bin/bash
# PMC passes over tools/obj_bench.py (k_int_objective and k_int_objective_walk in one process): instruction mix and wait counters
# per kernel -> gpurun_out/pmc_obj_<W>.json       (GPU box, repo root)     tools/pmc_obj.sh C2a C5int
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for W in "$@"; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM -d $R/gpurun_out/poa_$W --output-format csv -- python3 $R/tools/obj_bench.py $W --passes 2 > $R/gpurun_out/poa_$W.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES -d $R/gpurun_out/pob_$W --output-format csv -- python3 $R/tools/obj_bench.py $W --passes 2 > $R/gpurun_out/pob_$W.log 2>&1
  (cd $R && python3 tools/pmc_summary.py gpurun_out/poa_$W gpurun_out/pob_$W > gpurun_out/pmc_obj_$W.json)
  rm -rf $R/gpurun_out/poa_$W $R/gpurun_out/pob_$W
  echo "$W done"
done
