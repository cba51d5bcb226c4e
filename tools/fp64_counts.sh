#!/bin/bash
# Counted fp64 operations of the FP64-bound kernels (integrated-rectifier maps, Gram matrices, objective reductions):
# PMC passes of short bench runs per workload (GPU box, repo root) -> gpurun_out/fp64_<W>.json -> profiles/fp64_counts.json
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for W in C2a C3int C5int EX01 C5; do
  ARGS="--workload $W --steps 2 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline --no-other-configs --no-api --no-optimize"   # (--no-optimize: every dispatch of the objective kernels then belongs to a pass of bench.py:objective_roofline)   # (--no-api: full-size launches only - the pipelined host boundary launches the same kernels on chunks of rows)
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F64 -d $R/gpurun_out/fp64a_$W --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/fp64a_$W.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/fp64b_$W --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/fp64b_$W.log 2>&1
  (cd $R && python3 tools/pmc_summary.py gpurun_out/fp64a_$W gpurun_out/fp64b_$W > gpurun_out/fp64_$W.json)
  rm -rf $R/gpurun_out/fp64a_$W $R/gpurun_out/fp64b_$W          # (raw counter CSVs: tens of MB; gpurun_out returns <= 64 MiB)
  echo "$W done"
done
cd $R && python3 tools/fp64_counts.py gpurun_out C2a C3int C5int EX01 C5 && cp profiles/fp64_counts.json profiles/r05_fp64_pmc_summary.json gpurun_out/
