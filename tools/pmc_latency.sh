#!/bin/bash
# Stall / latency counters for the map kernels (separate passes; run on the GPU box from the repo root).
#   bash tools/pmc_latency.sh [bench args...]
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_LEVEL_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/lat_$i --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline --no-optimize --no-other-configs "$@" > $R/gpurun_out/lat_$i.log 2>&1
done
cd $R && python3 tools/pmc_summary.py gpurun_out/lat_* > gpurun_out/lat_summary.json
