#!/bin/bash
# Instruction-mix counters of the map kernels in one pass (run on the GPU box from the repo root).
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $R/gpurun_out/quick --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline --no-optimize --no-other-configs "$@" > $R/gpurun_out/quick.log 2>&1
cd $R && python3 tools/pmc_summary.py gpurun_out/quick > gpurun_out/quick.json
