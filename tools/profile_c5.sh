#!/bin/bash
# Profiles of the C5 bench for profiles/ (run on the GPU box from the repo root):
#   kernel-trace stats, then separate PMC passes (FETCH_SIZE, WRITE_SIZE, SQ instruction mix / stalls).
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
# kernel-trace statistics: the default bench command (pre-warmed clocks, 200 timed steps) without the host-side extras;
# counter passes: a few dispatches are enough and the counters do not depend on the clock
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats --output-format csv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/prof_stats.log 2>&1
# the TIMED launches alone (tools/rocprof_timed.py: dispatches [first_timed_step, first_timed_step + K) of each map kernel out of the
# trace of the default command without the host-side extras) -> gpurun_out/r05_timed_launches.json (copy to profiles/)
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_timed --output-format csv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-optimize --no-other-configs --no-api > $R/gpurun_out/prof_timed.log 2>&1
(cd $R && python3 tools/rocprof_timed.py gpurun_out/prof_timed gpurun_out/prof_timed.log --out gpurun_out/r05_timed_launches.json > gpurun_out/prof_timed_summary.txt 2>&1; find gpurun_out/prof_timed -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} gpurun_out/prof_timed_kernel_stats.csv; rm -rf gpurun_out/prof_timed)
# (--no-api: the pipelined host boundary launches the same kernels on chunks of rows - full-size launches only for the counters)
ARGS="--steps 5 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline --no-optimize --no-other-configs --no-api"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_write --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM -d $R/gpurun_out/prof_sq --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_sq.log 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq > gpurun_out/prof_pmc_summary.json
find gpurun_out/prof_stats -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} gpurun_out/prof_kernel_stats.csv
head -3 gpurun_out/prof_stats.log > /dev/null 2>&1
# (raw traces: hundreds of MB - the 2000 filter cycles alone are half a million dispatches; gpurun_out returns <= 64 MiB)
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq
