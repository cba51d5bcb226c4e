#!/bin/bash
# kernel-trace statistics of optimize() at C5, one optimiser thread (the profiler's launch interception is not safe
# against launches from several host threads: it crashed in one of two runs)
export TTM_OPT_THREADS=1
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_opt --output-format csv -- python3 $R/tools/time_opt_batch.py C5 > $R/gpurun_out/prof_opt.log 2>&1
cd $R
find gpurun_out/prof_opt -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} gpurun_out/prof_opt_kernel_stats.csv
