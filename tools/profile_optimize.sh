#!/bin/bash
# kernel-trace statistics of optimize(), one optimiser thread (the profiler's launch interception is not safe against
# launches from several host threads: it crashed in one of two runs).  usage: tools/profile_optimize.sh [C5 | C2a]
export TTM_OPT_THREADS=1
R=${GRAFT_REPO_ROOT:-$PWD}
W=${1:-C5}
if [ "$W" = "C2a" ]; then PROG="$R/tools/time_opt_int.py"; ARGS="100000"; else PROG="$R/tools/time_opt_batch.py"; ARGS="$W"; fi
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_opt
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_opt --output-format csv -- python3 $PROG $ARGS > $R/gpurun_out/prof_opt.log 2>&1
cd $R
find gpurun_out/prof_opt -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} gpurun_out/prof_opt_kernel_stats.csv
