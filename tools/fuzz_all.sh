#!/bin/bash
# End-of-round sweep of the random-map tools on the GPU (one call): logs under gpurun_out/, summaries copied to profiles/ by hand.
#   bash tools/fuzz_all.sh [TAG]
tag=${1:-r05}
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 300 python tools/fuzz_gpu.py 400 700 > gpurun_out/${tag}_fuzz_gpu.log 2>&1; echo "fuzz_gpu rc $?"; tail -1 gpurun_out/${tag}_fuzz_gpu.log
timeout -k 10 300 python tools/fuzz_few.py 6000 8000 > gpurun_out/${tag}_fuzz_few.log 2>&1; echo "fuzz_few rc $?"; tail -1 gpurun_out/${tag}_fuzz_few.log
timeout -k 10 300 python tools/fuzz_hot.py 300 500 > gpurun_out/${tag}_fuzz_hot.log 2>&1; echo "fuzz_hot rc $?"; tail -1 gpurun_out/${tag}_fuzz_hot.log
timeout -k 10 420 python tools/fuzz_paths.py 12300 14300 > gpurun_out/${tag}_fuzz_paths.log 2>&1; echo "fuzz_paths rc $?"; tail -1 gpurun_out/${tag}_fuzz_paths.log
timeout -k 10 300 python tools/fuzz_paths.py 106200 107200 > gpurun_out/${tag}_fuzz_paths_long.log 2>&1; echo "fuzz_paths long rc $?"; tail -1 gpurun_out/${tag}_fuzz_paths_long.log
timeout -k 10 420 python tools/fuzz_int.py 300 450 > gpurun_out/${tag}_fuzz_int2.log 2>&1; echo "fuzz_int rc $?"; tail -1 gpurun_out/${tag}_fuzz_int2.log
