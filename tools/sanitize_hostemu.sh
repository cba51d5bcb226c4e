#!/bin/bash
# Sanitizer legs of the host side ("race detection"; CPU only - GPU AddressSanitizer is not available on the
# pool).  What is instrumented is the host test double of the C ABI (tests/hostemu/ttm_hostemu.cpp): the per-sample bodies of
# csrc/ttm_eval.h / ttm_dense.h / ttm_xprog.h / ttm_uform.h, the optimiser loops of csrc/ttm_optim.cpp (L-BFGS-B, BFGS, the batch
# drivers with their host threads) and csrc/ttm_handover.h (the hand-over that bounds ncclCommInitRank in csrc/ttm_comm.cpp).
#   1. AddressSanitizer + UndefinedBehaviorSanitizer build, the host-double tests of the CPU suite under it;
#   2. ThreadSanitizer build, the tests that drive ttm_optimize_separable_batch / ttm_optimize_integrated_batch with 8 host
#      threads (transport_map.optimizer_threads = 8, the default) under it;
#   3. ThreadSanitizer run of tools/sanitize/handover_tsan.cpp.
# usage: tools/sanitize_hostemu.sh [asan|tsan|handover|all]      (from the repo root; reports go to stdout, exit code 1 on a finding)
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/ttm_sanitize
mkdir -p $OUT
what=${1:-all}
rc=0
SRC=$R/tests/hostemu/ttm_hostemu.cpp
if [ "$what" = asan ] || [ "$what" = all ]; then
  g++ -O1 -g -std=c++17 -ffp-contract=off -fPIC -shared -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -o $OUT/libttm_hostemu_asan.so $SRC || exit 2
  echo "== ASan + UBSan: host-double tests"
  (cd $R && LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
     TTM_HOSTEMU_LIB=$OUT/libttm_hostemu_asan.so python -m pytest tests/test_transport_map.py tests/test_int_dense.py tests/test_uform.py tests/test_band.py \
     tests/test_random_maps.py tests/test_random_few.py tests/test_kernels.py tests/test_linearization.py tests/test_newton_inverse.py tests/test_native_bfgs.py \
     tests/test_native_lbfgsb.py tests/test_hostemu_vs_oracle.py -q -m "not gpu" -p no:cacheprovider > $OUT/asan.log 2>&1)
  tail -3 $OUT/asan.log
  if grep -q "ERROR: AddressSanitizer\|runtime error:" $OUT/asan.log; then echo "ASan / UBSan REPORTS:"; grep -n "ERROR: AddressSanitizer\|runtime error:" $OUT/asan.log | head -20; rc=1; else echo "ASan / UBSan: 0 reports"; fi
  grep -q " passed" $OUT/asan.log || { echo "the tests did not run: see $OUT/asan.log"; rc=1; }
fi
if [ "$what" = tsan ] || [ "$what" = all ]; then
  g++ -O1 -g -std=c++17 -ffp-contract=off -fPIC -shared -pthread -fsanitize=thread -fno-omit-frame-pointer -o $OUT/libttm_hostemu_tsan.so $SRC || exit 2
  echo "== TSan: batch optimisers with 8 host threads"
  (cd $R && LD_PRELOAD=$(gcc -print-file-name=libtsan.so) TSAN_OPTIONS=ignore_noninstrumented_modules=1:halt_on_error=0:report_signal_unsafe=0 \
     TTM_HOSTEMU_LIB=$OUT/libttm_hostemu_tsan.so python -m pytest tests/test_transport_map.py -q -m "not gpu" -k "optimize or entf or ents" -p no:cacheprovider > $OUT/tsan.log 2>&1)
  tail -3 $OUT/tsan.log
  if grep -q "WARNING: ThreadSanitizer" $OUT/tsan.log; then echo "TSan REPORTS:"; grep -n -A12 "WARNING: ThreadSanitizer" $OUT/tsan.log | head -60; rc=1; else echo "TSan: 0 reports"; fi
  grep -q " passed" $OUT/tsan.log || { echo "the tests did not run: see $OUT/tsan.log"; rc=1; }
fi
if [ "$what" = handover ] || [ "$what" = all ]; then
  echo "== TSan: ncclCommInitRank hand-over (csrc/ttm_handover.h)"
  g++ -O1 -g -std=c++17 -fsanitize=thread -pthread $R/tools/sanitize/handover_tsan.cpp -o $OUT/handover_tsan || exit 2
  $OUT/handover_tsan 600 > $OUT/handover.log 2>&1 || rc=1
  tail -2 $OUT/handover.log
  if grep -q "WARNING: ThreadSanitizer" $OUT/handover.log; then echo "TSan REPORTS (hand-over)"; rc=1; fi
fi
exit $rc
