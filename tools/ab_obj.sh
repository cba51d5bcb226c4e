#!/bin/bash
# A/B of tuning variants of the objective kernel inside ONE gpurun call: tools/ab_obj.sh "C2a C5int" base n4 n3 ...
w=$1; shift
for v in "$@"; do
  echo "== $v $(cat tools/variants_lib/$v.flags)"
  TTM_BUILD_LIB=$PWD/tools/variants_lib/libttm_$v.so TTM_INT_FLAGS="$(cat tools/variants_lib/$v.flags)" python tools/obj_bench.py $w 2>&1 | grep -v amdgpu.ids
done
