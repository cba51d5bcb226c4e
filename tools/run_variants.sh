#!/bin/bash
# time the forward band kernel of every tuning variant in tools/variants_lib (GPU box, one call: same card)
# usage: tools/run_variants.sh "-DFLAGS of variant A" NAME_A "-DFLAGS of B" NAME_B ...   (base library first)
export TTM_BAND_CHECK_ONLY=${ONLY:-fwd}
echo "== base"; python tools/band_check.py 1000000 --no-oracle 2>&1 | grep "k_band\|vs k_"
while [ $# -gt 1 ]; do
  flags=$1; name=$2; shift 2
  echo "== $name ($flags)"
  TTM_BUILD_LIB=$PWD/tools/variants_lib/libttm_$name.so TTM_BAND_FLAGS="$flags" python tools/band_check.py 1000000 --no-oracle 2>&1 | grep "k_band\|vs k_\|rror"
done
