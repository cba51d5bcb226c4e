#!/bin/bash
# time the band inverse of tuning variants in tools/variants_lib inside one call, base library between them
#   usage: tools/ab_band_inv.sh "-DFLAGS of A" NAME_A "-DFLAGS of B" NAME_B ...   (built beforehand with tools/build_variant.sh NAME "FLAGS";
#   the flags are part of the build stamp: without them the variant would be rebuilt as the default library)
export TTM_BAND_CHECK_ONLY=inv
run() { python tools/band_check.py 1000000 --no-oracle 2>&1 | grep "inverse k_band\|round trip"; }
echo "== base"; run
while [ $# -gt 1 ]; do
  flags=$1; name=$2; shift 2
  echo "== $name ($flags)"
  TTM_BUILD_LIB=$PWD/tools/variants_lib/libttm_$name.so TTM_BAND_FLAGS="$flags" run
  echo "== base"; run
done
