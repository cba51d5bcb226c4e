#!/bin/bash
# build a tuning variant of the library: tools/build_variant.sh NAME "-DFLAG ..."  -> tools/variants_lib/libttm_NAME.so
name=$1; flags=$2
TTM_BUILD_LIB=$PWD/tools/variants_lib/libttm_$name.so TTM_BAND_FLAGS="$flags" TTM_INT_FLAGS="$INT_FLAGS" python -c "
from triangular_transport_toolbox_amd import build; print(build.build_lib())"
