#!/bin/bash
# tuning variant of csrc/ttm_int.hip (mini build: two classes): tools/xobj_variant.sh NAME "-DXOBJ_NODES=4 ..." -> tools/variants_lib/libttm_NAME.so
name=$1; flags=$2
mkdir -p tools/variants_lib
echo "-DTTM_INT_MINI $flags" > tools/variants_lib/$name.flags
TTM_BUILD_LIB=$PWD/tools/variants_lib/libttm_$name.so TTM_INT_FLAGS="-DTTM_INT_MINI $flags" python -c "
from triangular_transport_toolbox_amd import build; print(build.build_lib())"
