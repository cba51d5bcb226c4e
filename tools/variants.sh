#!/bin/bash
# build libttm.so variants on the GPU box and test + bench each (tuning aid)
cd $GRAFT_REPO_ROOT
for v in "-DTTM_UL_CW=8" "-DTTM_UL_CW=6" "-DTTM_UL_CW=4"; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -DNDEBUG $v -o triangular_transport_toolbox_amd/libttm.so triangular_transport_toolbox_amd/csrc/ttm_kernels.hip 2>/dev/null
  echo "== variant [$v]"
  timeout -k 10 240 python -m pytest tests/test_kernels.py tests/test_uform.py -m gpu -x -q 2>&1 | tail -2
  for cfg in "3 2" "2 1" "2 2"; do set -- $cfg
  TTM_U_XLEAD=$1 TTM_U_TLEAD=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --no-optimize > gpurun_out/b.json 2> gpurun_out/b.err
  python -c "
import json
d=json.load(open('gpurun_out/b.json'))
print('xlead $1 tlead $2:', round(d['forward_ms'],4), round(d['inverse_ms'],4), round(d['pullback_fused_ms'],4), round(d['ms_per_step'],4))
"; done
done
