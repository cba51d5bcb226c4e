#!/bin/bash
# build libttm.so variants on the GPU box and test + bench each (tuning aid)
cd $GRAFT_REPO_ROOT
for v in "-DTTM_HL_WAVES=8" "-DTTM_HL_WAVES=6" "-DTTM_HL_WAVES=5" ""; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -DNDEBUG $v -o triangular_transport_toolbox_amd/libttm.so triangular_transport_toolbox_amd/csrc/ttm_kernels.hip 2>/dev/null
  echo "== variant [$v]"
  for ns in 2 4; do
  TTM_HL_NS=$ns timeout -k 10 200 python bench.py --no-cpu-baseline --no-optimize > gpurun_out/b.json 2> gpurun_out/b.err
  python -c "
import json
d=json.load(open('gpurun_out/b.json'))
print('ns $ns:', round(d['forward_ms'],4), round(d['inverse_ms'],4), round(d['pullback_fused_ms'],4), round(d['ms_per_step'],4))
"; done
done
