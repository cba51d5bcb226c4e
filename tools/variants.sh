#!/bin/bash
# build libttm.so variants on the GPU box and bench each (tuning aid)
cd $GRAFT_REPO_ROOT
for v in "-DTTM_FWD_ETAB(NS)=(NS==4)" "-DTTM_FWD_ETAB(NS)=true"; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -DNDEBUG "$v" -o triangular_transport_toolbox_amd/libttm.so triangular_transport_toolbox_amd/csrc/ttm_kernels.hip 2>/dev/null
  echo "== variant [$v]"
  for ns in 2 4; do
  TTM_HL_NS=$ns timeout -k 10 200 python bench.py --no-cpu-baseline --no-optimize --workload C5 > gpurun_out/b.json 2> gpurun_out/b.err
  python -c "
import json
d=json.load(open('gpurun_out/b.json'))
print('ns $ns:', round(d['forward_ms'],4), round(d['pullback_fused_ms'],4))
"; done
done
