#!/bin/bash
# build libttm.so variants on the GPU box and bench each (tuning aid)
#   [ENVS="TTM_HL_NS=2 TTM_HL_NS=4"] bash tools/variants.sh "<flags of variant 1>" "<flags of variant 2>" ...
# (an empty string = the default build; every variant is benched once per entry of ENVS)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -DNDEBUG $v -o triangular_transport_toolbox_amd/libttm.so triangular_transport_toolbox_amd/csrc/ttm_kernels.hip 2>gpurun_out/variant_build.err || { echo "build failed [$v]"; continue; }
  for e in ${ENVS:-_=_}; do
  echo "== variant [$v] env [$e]"
  env $e timeout -k 10 200 python bench.py --no-cpu-baseline --no-optimize --no-other-configs --workload C5 --prewarm-seconds ${PREWARM:-1.0} > gpurun_out/b.json 2> gpurun_out/b.err
  python -c "
import json
d=json.load(open('gpurun_out/b.json'))
print('fwd', round(d['forward_ms'],4), 'inv', round(d['inverse_ms'],4), 'pullback', round(d['pullback_fused_ms'],4), 'step', round(d['ms_per_step'],4), 'rt_err', d['roundtrip_max_abs_err'])
"
  done
done
