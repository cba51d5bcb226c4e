#!/bin/bash
# A/B of two builds inside one gpurun call: default library against libttm_unp.so (TTM_RT_UNPAIRED)
out=gpurun_out/pair; mkdir -p $out
for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-optimize --no-other-configs --steps 100 > $out/p$rep.json 2> $out/p$rep.err
  TTM_BUILD_LIB=$PWD/triangular_transport_toolbox_amd/libttm_unp.so TTM_BUILD_FLAGS="-DTTM_RT_UNPAIRED" python bench.py --no-cpu-baseline --no-optimize --no-other-configs --steps 100 > $out/u$rep.json 2> $out/u$rep.err
done
python - <<'PY'
import json
for n in ('p1','u1','p2','u2'):
    j=json.load(open('gpurun_out/pair/%s.json'%n)); print(n, 'fwd %.4f inv %.4f step %.4f err %.2e' % (j['forward_ms'], j['inverse_ms'], j['ms_per_step'], j['roundtrip_max_abs_err']))
PY
