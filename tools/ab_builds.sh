#!/bin/bash
# A/B of two builds of the library inside ONE gpurun call (boxes differ by +-5 %): the default libttm.so against another
# build given as $1 (built beforehand with TTM_BUILD_LIB=$1 [TTM_BUILD_FLAGS=...]; touch it after changing sources so
# that it is not rebuilt), flags in $2.   usage: tools/ab_builds.sh path/to/other.so ["-DFLAG ..."]
other=$1; flags=$2
out=gpurun_out/ab_builds; mkdir -p $out
for rep in 1 2 3; do
  python bench.py --no-cpu-baseline --no-optimize --no-other-configs --steps 100 > $out/a$rep.json 2> $out/a$rep.err
  TTM_BUILD_LIB=$other TTM_BUILD_FLAGS="$flags" python bench.py --no-cpu-baseline --no-optimize --no-other-configs --steps 100 > $out/b$rep.json 2> $out/b$rep.err
done
python - <<'PY'
import json
for n in ('a1','b1','a2','b2','a3','b3'):
    j=json.load(open('gpurun_out/ab_builds/%s.json'%n)); print(n, 'fwd %.4f inv %.4f step %.4f err %.2e' % (j['forward_ms'], j['inverse_ms'], j['ms_per_step'], j['roundtrip_max_abs_err']))
PY
