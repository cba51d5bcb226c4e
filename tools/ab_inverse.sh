#!/bin/bash
# A/B of the table-inverse kernels inside ONE gpurun call (boxes differ by +-2 %): C5 bench lines under different
# tuning knobs (read once per process by libttm.so).  usage: tools/ab_inverse.sh OUTDIR "ENV1" "ENV2" ...
out=$1; shift
mkdir -p "$out"
i=0
for cfg in "$@"; do
    i=$((i+1))
    env $cfg python bench.py --no-cpu-baseline --no-optimize --no-other-configs --steps 100 > "$out/ab_$i.json" 2> "$out/ab_$i.err" || { echo "variant '$cfg' failed"; tail -5 "$out/ab_$i.err"; }
    python - "$out/ab_$i.json" "$cfg" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    print('%-40s fwd %.4f ms  inv %.4f ms  step %.4f ms  err %.2e  kernel %s' % (sys.argv[2] or '(default)', j['forward_ms'], j['inverse_ms'], j['ms_per_step'], j['roundtrip_max_abs_err'], j['roofline']['inverse_kernel']))
except Exception as e:
    print(sys.argv[2], 'no result', e)
PY
done
