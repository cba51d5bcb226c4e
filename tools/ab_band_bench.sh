#!/bin/bash
# C5 bench lines of tuning variants of csrc/ttm_band.hip inside one call, the default library between them
#   usage: tools/ab_band_bench.sh "-DFLAGS of A" NAME_A "-DFLAGS of B" NAME_B ...   (built beforehand: tools/build_variant.sh NAME "FLAGS")
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('%-28s fwd %.4f  inv %.4f  step %.4f ms  err %.2e  %s / %s' % (sys.argv[2], j['forward_ms'], j['inverse_ms'], j['ms_per_step'], j['roundtrip_max_abs_err'], j['roofline']['forward_kernel'], j['roofline']['inverse_kernel']))
except Exception as e:
    print(sys.argv[2], 'no result', e)
PY
}
ARGS="--no-cpu-baseline --no-optimize --no-other-configs --no-api --steps 100"
mkdir -p gpurun_out/abb
python bench.py $ARGS > gpurun_out/abb/base.json 2> gpurun_out/abb/base.err; line gpurun_out/abb/base.json base
while [ $# -gt 1 ]; do
  flags=$1; name=$2; shift 2
  TTM_BUILD_LIB=$PWD/tools/variants_lib/libttm_$name.so TTM_BAND_FLAGS="$flags" python bench.py $ARGS > gpurun_out/abb/$name.json 2> gpurun_out/abb/$name.err; line gpurun_out/abb/$name.json "$name"
  python bench.py $ARGS > gpurun_out/abb/base.json 2> gpurun_out/abb/base.err; line gpurun_out/abb/base.json base
done
