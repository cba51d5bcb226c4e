for nb in 1023 511 255 2047; do for cfg in "2 1" "3 1"; do set -- $cfg; TTM_INV_NB=$nb TTM_U_XLEAD=$1 TTM_U_TLEAD=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --no-optimize > gpurun_out/b.json 2> gpurun_out/b.err; python -c "
import json
d=json.load(open('gpurun_out/b.json'))
print('nb $nb xlead $1 tlead $2', round(d['forward_ms'],4), round(d['inverse_ms'],4), round(d['ms_per_step'],4), d['roundtrip_max_abs_err'])
"; done; done
