for ns in 2 4; do for cfg in "3 2" "2 1" "2 2" "3 1"; do set -- $cfg; TTM_HL_NS=$ns TTM_U_XLEAD=$1 TTM_U_TLEAD=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --no-optimize > gpurun_out/b.json 2> gpurun_out/b.err; python -c "
import json
d=json.load(open('gpurun_out/b.json'))
print('ns $ns xlead $1 tlead $2', round(d['forward_ms'],4), round(d['inverse_ms'],4), round(d['pullback_fused_ms'],4), round(d['ms_per_step'],4), d['roundtrip_max_abs_err'])
"; done; done
