for cfg in "1 1" "2 1" "3 1" "4 1" "2 2" "3 2" "4 2"; do set -- $cfg; TTM_U_XLEAD=$1 TTM_U_TLEAD=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --no-optimize > gpurun_out/b.json 2> gpurun_out/b.err; python -c "
import json
d=json.load(open('gpurun_out/b.json'))
print('xlead $1 tlead $2', round(d['forward_ms'],4), round(d['pullback_fused_ms'],4))
"; done
