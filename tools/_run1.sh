set -e
python -m pytest tests/test_full_size.py -m gpu -x -q -k "c5int or c3int or block_map or c2a" > gpurun_out/r5_t1.log 2>&1 || { tail -40 gpurun_out/r5_t1.log; exit 1; }
tail -5 gpurun_out/r5_t1.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench1.json 2> gpurun_out/r5_bench1.err || { tail -30 gpurun_out/r5_bench1.err; exit 1; }
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5_bench1.json') if l.startswith('{"metric"')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_timing'])
print(json.dumps(d.get('objective_roofline'), indent=1)[:3000])
for k,v in d.get('other_configs',{}).items():
    print(k, json.dumps(v.get('objective_roofline'))[:1500], v.get('optimize_s'), v.get('optimize_full_N_s'), v.get('ms_per_step'), v.get('max_rel_err_vs_oracle_step'), v.get('cpu_step_s'))
print(d.get('other_configs_error'), d.get('other_configs_C4_block_map_error'))
PY
