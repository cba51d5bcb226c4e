set -e
python -m pytest tests/test_int_dense.py -m gpu -x -q > gpurun_out/r5_t4.log 2>&1 || { tail -40 gpurun_out/r5_t4.log; exit 1; }
tail -2 gpurun_out/r5_t4.log
bash tools/fp64_counts.sh > gpurun_out/r5_fp64.log 2>&1 || { tail -20 gpurun_out/r5_fp64.log; exit 1; }
grep -A5 "^C2a\|^C3int\|^C5int" gpurun_out/r5_fp64.log | grep -v "^--"
python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench2.json 2> gpurun_out/r5_bench2.err || { tail -30 gpurun_out/r5_bench2.err; exit 1; }
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5_bench2.json') if l.startswith('{"metric"')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k,v in d.get('other_configs',{}).items():
    o=(v.get('objective_roofline') or {}).get('objective') or {}
    f=v.get('fp64') or {}
    print(k, 'fwd %.4f inv %.4f'%(v.get('forward_ms',0),v.get('inverse_ms',0)), 'obj frac', o.get('frac'), 'ms/eval', o.get('ms_per_evaluation'), 'fwd frac', (f.get('forward') or {}).get('frac'), 'inv frac', (f.get('inverse') or {}).get('frac'), 'opt', v.get('optimize_s'), v.get('optimize_full_N_s'), v.get('ms_per_cycle'), v.get('ms_per_step'))
PY
