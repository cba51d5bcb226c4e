set -e
python -m pytest tests/test_int_dense.py tests/test_transport_map.py tests/test_native_bfgs.py tests/test_newton_inverse.py tests/test_random_maps.py -m gpu -x -q > gpurun_out/r5_t2.log 2>&1 || { tail -40 gpurun_out/r5_t2.log; exit 1; }
tail -3 gpurun_out/r5_t2.log
python -m pytest tests/test_full_size.py -m gpu -x -q -k "c5int or c3int or c2a" > gpurun_out/r5_t2b.log 2>&1 || { tail -40 gpurun_out/r5_t2b.log; exit 1; }
tail -3 gpurun_out/r5_t2b.log
bash tools/fp64_counts.sh > gpurun_out/r5_fp64.log 2>&1 || { tail -20 gpurun_out/r5_fp64.log; exit 1; }
tail -30 gpurun_out/r5_fp64.log
for W in C2a C3int C5int; do python bench.py --workload $W --steps 3 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline --no-other-configs --no-api > gpurun_out/r5_b_$W.json 2>gpurun_out/r5_b_$W.err; python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r5_b_$W.json') if l.startswith('{"metric"')][-1])
print('$W', json.dumps(d.get('objective_roofline'))[:900], d.get('optimize_s'))
PY
done
