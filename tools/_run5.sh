set -e
python -m pytest tests/test_transport_map.py tests/test_native_lbfgsb.py tests/test_kernels.py -m gpu -x -q > gpurun_out/r5_t5.log 2>&1 || { tail -40 gpurun_out/r5_t5.log; exit 1; }
tail -2 gpurun_out/r5_t5.log
python -m pytest tests/test_full_size.py -m gpu -x -q -k "c4 or c3_full" > gpurun_out/r5_t5b.log 2>&1 || { tail -40 gpurun_out/r5_t5b.log; exit 1; }
tail -2 gpurun_out/r5_t5b.log
python - <<'PY'
import numpy as np, torch, time, sys
sys.path.insert(0,'.')
import bench
r = bench.entf_config(torch, cycles=600)
print('C4', r['ms_per_cycle'])
tm, X, cfg = bench.build_map('C5', 0)
def topt():
    for k in range(tm.D):
        tm.coeffs_mon[k] = tm.coeffs_mon[k]*0 + tm.coeffs_init; tm.coeffs_nonmon[k] = tm.coeffs_nonmon[k]*0 + tm.coeffs_init
    torch.cuda.synchronize(); t0=time.perf_counter(); tm.optimize(); torch.cuda.synchronize(); return time.perf_counter()-t0
topt(); print('C5 optimize', topt(), topt())
print(json.dumps(bench.objective_roofline(torch, tm, 'C5')['objective_separable'])[:400]) if False else None
o = bench.objective_roofline(torch, tm, 'C5')['objective_separable']; print('sep eval ms', o['ms_per_evaluation'], o['frac'])
PY
