set -e
python -m pytest tests/test_band.py tests/test_random_few.py -m gpu -x -q > gpurun_out/r5_t7.log 2>&1 || { tail -60 gpurun_out/r5_t7.log; exit 1; }
tail -2 gpurun_out/r5_t7.log
python -m pytest tests/test_transport_map.py tests/test_full_size.py -m gpu -x -q -k "ents or entf or block_map or c3_full or c2b or example03" > gpurun_out/r5_t7b.log 2>&1 || { tail -60 gpurun_out/r5_t7b.log; exit 1; }
tail -2 gpurun_out/r5_t7b.log
python - <<'PY'
import sys, torch
sys.path.insert(0,'.')
import bench
r = bench.ents_block_config(torch, steps=20)
print({k:v for k,v in r.items() if k not in ('workload','cpu_step')})
PY
