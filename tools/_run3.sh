set -e
python tools/int_bench.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_int_dense.py tests/test_transport_map.py tests/test_native_bfgs.py tests/test_newton_inverse.py tests/test_random_maps.py tests/test_kernels.py -m gpu -x -q > gpurun_out/r5_t3.log 2>&1 || { tail -40 gpurun_out/r5_t3.log; exit 1; }
tail -3 gpurun_out/r5_t3.log
python -m pytest tests/test_full_size.py -m gpu -x -q -k "c5int or c3int or c2a" > gpurun_out/r5_t3b.log 2>&1 || { tail -40 gpurun_out/r5_t3b.log; exit 1; }
tail -3 gpurun_out/r5_t3b.log
