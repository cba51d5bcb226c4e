mkdir -p gpurun_out
timeout -k 10 400 python tools/fuzz_gpu.py 700 1300 > gpurun_out/r05_fuzz_gpu_b.log 2>&1; tail -1 gpurun_out/r05_fuzz_gpu_b.log
timeout -k 10 400 python tools/fuzz_few.py 8000 12000 > gpurun_out/r05_fuzz_few_b.log 2>&1; tail -1 gpurun_out/r05_fuzz_few_b.log | cut -c1-80
timeout -k 10 400 python tools/fuzz_hot.py 500 800 > gpurun_out/r05_fuzz_hot_b.log 2>&1; tail -1 gpurun_out/r05_fuzz_hot_b.log
timeout -k 10 500 python tools/fuzz_paths.py 14300 18300 > gpurun_out/r05_fuzz_paths_b.log 2>&1; tail -1 gpurun_out/r05_fuzz_paths_b.log
timeout -k 10 400 python tools/fuzz_paths.py 107200 109200 > gpurun_out/r05_fuzz_paths_long_b.log 2>&1; tail -1 gpurun_out/r05_fuzz_paths_long_b.log
timeout -k 10 500 python tools/fuzz_int.py 450 750 > gpurun_out/r05_fuzz_int_b.log 2>&1; tail -1 gpurun_out/r05_fuzz_int_b.log
