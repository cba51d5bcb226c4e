#!/usr/bin/env python3
"""optimize() of the integrated-rectifier spiral map (C2a) at N = 1e5: wall time and evaluation counts."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
tm, X, cfg = bench.build_map('C2a', 0, n_override=N)
print('N', tm._N, flush=True)
for threads in [int(t) for t in os.environ.get('TTM_OPT_THREADS', '1,2').split(',')]:
    tm.optimizer_threads = threads
    best = 1e9
    for rep in range(3):
        for k in range(tm.D):
            tm.coeffs_mon[k] = tm.coeffs_mon[k] * 0 + tm.coeffs_init
            tm.coeffs_nonmon[k] = tm.coeffs_nonmon[k] * 0 + tm.coeffs_init
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tm.optimize()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print('C2a threads', threads, 'optimize %.4f s' % best, flush=True)
