#!/usr/bin/env python3
"""optimize() of a separable map through the batched native path: wall time against the number of optimiser threads,
and the split Gram pass / host setup + basis launches / optimiser loops."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

for wl in (sys.argv[1:] or ['C5']):
    tm, X, cfg = bench.build_map(wl, 0)
    sync = torch.cuda.synchronize

    def reset():
        for k in range(tm.D):
            tm.coeffs_mon[k] = tm.coeffs_mon[k] * 0 + tm.coeffs_init
            tm.coeffs_nonmon[k] = tm.coeffs_nonmon[k] * 0 + tm.coeffs_init
    for threads in [int(t) for t in os.environ.get('TTM_OPT_THREADS', '1,2,4,8,12,16,8').split(',')]:
        tm.optimizer_threads = threads
        best = 1e9
        for rep in range(3):
            reset()
            sync(); t0 = time.perf_counter()
            tm.optimize()
            sync(); best = min(best, time.perf_counter() - t0)
        print(wl, 'threads', threads, 'optimize %.4f s' % best, flush=True)
    # parts (threads = 8)
    K = list(range(tm.D))
    reset()
    sync(); t0 = time.perf_counter()
    grams = tm._gram_many(K)
    sync(); t1 = time.perf_counter()
    for k in K:
        tm.separable_setup(k, G=grams[k])
    t2 = time.perf_counter()
    print(wl, 'gram pass %.4f s, host reduced problems %.4f s' % (t1 - t0, t2 - t1), flush=True)
