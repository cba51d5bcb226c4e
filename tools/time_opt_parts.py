#!/usr/bin/env python3
"""Where optimize() of a separable map spends its time: Gram / reduced problem, cached basis, optimiser loop."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

for wl in (sys.argv[1:] or ['C5']):
    tm, X, cfg = bench.build_map(wl, 0)
    for rep in range(2):
        for k in range(tm.D):
            tm.coeffs_mon[k] = tm.coeffs_mon[k] * 0 + tm.coeffs_init
            tm.coeffs_nonmon[k] = tm.coeffs_nonmon[k] * 0 + tm.coeffs_init
        t = dict(setup=0.0, cache=0.0, loop=0.0)
        nfev = 0
        sync = torch.cuda.synchronize
        for k in range(tm.D):
            sync(); t0 = time.perf_counter()
            A, solve = tm.separable_setup(k)
            sync(); t1 = time.perf_counter()
            tm._sep_cache_begin(k)
            sync(); t2 = time.perf_counter()
            bounds = [[tm.optimization_constraints_lb[k][i], tm.optimization_constraints_ub[k][i]] for i in range(len(tm.optimization_constraints_lb[k]))]
            opt = tm._optimize_separable_native(A, k, np.asarray(tm.coeffs_mon[k], dtype=float), bounds)
            sync(); t3 = time.perf_counter()
            tm._sep_cache_end()
            t['setup'] += t1 - t0; t['cache'] += t2 - t1; t['loop'] += t3 - t2
            nfev += opt.nfev
        print(wl, {a: round(b, 4) for a, b in t.items()}, 'evaluations', nfev, 'us per evaluation %.1f' % (1e6 * t['loop'] / nfev), flush=True)
