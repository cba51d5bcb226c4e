#!/usr/bin/env python3
"""optimize() of C5 (40 components, N = 1e6), batched native path: wall clock + cProfile of the host side by internal time."""
import cProfile, os, pstats, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
tm, X, cfg = bench.build_map(os.environ.get('WL', 'C5'), 0)
tm.direct_objective = os.environ.get('DIRECT', '0') == '1'


def reset():
    for k in range(tm.D):
        tm.coeffs_mon[k] = tm.coeffs_mon[k] * 0 + tm.coeffs_init
        tm.coeffs_nonmon[k] = tm.coeffs_nonmon[k] * 0 + tm.coeffs_init


for rep in range(3):
    reset(); torch.cuda.synchronize(); t0 = time.perf_counter(); tm.optimize(); torch.cuda.synchronize()
    print('optimize: %.2f ms, evaluations %s' % (1e3 * (time.perf_counter() - t0), getattr(tm, 'last_optimize_evaluations', None)))
reset()
pr = cProfile.Profile(); pr.enable(); tm.optimize(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(6)
print('coefficients', [float(np.round(tm.coeffs_mon[k][0], 12)) for k in (0, 7, 39 if tm.D > 39 else tm.D - 1)])
