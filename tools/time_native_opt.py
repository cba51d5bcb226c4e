#!/usr/bin/env python3
"""optimize() of C5 / C3: the library's own L-BFGS-B loop against SciPy's loop driven from Python (same device reductions)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from triangular_transport_toolbox_amd.transport_map import transport_map  # noqa: E402

for wl in (sys.argv[1:] or ['C5', 'C3']):
    tm, X, cfg = bench.build_map(wl, 0)
    for native in (True, False, True, False):
        transport_map.native_optimizer = native
        for k in range(tm.D):
            tm.coeffs_mon[k] = tm.coeffs_mon[k] * 0 + tm.coeffs_init
            tm.coeffs_nonmon[k] = tm.coeffs_nonmon[k] * 0 + tm.coeffs_init
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tm.optimize()
        torch.cuda.synchronize()
        print(wl, 'native' if native else 'scipy ', '%.4f s' % (time.perf_counter() - t0), 'J = %.12f' % tm.objective_total, flush=True)
    # where the time goes (native): setup vs loop
    import cProfile, pstats
    transport_map.native_optimizer = True
    pr = cProfile.Profile(); pr.enable(); tm.optimize(); torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
