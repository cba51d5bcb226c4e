#!/usr/bin/env python3
"""Objective + gradient evaluation of the integrated-rectifier workloads (k_int_objective): ms per evaluation by HIP events over
passes through all components, and the sums against the kernel they replace (option int_xprog = 0: k_int_objective_walk).
    python tools/obj_bench.py [C2a C3int C5int] [--passes 5]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402

names = [a for a in sys.argv[1:] if not a.startswith('--')] or ['C2a', 'C3int', 'C5int']
passes = int(sys.argv[sys.argv.index('--passes') + 1]) if '--passes' in sys.argv else 5
for name in names:
    tm, X, cfg = bench.build_map(name, 0)
    D = tm.D
    cs = [np.ascontiguousarray(np.concatenate((tm.coeffs_nonmon[k], tm.coeffs_mon[k]))) for k in range(D)]

    def one_pass():
        for k in range(D):
            tm._objective_launch(k, cs[k])
    res = {}
    for mode in (1, 0):
        tm._lib.ttm_set_option(b'int_xprog', mode)
        one_pass()
        kern = bench._last_kernel(tm)
        ms = bench._events_ms(torch, one_pass, passes)
        sums = [tm._device_sums(k, cs[k]) for k in (0, D // 2, D - 1)]
        res[mode] = (kern, ms, sums)
    tm._lib.ttm_reset_options()
    err = max(float(np.max(np.abs(a - b) / (np.abs(b) + 1.0))) for a, b in zip(res[1][2], res[0][2]))
    print('%-6s N=%d D=%d  %s %.4f ms/eval   %s %.4f ms/eval   ratio %.2f   max rel diff of the sums %.2e' %
          (name, tm._N, D, res[1][0], res[1][1] / D, res[0][0], res[0][1] / D, res[0][1] / res[1][1], err), flush=True)
