#!/usr/bin/env python3
"""Forward map, bisection and Newton root searches of the integrated-rectifier workloads through the X programs (csrc/ttm_xprog.h)
against the kernels that walk the term tables per sample (option int_xprog = 0): ms per launch by HIP events, same process.
    python tools/int_bench.py [C2a C3int C5int]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402

names = [a for a in sys.argv[1:] if not a.startswith('--')] or ['C2a', 'C3int', 'C5int']
for name in names:
    tm, X, cfg = bench.build_map(name, 0)
    N, D, d = tm._N, tm.D, tm._cm.d_cols
    coef = tm._pack_coeffs()
    Xs, Z, Xinv = tm._Xs, tm._cols(D, N), tm._cols(d, N, zero=True)
    res = {}
    for mode in (1, 0):
        tm._lib.ttm_set_option(b'int_xprog', mode)
        f = lambda: tm.forward_device(Xs, N, coef=coef, Z=Z)          # noqa: E731
        i = lambda: tm.inverse_device(Z, N, coef=coef, X=Xinv)        # noqa: E731
        f(); kf = bench._last_kernel(tm)
        tf = bench._events_ms(torch, f, 10)
        Zc = Z[:, :N].clone()
        i(); ki = bench._last_kernel(tm)
        ti = bench._events_ms(torch, i, 3, warm=1)
        Xc = Xinv[:, :N].clone()
        tm.root_finder = 'newton'
        tn = bench._events_ms(torch, i, 3, warm=1)
        Xn = Xinv[:, :N].clone()
        tm.root_finder = 'reference'
        res[mode] = (kf, tf, ki, ti, tn, Zc, Xc, Xn)
    tm._lib.ttm_reset_options()
    a, b = res[1], res[0]
    print('%-6s N=%d  forward %s %.4f / %s %.4f ms (x%.2f)  bisect %.3f / %.3f ms (x%.2f)  newton %.3f / %.3f ms (x%.2f)  max diff Z %.1e  X %.1e  Xn %.1e' %
          (name, N, a[0], a[1], b[0], b[1], b[1] / a[1], a[3], b[3], b[3] / a[3], a[4], b[4], b[4] / a[4],
           float((a[5] - b[5]).abs().max()), float((a[6] - b[6]).abs().max()), float((a[7] - b[7]).abs().max())), flush=True)
