#!/usr/bin/env python3
"""What a changed coefficient vector costs before the lookups can run (C5): packing, fold + U-form build, table build +
index + the sortedness read-back."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tm, X, cfg = bench.build_map('C5', 0)
N = tm._N
Z = tm._cols(tm.D, N)
Xi = tm._cols(tm._cm.d_cols, N, zero=True)
sync = torch.cuda.synchronize


def t(fn, n=200):
    for _ in range(20):
        fn()
    sync(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    sync()
    return 1e6 * (time.perf_counter() - t0) / n


def pack():
    tm._pack_memo = None
    return tm._pack_coeffs()


coef = pack()
print('pack + fold (host packing, H2D, ttm_fold, U-form build): %.1f us' % t(pack))
host = np.concatenate([np.concatenate((tm.coeffs_nonmon[k], tm.coeffs_mon[k])) for k in range(tm.D)])
print('   of which host packing alone: %.1f us' % t(lambda: np.concatenate([np.concatenate((np.asarray(tm.coeffs_nonmon[k], dtype=float).ravel(), np.asarray(tm.coeffs_mon[k], dtype=float).ravel())) for k in range(tm.D)])))
print('   H2D of the packed vector: %.1f us' % t(lambda: tm._to_dev(host)))


def tables():
    c = pack()
    tm._inverse_table(c, 0, tm.D, Z, Xi, N)


def lookups():
    tm.forward_device(tm._Xs, N, coef=coef, Z=Z)
    tm.inverse_device(Z, N, coef=coef, X=Xi)


tl = t(lookups)
tu = t(lambda: (tm.forward_device(tm._Xs, N, coef=pack(), Z=Z), tm.inverse_device(Z, N, coef=tm._pack_memo[2], X=Xi)))
print('forward + inverse, cached: %.1f us; with a fresh coefficient vector every step: %.1f us (setup %.1f us)' % (tl, tu, tu - tl))
