#!/usr/bin/env python3
"""What a new coefficient vector costs in front of the C5 step (bench.py's uncached_* keys), fused setup launches against
the separate ones (options fold_fused / table_fused = 0)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch

tm, X, cfg = bench.build_map('C5', 0)
N, D, d = tm._N, tm.D, tm._cm.d_cols
coef = tm._pack_coeffs()
Xs, Z, Xinv = tm._Xs, tm._cols(D, N), tm._cols(d, N, zero=True)
def step():
    tm.forward_device(Xs, N, coef=coef, Z=Z); tm.inverse_device(Z, N, coef=coef, X=Xinv)
prev = [None]
def step_unc():
    tm._pack_memo = None
    c = tm._pack_coeffs()
    tm.forward_device(Xs, N, coef=c, Z=Z); tm.inverse_device(Z, N, coef=c, X=Xinv)
    if tm.deferred_checks:
        if prev[0] is not None and not tm.validate(prev[0]):
            raise RuntimeError('deferred check failed')
        prev[0] = c
def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
for _ in range(300): step()
out = {}
for fused in (1, 0, 1, 0):
    tm._lib.ttm_reset_options()
    if not fused:
        tm._lib.ttm_set_option(b'fold_fused', 0); tm._lib.ttm_set_option(b'table_fused', 0)
    for _ in range(100): step()
    base = timed(step, 100)
    tm.deferred_checks = False
    for _ in range(5): step_unc()
    eager = timed(step_unc, 50)
    tm.deferred_checks = True; prev[0] = None
    for _ in range(5): step_unc()
    deferred = timed(step_unc, 50)
    tm.validate(prev[0]); tm.deferred_checks = False
    print('fused %d: step %.4f ms  uncached %.4f ms (setup %.1f us)  deferred checks %.4f ms (setup %.1f us)' %
          (fused, base, eager, 1e3 * (eager - base), deferred, 1e3 * (deferred - base)), flush=True)
