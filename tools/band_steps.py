#!/usr/bin/env python3
"""Launch time of the band kernels against the number of components swept (fixed cost of a launch / block vs cost per step)."""
import ctypes, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triangular_transport_toolbox_amd import _capi, specs
from triangular_transport_toolbox_amd.transport_map import transport_map
N = 1000000
cfg = specs.config('C5')
X = cfg['sampler'](N)
tm = transport_map(X=X, monotone=cfg['monotone'], nonmonotone=cfg['nonmonotone'], verbose=False, **cfg['kwargs'])
npz = np.load(os.path.join(ROOT, 'tests', 'golden', 'c5_sep.npz'))
tm.coeffs_mon = [npz['coeffs_mon_%d' % k] for k in range(tm.D)]
tm.coeffs_nonmon = [npz['coeffs_nonmon_%d' % k] for k in range(tm.D)]
lib = _capi.load()
coef = tm._pack_coeffs()
Xs = tm._Xs
Z = tm._cols(tm.D, N)


def timed(fn, n=100):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t_end = time.time() + 0.5
    while time.time() < t_end:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for k1 in (2, 10, 20, 30, 40):
    def f():
        _capi.check(lib.ttm_forward(tm._pp, tm._ptr(coef), tm._ptr(coef._ttm_fold), tm._ptr(Xs), Xs.shape[1], N, 0, k1, tm._ptr(Z), Z.shape[1],
                                    None, None, None, tm._stream()))
    ms = timed(f)
    print('forward k1=%2d: %.4f ms  (%.2f us per step)' % (k1, ms, 1e3 * ms / k1))
