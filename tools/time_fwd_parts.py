#!/usr/bin/env python3
"""Forward launch of the integrated spiral map (C2a), component by component: where the 0.24 ms go."""
import os, sys, time, ctypes
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from triangular_transport_toolbox_amd import _capi  # noqa: E402

tm, X, cfg = bench.build_map(sys.argv[1] if len(sys.argv) > 1 else 'C2a', 0)
coef = tm._current(None)
N = tm._N
Z = tm._cols(tm.D, N)
print('family', tm._cm.family, 'rectifier', tm._prog.rectifier, 'complex', list(tm._cm.complex))
for k0, k1 in [(0, tm.D)] + [(k, k + 1) for k in range(tm.D)]:
    def run():
        _capi.check(tm._lib.ttm_forward(tm._pp, tm._ptr(coef), tm._ptr(coef._ttm_fold), tm._ptr(tm._Xs), tm._Xs.shape[1], N, k0, k1,
                                        tm._ptr(Z), Z.shape[1], None, None, None, tm._stream()))
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        run()
    torch.cuda.synchronize()
    print('components [%d, %d): %.4f ms  kernel %s' % (k0, k1, 1e3 * (time.perf_counter() - t0) / 200, tm._lib.ttm_last_kernel().decode()), flush=True)
