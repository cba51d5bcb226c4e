#!/usr/bin/env python3
"""ttm_objective_sep_cached: device time per launch (back to back) against the launch + synchronise round trip."""
import os, sys, time, ctypes
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from triangular_transport_toolbox_amd import _capi  # noqa: E402

for wl, n in (('C5', 1000000), ('C5', 100000), ('C3', 500000)):
    tm, X, cfg = bench.build_map(wl, 0, n)
    k = tm.D - 1
    tm._sep_cache_begin(k)
    dpsi = tm._sep_cache[1]
    m = int(tm._cm.n_mon[k])
    c = np.full(m, 0.3)
    work = tm._workspace(tm._lib.ttm_reduce_work_size(1 + m))
    args = (tm._ptr(dpsi), dpsi.shape[1], tm._N, m, ctypes.c_void_p(c.ctypes.data), float(tm.delta), tm._ptr(work),
            ctypes.c_void_p(tm._obj_cnt.data_ptr()), ctypes.c_void_p(tm._obj_out.data_ptr()), tm._stream())
    fn = tm._lib.ttm_objective_sep_cached
    for _ in range(200):
        fn(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        fn(*args)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(500):
        fn(*args)
        torch.cuda.current_stream().synchronize()
    t2 = time.perf_counter()
    print('%s N=%d m=%d: back to back %.1f us per launch, launch + synchronise %.1f us' % (wl, n, m, 1e6 * (t1 - t0) / 2000, 1e6 * (t2 - t1) / 500), flush=True)
    # the generic separable objective (derivative basis recomputed from x_k per evaluation, ttm_objective_host)
    cfull = np.concatenate((np.zeros(int(tm._cm.n_nm[k])), c))
    args2 = (tm._pp, int(k), ctypes.c_void_p(cfull.ctypes.data), tm._ptr(tm._Xs), tm._Xs.shape[1], tm._N, tm._ptr(work),
             ctypes.c_void_p(tm._obj_cnt.data_ptr()), ctypes.c_void_p(tm._obj_out.data_ptr()), tm._stream())
    fn2 = tm._lib.ttm_objective_host
    for _ in range(50):
        fn2(*args2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        fn2(*args2)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print('   generic objective (basis recomputed): back to back %.1f us per launch' % (1e6 * (t1 - t0) / 500), flush=True)
