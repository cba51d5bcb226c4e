#!/usr/bin/env python3
"""Forward-map launch time of the dense integrated kernel against the quadrature order (C2a / C5int shapes): T(Q) = a + b Q
separates the per-sample generic phase (weights through the term tables) from the node loop."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
from triangular_transport_toolbox_amd import specs
from triangular_transport_toolbox_amd.transport_map import transport_map

def ev_ms(fn, n=10):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))

for name in ('C2a', 'C5int'):
    cfgname, N, fixture, _ = bench.WORKLOADS[name]
    cfg = specs.config(cfgname)
    X = cfg['sampler'](N, seed=0)
    res = {}
    for Q in (5, 10, 25, 50):
        kw = dict(cfg['kwargs']); kw['quadrature_input'] = {'order': Q}
        tm = transport_map(X=X, monotone=cfg['monotone'], nonmonotone=cfg['nonmonotone'], verbose=False, **kw)
        tm.coeffs_mon, tm.coeffs_nonmon = bench.load_coeffs(fixture, tm.D)
        coef = tm._pack_coeffs()
        Z = tm._cols(tm.D, N)
        res[Q] = ev_ms(lambda: tm.forward_device(tm._Xs, N, coef=coef, Z=Z))
        del tm
    b = (res[50] - res[10]) / 40.0
    a = res[10] - 10 * b
    print(name, {q: round(v, 4) for q, v in res.items()}, 'per node %.2f us, generic phase %.1f us (%.0f %% of Q = 25)' % (1e3 * b, 1e3 * a, 100 * a / res[25]))
