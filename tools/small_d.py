#!/usr/bin/env python3
"""Small-D maps (C2b, C3): launch time of the forward map, the table inverse and the log-det-only pullback pass against the
ensemble size - fixed cost of a launch (t at N -> 0) against the streaming rate (slope) - next to an elementwise
pass over the same bytes."""
import ctypes, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triangular_transport_toolbox_amd import _capi, specs
from triangular_transport_toolbox_amd.transport_map import transport_map

names = sys.argv[1:] or ['C2b', 'C3']
lib = _capi.load()
lib.ttm_last_kernel.restype = ctypes.c_char_p


def timed(fn, n=200):
    """us per launch, launches replayed from a captured graph (a launch here is shorter than the Python call that makes it)"""
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(25):
                fn()
    torch.cuda.synchronize()
    t_end = time.time() + 0.2
    while time.time() < t_end:
        g.replay()
        torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    reps = max(1, n // 25)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (25 * reps) * 1e3


for name in names:
    cfg = specs.config(name)
    gold = np.load(os.path.join(ROOT, 'tests', 'golden', name.lower() + '_sep.npz'))
    for N in [int(float(v)) for v in os.environ.get('SMALL_D_N', '125e3,250e3,500e3,1e6,2e6,4e6').split(',')]:
        X = cfg['sampler'](N)
        tm = transport_map(X=X, monotone=cfg['monotone'], nonmonotone=cfg['nonmonotone'], verbose=False, **cfg['kwargs'])
        tm.coeffs_mon = [gold['coeffs_mon_%d' % k] for k in range(tm.D)]
        tm.coeffs_nonmon = [gold['coeffs_nonmon_%d' % k] for k in range(tm.D)]
        D = tm.D
        Xs = tm._Xs
        Z = tm.forward_device(Xs, N)
        kf = lib.ttm_last_kernel().decode()
        tf = timed(lambda: tm.forward_device(Xs, N, Z=Z))
        Xi = tm.inverse_device(Z, N)
        ki = lib.ttm_last_kernel().decode()
        ti = timed(lambda: tm.inverse_device(Z, N, X=Xi))
        t0 = time.perf_counter()
        for _ in range(200):
            tm.inverse_device(Z, N, X=Xi)
        host = (time.perf_counter() - t0) / 200 * 1e6
        torch.cuda.synchronize()
        def pair():
            tm.forward_device(Xs, N, Z=Z)
            tm.inverse_device(Z, N, X=Xi)
        tp = timed(pair)
        ld = torch.empty(N + 2, dtype=torch.float64, device='cuda')
        sg = torch.ones(D, dtype=torch.float64, device='cuda')
        tm.density_device(Xs, N, logdet=ld, sigma=sg)
        kd = lib.ttm_last_kernel().decode()
        td = timed(lambda: tm.density_device(Xs, N, logdet=ld, sigma=sg))
        a = torch.empty(D * N, dtype=torch.float64, device='cuda'); b = torch.empty_like(a)
        tc = timed(lambda: torch.abs(a, out=b))
        mb = 16.0 * N * D / 1e6
        print('%s N=%8d  %5.1f MB | fwd %-16s %6.2f us %5.2f TB/s | inv %-16s %6.2f us %5.2f TB/s | pair %6.2f us %5.2f TB/s | logdet %s %6.2f us %5.2f TB/s | elementwise %6.2f us %5.2f TB/s | host call %5.1f us' %
              (name, N, mb, kf, tf, mb / tf, ki, ti, mb / ti, tp, 2 * mb / tp, kd, td, 8e-6 * N * (Xs.shape[0] + 1) / td, tc, mb / tc, host), flush=True)
        del tm, a, b
