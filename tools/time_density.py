#!/usr/bin/env python3
"""Launch times of the density passes of banded maps (graph replay): fused S + log det + |S|^2, sum of squares only,
log-determinant only (k_band_logdet), for C5 / C2b / C3.   python tools/time_density.py [C5 C2b C3]"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402


def main():
    names = sys.argv[1:] or ['C5', 'C2b', 'C3']
    out = {}
    for name in names:
        tm, X, cfg = bench.build_map(name, 0)
        N, D, d = tm._N, tm.D, tm._cm.d_cols
        coef = tm._pack_coeffs()
        Xs, Z = tm._Xs, tm._cols(D, N)
        ld, ss = tm._empty(N), tm._empty(N)
        sigma = tm._to_dev(np.asarray(tm.X_std[:D], dtype=float))
        du = bench.d_used(tm)
        r = {}
        for key, fn, nbytes in (
                ('forward', lambda: tm.forward_device(Xs, N, coef=coef, Z=Z), 8.0 * N * (du + D)),
                ('fused_Z_logdet_sumsq', lambda: tm.forward_device(Xs, N, coef=coef, Z=Z, logdet=ld, sigma=sigma, sumsq=ss), 8.0 * N * (du + D + 2)),
                ('logdet_sumsq', lambda: tm.density_device(Xs, N, coef=coef, logdet=ld, sigma=sigma, sumsq=ss), 8.0 * N * (du + 2)),
                ('sumsq_only', lambda: tm.density_device(Xs, N, coef=coef, sumsq=ss), 8.0 * N * (du + 1)),
                ('logdet_only', lambda: tm.density_device(Xs, N, coef=coef, logdet=ld, sigma=sigma), 8.0 * N * (du + 1))):
            fn()
            kern = bench._last_kernel(tm)
            ms = bench.graph_ms(torch, fn)
            r[key] = {'kernel': kern, 'ms': ms, 'frac_of_8TBs': nbytes / (ms * 1e-3) / 8e12}
        out[name] = r
        print(name, json.dumps(r), flush=True)
        del tm, Xs, Z
    tag = os.environ.get('TTM_TAG', 'default')
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'time_density_%s.json' % tag), 'w'), indent=1)


if __name__ == '__main__':
    main()
