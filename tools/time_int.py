#!/usr/bin/env python3
"""Launch times of the integrated-rectifier kernels (forward map, bisection / Newton root search, objective + gradient) of
C2a / C3int / C5int, dense monomial-form kernels (csrc/ttm_int.hip) against the generic ones (option int_dense = 0), with the
largest difference between the two paths' results.      python tools/time_int.py [C2a C3int C5int]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402


def ev_ms(fn, n, warm=2):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


def main():
    names = sys.argv[1:] or ['C2a', 'C3int', 'C5int']
    out = {}
    for name in names:
        tm, X, cfg = bench.build_map(name, 0)
        N, D, d = tm._N, tm.D, tm._cm.d_cols
        coef = tm._pack_coeffs()
        Xs, Z, Xinv = tm._Xs, tm._cols(D, N), tm._cols(d, N, zero=True)
        res = {}
        keep = {}
        for mode in (0, 12, 16, -1):
            tm._lib.ttm_reset_options()
            if mode == 0:
                tm._lib.ttm_set_option(b'int_dense', 0)
            elif mode > 0:
                tm._lib.ttm_set_option(b'int_wgs', mode)
            r = {}
            r['forward_ms'] = ev_ms(lambda: tm.forward_device(Xs, N, coef=coef, Z=Z), 10)
            r['forward_kernel'] = bench._last_kernel(tm)
            keep[(mode, 'Z')] = Z.clone()
            tm.root_finder = 'reference'
            r['bisect_ms'] = ev_ms(lambda: tm.inverse_device(Z, N, coef=coef, X=Xinv), 3, warm=1)
            r['bisect_kernel'] = bench._last_kernel(tm)
            keep[(mode, 'Xb')] = Xinv.clone()
            tm.root_finder = 'newton'
            r['newton_ms'] = ev_ms(lambda: tm.inverse_device(Z, N, coef=coef, X=Xinv), 3, warm=1)
            tm.root_finder = 'reference'
            # objective + gradient of the last component on the resident ensemble
            k = D - 1
            div = len(tm.coeffs_nonmon[k])
            c = np.concatenate((tm.coeffs_nonmon[k], tm.coeffs_mon[k]))
            nrep = 20
            G = tm.objective_function_jacobian(c, k, div)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(nrep):
                tm.objective_function_jacobian(c * (1.0 + 1e-9 * (i + 1)), k, div)
            torch.cuda.synchronize()
            r['objective_call_ms'] = 1e3 * (time.perf_counter() - t0) / nrep
            keep[(mode, 'G')] = np.array(G)
            keep[(mode, 'J')] = tm.objective_function(c, k, div)
            res['generic' if mode == 0 else ('dense_wgs%d' % mode if mode > 0 else 'dense')] = r
        tm._lib.ttm_reset_options()
        res['max_abs_diff'] = {'Z': float((keep[(0, 'Z')] - keep[(-1, 'Z')]).abs().max().item()),
                               'X_bisect': float((keep[(0, 'Xb')] - keep[(-1, 'Xb')])[:, 1:N].abs().max().item()),
                               'J': abs(keep[(0, 'J')] - keep[(-1, 'J')]), 'gradJ': float(np.max(np.abs(keep[(0, 'G')] - keep[(-1, 'G')])))}
        res['N'], res['D'] = N, D
        out[name] = res
        print(name, json.dumps(res), flush=True)
        del tm, Xs, Z, Xinv
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'time_int.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
