#!/usr/bin/env python3
"""Band (push-form) kernels against the kernels they replace and against the CPU oracle, with timings (GPU box).
   usage: python tools/band_check.py [N] [--no-oracle]"""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triangular_transport_toolbox_amd import _capi, specs                           # noqa: E402
from triangular_transport_toolbox_amd.transport_map import transport_map           # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1000000
cfg = specs.config('C5')
X = cfg['sampler'](N)
tm = transport_map(X=X, monotone=cfg['monotone'], nonmonotone=cfg['nonmonotone'], verbose=False, **cfg['kwargs'])
npz = np.load(os.path.join(ROOT, 'tests', 'golden', 'c5_sep.npz'))
tm.coeffs_mon = [npz['coeffs_mon_%d' % k] for k in range(tm.D)]
tm.coeffs_nonmon = [npz['coeffs_nonmon_%d' % k] for k in range(tm.D)]
lib = _capi.load()
lib.ttm_last_kernel.restype = ctypes.c_char_p
print('u_p_lag', tm._cm.u_p_lag, 'stride', tm._cm.u_p_stride, 'u_size', tm._cm.u_size)


def opt(name, v):
    assert lib.ttm_set_option(name.encode(), int(v)) == 0


FAST = bool(os.environ.get('TTM_BAND_CHECK_FAST'))      # (under the profiler: a few launches only)


def timed(fn, n=200):
    n = 5 if FAST else n
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n


out = {}
Xs, Nn = tm._Xs, tm._N
res = {}
for mode in (0, -1):
    opt('band_fwd', mode)
    Z = tm.forward_device(Xs, Nn)
    torch.cuda.synchronize()
    name = lib.ttm_last_kernel().decode()
    res[mode] = Z[:, :Nn].clone()
    # keep the clock up, then time
    t_end = time.time() + (0.0 if FAST else 1.0)
    while time.time() < t_end:
        for _ in range(50):
            tm.forward_device(Xs, Nn, Z=Z)
        torch.cuda.synchronize()
    ms = timed(lambda: tm.forward_device(Xs, Nn, Z=Z))
    out['fwd_%s' % name] = ms
    print('forward', name, '%.4f ms' % ms, ' frac of 8 TB/s: %.3f' % (8.0 * Nn * 2 * tm.D / (ms * 1e-3) / 8e12))
d = (res[0] - res[-1]).abs()
rel = (d / (res[0].abs() + 1.0)).max().item()
print('band vs k_forward_hl: max abs %.3e  max rel %.3e' % (d.max().item(), rel))
out['fwd_band_vs_hl_rel'] = rel
if '--no-oracle' not in sys.argv:
    from oracle.ttm_oracle import OracleMap
    idx = np.concatenate([np.arange(0, N, max(1, N // 4000)), np.argsort(np.abs(X).max(axis=1))[-200:]])
    om = OracleMap(X=X, monotone=cfg['monotone'], nonmonotone=cfg['nonmonotone'], **cfg['kwargs'])
    om.coeffs_mon, om.coeffs_nonmon = [c.copy() for c in tm.coeffs_mon], [c.copy() for c in tm.coeffs_nonmon]
    Zo = om.map(X[idx])
    for mode in (0, -1):
        Zk = res[mode][:, idx].T.cpu().numpy()
        e = np.max(np.abs(Zk - Zo) / (np.abs(Zo) + 1.0))
        print('mode', mode, 'vs oracle: max rel %.3e' % e)
        out['fwd_vs_oracle_%d' % mode] = float(e)
# ---- fused density pass (S, log det, sum of squares) ---------------------------------------------------------------
dres = {}
sigma = tm._to_dev(np.asarray(tm.X_std[:tm.D], dtype=float))
for mode in (0, -1):
    opt('band_fwd', mode)
    for withz in (True, False):
        ld, ss = tm._empty(Nn), tm._empty(Nn)
        Zd = tm._cols(tm.D, Nn) if withz else None
        call = (lambda: tm.forward_device(Xs, Nn, Z=Zd, logdet=ld, sigma=sigma, sumsq=ss)) if withz else \
            (lambda: _capi.check(lib.ttm_forward(tm._pp, tm._ptr(tm._pack_coeffs()), tm._ptr(tm._pack_coeffs()._ttm_fold), tm._ptr(Xs), Xs.shape[1], Nn,
                                                 0, tm.D, None, Nn, tm._ptr(ld), tm._ptr(sigma), tm._ptr(ss), tm._stream())))
        call(); torch.cuda.synchronize()
        name = lib.ttm_last_kernel().decode()
        dres[(mode, withz)] = (ld.clone(), ss.clone())
        ms = timed(call, 50)
        out['density_%s_%s' % (name, 'z' if withz else 'noz')] = ms
        print('density', name, 'with Z' if withz else 'no Z  ', '%.4f ms' % ms, ' frac of 8 TB/s on 8N(d+1): %.3f' % (8.0 * Nn * (tm.D + 1) / (ms * 1e-3) / 8e12))
for withz in (True, False):
    a, b = dres[(0, withz)], dres[(-1, withz)]
    print('density band vs hl (Z %s): logdet max abs %.3e  sumsq max rel %.3e' % (withz, (a[0] - b[0]).abs().max().item(),
                                                                             ((a[1] - b[1]).abs() / (a[1].abs() + 1)).max().item()))
ONLY = os.environ.get('TTM_BAND_CHECK_ONLY', '')
if ONLY == 'fwd':
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'band_check.json'), 'w'), indent=1)
    sys.exit(0)
# ---- inverse --------------------------------------------------------------------------------------------------------
Zs = res[-1] if res[-1].shape[1] == Xs.shape[1] else None
Zin = tm._cols(tm.D, Nn)
Zin[:, :Nn] = res[0]
ires = {}
for mode in (0, -1):
    opt('band_inv', mode)
    Xi = tm.inverse_device(Zin, Nn)
    torch.cuda.synchronize()
    name = lib.ttm_last_kernel().decode()
    ires[mode] = Xi[:, :Nn].clone()
    t_end = time.time() + (0.0 if FAST else 1.0)
    while time.time() < t_end:
        for _ in range(50):
            tm.inverse_device(Zin, Nn, X=Xi)
        torch.cuda.synchronize()
    ms = timed(lambda: tm.inverse_device(Zin, Nn, X=Xi))
    out['inv_%s' % name] = ms
    print('inverse', name, '%.4f ms' % ms, ' frac of 8 TB/s: %.3f' % (8.0 * Nn * 2 * tm.D / (ms * 1e-3) / 8e12))
d = (ires[0] - ires[-1]).abs()
rel = (d / (ires[0].abs() + 1.0)).max().item()
print('band vs k_inverse_rt: max abs %.3e  max rel %.3e' % (d.max().item(), rel))
print('round trip (band): max abs %.3e' % (ires[-1] - Xs[:, :Nn]).abs().max().item())
out['inv_band_vs_rt_rel'] = rel
if '--no-oracle' not in sys.argv:
    Zh = res[0][:, idx].T.cpu().numpy()
    Xo = om.inverse_map(Zh)
    for mode in (0, -1):
        Xk = ires[mode][:, idx].T.cpu().numpy() * tm.X_std[None, :] + tm.X_mean[None, :]
        e = np.max(np.abs(Xk - Xo) / (np.abs(Xo) + 1.0))
        print('inverse mode', mode, 'vs oracle: max rel %.3e' % e)
        out['inv_vs_oracle_%d' % mode] = float(e)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'band_check.json'), 'w'), indent=1)
