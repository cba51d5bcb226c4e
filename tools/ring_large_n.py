#!/usr/bin/env python3
"""k_band_inverse_ring at ensemble sizes with several tiles per workgroup (N = 2.5e6, 4.1e6 + 3: odd tail) against k_band_inverse bit for bit,
against the oracle on a subset with tails, and the round trip through k_band_forward; C5 map (d = 40)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests.test_full_size import build, subset_with_tails
from tests.util import relerr
from triangular_transport_toolbox_amd import _capi

lib = _capi.load()
lib.ttm_last_kernel.restype = ctypes.c_char_p
for N in (2500000, 4100003):
    tm, om, X = build('C5', 'c5_sep', N)
    Z = tm.forward_device(tm._Xs, tm._N)
    res = {}
    for ring in (1, 0):
        _capi.set_option('band_ring', ring)
        Xi = tm.inverse_device(Z, tm._N)
        torch.cuda.synchronize()
        res[ring] = (Xi[:, :N].clone(), lib.ttm_last_kernel().decode())
    _capi.set_option('band_ring', -1)
    same = torch.equal(res[1][0], res[0][0])
    md = (res[1][0] - res[0][0]).abs().max().item()
    rt = (res[1][0] - tm._Xs[:, :N]).abs().max().item()
    idx = subset_with_tails(X, 4000)
    Zh = Z[:, :N].T[torch.from_numpy(idx).to(Z.device)].cpu().numpy()
    Xo = om.inverse_map(Zh)
    Xs = (Xo - om.X_mean) / om.X_std
    got = res[1][0].T[torch.from_numpy(idx).to(Z.device)].cpu().numpy()
    print('N', N, res[1][1], 'vs', res[0][1], 'bit-identical' if same else 'max abs diff %.3e' % md, '| round trip max abs %.2e | oracle (ring) rel %.2e' % (rt, relerr(got, Xs)), flush=True)
    assert res[1][1] == 'k_band_inverse_ring' and res[0][1] == 'k_band_inverse'
    assert md < 1e-12 and relerr(got, Xs) < 1e-10
    del tm, om, X, Z, res
    torch.cuda.empty_cache()
print('ok')
