#!/usr/bin/env python3
"""Every kernel path of separable maps against the oracle and against each other on random maps (GPU): D = 1..8 components,
bands 1..4, 0..2 conditioning columns, Hermite-function or plain-only terms of any polynomial family, linear own terms;
per map the launch-planning options are walked through (generic / planned / U-form with and without hot records /
loader waves / band and few-component kernels / resident-table and generic inverse, rows per thread), each time with a
table inverse in front of the forward map (whatever one kernel leaves in LDS is what the next one finds).
    python tools/fuzz_paths.py SEED_LO SEED_HI"""
import ctypes, os, sys, traceback, warnings
import numpy as np
warnings.filterwarnings('ignore')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.util import relerr  # noqa: E402
from oracle.ttm_oracle import OracleMap  # noqa: E402
from triangular_transport_toolbox_amd import _capi  # noqa: E402
from triangular_transport_toolbox_amd.transport_map import transport_map  # noqa: E402

lib = _capi.load()
lib.ttm_last_kernel.restype = ctypes.c_char_p
lib.ttm_set_option.argtypes = [ctypes.c_char_p, ctypes.c_int32]

PATHS = [dict(), dict(u_loader=1), dict(u_loader=1, band_fwd=0, band_inv=0), dict(u_loader=1, band_fwd=0, band_inv=0, rt_band=0),
         dict(u_loader=1, band_fwd=0, band_inv=0, hl_ns=2, rt_ns=2), dict(u_loader=1, u_no_hot=1), dict(u_loader=0, band_fwd=0, u_ns=4),
         dict(no_uform=1), dict(no_plan=1, no_uform=1), dict(u_loader=1, rt_off=1, band_inv=0), dict(u_loader=1, band_fwd=1, band_inv=1, band_cus=3, rt_block=2),
         # (round 4: the table inverse with resident-table images - all resident, rings of 12 / 16 slots, several tiles - and without)
         dict(u_loader=1, band_ring=0), dict(u_loader=1, rt_block=12), dict(u_loader=1, rt_block=16, band_cus=1), dict(u_loader=1, rt_window=400)]


def random_spec(rng, long_map=False):
    D = int(rng.integers(1, 9))
    skip = int(rng.choice([0, 0, 0, 1, 2]))
    band = int(rng.choice([1, 2, 2, 2, 3, 4]))
    if long_map:                                     # (seeds >= 100000: maps long enough for the ring of resident tables to be refilled)
        D, skip, band = int(rng.integers(13, 34)), 0, int(rng.choice([1, 2, 2, 2]))
    family = 'hermite function'
    hf_max, plain_max = int(rng.choice([2, 3, 5])), int(rng.choice([0, 1, 3]))
    if rng.random() < 0.4 and not long_map:
        family = str(rng.choice(['legendre', 'chebyshev', 'power series', "probabilist's hermite", 'hermite']))
        hf_max, plain_max = 0, int(rng.choice([1, 2, 3]))
    mon, non = [], []
    for k in range(D):
        kc = k + skip
        nm = [[]]
        for lag in range(1, band + 1):
            j = kc - lag
            if j < 0 or rng.random() < 0.2:
                continue
            for o in range(1, plain_max + 1):
                if rng.random() < 0.7:
                    nm.append([j] * o)
            for o in range(1, hf_max + 1):
                if rng.random() < 0.6:
                    nm.append([j] * o + ['HF'])
        non.append(nm)
        m = ['LET %d' % kc] + ['iRBF %d' % kc] * int(rng.integers(0, 3)) + ['RET %d' % kc]
        if rng.random() < 0.25 and not long_map:
            m = [[kc]] + m
        mon.append(m)
    return D, skip, mon, non, family


def one(seed):
    rng = np.random.default_rng(9000 + seed)
    D, skip, mon, non, family = random_spec(rng, long_map=seed >= 100000)
    d = D + skip
    n = int(rng.choice([257, 2049, 5003]))
    X = rng.standard_normal((n, d)) @ (np.tril(rng.standard_normal((d, d)) * 0.3) + np.eye(d)).T
    kw = dict(monotonicity='separable monotonicity', polynomial_type=family)
    tm = transport_map(X=X, monotone=mon, nonmonotone=non, verbose=False, **kw)
    om = OracleMap(X=X, monotone=mon, nonmonotone=non, **kw)
    for k in range(D):
        cm_ = 0.2 + 0.5 * rng.random(len(tm.coeffs_mon[k]))
        cn_ = (0.06 if seed >= 100000 else 0.3) * rng.standard_normal(len(tm.coeffs_nonmon[k])) / (1 + np.arange(len(tm.coeffs_nonmon[k])))
        tm.coeffs_mon[k], om.coeffs_mon[k] = cm_.copy(), cm_.copy()
        tm.coeffs_nonmon[k], om.coeffs_nonmon[k] = cn_.copy(), cn_.copy()
    Zo = om.map(X)
    Zin = rng.standard_normal((n, D))
    star = X[:, :skip] if skip else None
    Xo = om.inverse_map(Zin, X_star=star)
    inside = np.isfinite(Xo).all(axis=1) & (np.abs((Xo - om.X_mean[skip:]) / om.X_std[skip:]) < 9.9).all(axis=1)
    po = None
    if skip == 0:
        with np.errstate(all='ignore'):
            po = om.evaluate_pullback_density(X[:300])
    seen = []
    for path in PATHS:
        lib.ttm_reset_options()
        for name, v in path.items():
            lib.ttm_set_option(name.encode(), v)
        Xi = tm.inverse_map(Zin, X_star=star)
        tm.inverse_device(tm._cols(D, tm._N, zero=True), tm._N, X=tm._Xs.clone())
        ki = lib.ttm_last_kernel().decode()
        Z = tm.map(X)
        tm.forward_device(tm._Xs, tm._N)
        kf = lib.ttm_last_kernel().decode()
        seen.append((kf, ki))
        assert np.isfinite(Z).all() and relerr(Z, Zo) < 2e-11, ('map', path, kf, relerr(Z, Zo))
        assert relerr(Xi[inside], Xo[inside]) < 1e-9, ('inverse', path, ki, relerr(Xi[inside], Xo[inside]))
        if po is not None:
            p = tm.evaluate_pullback_density(X[:300])
            ok = np.isfinite(po) & (po > 1e-290)
            assert np.array_equal(np.isfinite(p), np.isfinite(po)) and relerr(np.log(p[ok]), np.log(po[ok])) < 1e-9, ('density', path, kf)
    lib.ttm_reset_options()
    return seen


lo, hi = int(sys.argv[1]), int(sys.argv[2])
fails, count = 0, {}
for seed in range(lo, hi):
    try:
        for pair in one(seed):
            count[pair] = count.get(pair, 0) + 1
    except Exception as e:       # noqa: BLE001
        fails += 1
        lib.ttm_reset_options()
        fr = traceback.extract_tb(e.__traceback__)[-1]
        print('seed', seed, type(e).__name__, str(e)[:200], 'line', fr.lineno, flush=True)
print('seeds', lo, hi, 'fails', fails)
for k, v in sorted(count.items(), key=lambda kv: -kv[1]):
    print('  %5d  %s' % (v, k))
