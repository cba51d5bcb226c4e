"""
Random banded separable maps of a few components (1..4, lags 1..3, conditioning columns, all degree classes and
polynomial families, pure Hermite-function or mixed groups) through k_band_few / k_band_few_inverse / the density variant,
forced at small N, against the oracle: map, sweeps that start inside the map, table inverse (targets in the tails
included), pullback density.  16 fixed seeds here; tools/fuzz_few.py runs the same check over any range of seeds.
"""
import ctypes

import numpy as np
import pytest

from tests.util import relerr

pytestmark = pytest.mark.gpu


def _lib():
    from triangular_transport_toolbox_amd import _capi
    lib = _capi.load()
    lib.ttm_last_kernel.restype = ctypes.c_char_p
    lib.ttm_set_option.argtypes = [ctypes.c_char_p, ctypes.c_int32]
    return lib


def random_spec(rng):
    D = int(rng.integers(1, 5))
    skip = int(rng.integers(0, 3))
    maxlag = int(rng.integers(1, 4))
    hf_max = int(rng.choice([3, 5, 7]))
    plain_max = int(rng.choice([0, 1, 3, 6]))
    family = 'hermite function'
    if rng.random() < 0.3:                                     # another polynomial family: plain terms only
        family = str(rng.choice(['legendre', 'chebyshev', 'power series', "probabilist's hermite", 'hermite', 'laguerre']))
        hf_max, plain_max = 0, int(rng.choice([1, 3, 5]))
    mon, non = [], []
    for k in range(D):
        kc = k + skip
        nm = [[]]
        for lag in range(1, maxlag + 1):
            j = kc - lag
            if j < 0 or rng.random() < 0.25:
                continue
            for o in range(1, plain_max + 1):
                if rng.random() < 0.7:
                    nm.append([j] * o)
            for o in range(1, hf_max + 1):
                if rng.random() < 0.6:
                    nm.append([j] * o + ['HF'])
        non.append(nm)
        n_irbf = int(rng.integers(0, 4))
        m = ['LET %d' % kc] + ['iRBF %d' % kc] * n_irbf + ['RET %d' % kc]
        u = rng.random() if family != 'laguerre' else 1.0         # (L_1 = 1 - x decreases: with it the monotone part is not monotone)
        if u < 0.3:
            m = [[kc]] + m                                      # a linear term next to the special terms (example 05)
        elif u < 0.45 and k > 0:
            m = [[kc]]                                          # ... or instead of them (examples 06 / 07)
        mon.append(m)
    return D, skip, mon, non, family


def one(seed):
    from oracle.ttm_oracle import OracleMap
    from triangular_transport_toolbox_amd.transport_map import transport_map
    lib = _lib()
    rng = np.random.default_rng(seed)
    D, skip, mon, non, family = random_spec(rng)
    d = D + skip
    n = int(rng.choice([257, 2049, 5003]))
    X = rng.standard_normal((n, d)) @ (np.tril(rng.standard_normal((d, d)) * 0.4) + np.eye(d)).T + 0.3 * rng.standard_normal((n, d)) ** 2
    kw = dict(monotonicity='separable monotonicity', polynomial_type=family)
    tm = transport_map(X=X, monotone=mon, nonmonotone=non, verbose=False, **kw)
    om = OracleMap(X=X, monotone=mon, nonmonotone=non, **kw)
    for k in range(D):
        cm_ = 0.2 + 0.5 * rng.random(len(tm.coeffs_mon[k]))
        cn_ = 0.3 * rng.standard_normal(len(tm.coeffs_nonmon[k])) / (1 + np.arange(len(tm.coeffs_nonmon[k])))
        tm.coeffs_mon[k], om.coeffs_mon[k] = cm_.copy(), cm_.copy()
        tm.coeffs_nonmon[k], om.coeffs_nonmon[k] = cn_.copy(), cn_.copy()
    banded = tm._cm.u_p_lag > 0
    for name in (b'u_loader', b'band_fwd', b'band_inv'):
        lib.ttm_set_option(name, 1)
    Xq = X.copy()
    Xq[:8] += 12.0 * X.std(0)                                  # far tails
    Z, Zo = tm.map(Xq), om.map(Xq)
    tm.forward_device(tm._Xs, tm._N)
    kf = lib.ttm_last_kernel().decode()
    assert relerr(Z, Zo) < 5e-11, ('map', relerr(Z, Zo))      # (rows 12 sigma out under degree-5/6 polynomials: cancellation)
    Xs = (X - om.X_mean) / om.X_std
    for k in range(D):
        assert relerr(tm.s(Xs, k), om.s(Xs, k)) < 1e-11, ('s', k)
    Zin = rng.standard_normal((n, D))
    Zin[:20] *= 4.0
    star = X[:, :skip] if skip else None
    Xi, Xo = tm.inverse_map(Zin, X_star=star), om.inverse_map(Zin, X_star=star)
    tm.inverse_device(tm._cols(D, tm._N, zero=True), tm._N, X=tm._Xs.clone())
    ki = lib.ttm_last_kernel().decode()
    assert relerr(Xi, Xo) < 1e-9, ('inverse', relerr(Xi, Xo))
    # the kernels these replace (generic U-form map, table lookup in memory): the same numbers to rounding, whatever the
    # conditioning of the map
    lib.ttm_set_option(b'band_fwd', 0); lib.ttm_set_option(b'band_inv', 0)
    Zg, Xg = tm.map(Xq), tm.inverse_map(Zin, X_star=star)
    lib.ttm_set_option(b'band_fwd', 1); lib.ttm_set_option(b'band_inv', 1)
    assert relerr(Z, Zg) < 1e-12 and relerr(Xi, Xg) < 1e-11, ('vs generic', relerr(Z, Zg), relerr(Xi, Xg))
    if skip == 0:
        with np.errstate(all='ignore'):
            po = om.evaluate_pullback_density(Xq[:300])
        p = tm.evaluate_pullback_density(Xq[:300])
        ok = np.isfinite(po) & (po > 1e-290)
        assert np.array_equal(np.isfinite(p), np.isfinite(po)), 'density finite mask'
        assert relerr(np.log(p[ok]), np.log(po[ok])) < 1e-9, ('density', relerr(np.log(p[ok]), np.log(po[ok])))
    return banded, kf, ki, (D, skip, tm._cm.u_p_lag, tm._cm.u_h_cls, family)


@pytest.mark.parametrize('seed', range(16))
def test_random_maps_of_a_few_components(seed):
    lib = _lib()
    try:
        banded, kf, ki, info = one(seed)
    finally:
        lib.ttm_reset_options()
    # (a table that is not increasing - e.g. a Laguerre L_1 = 1 - x term with a positive coefficient - takes the sorted lookup)
    assert banded and kf == 'k_band_few' and ki in ('k_band_few_inverse', 'k_inverse_table'), (kf, ki, info)
