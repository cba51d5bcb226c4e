#!/usr/bin/env python3
"""Sweep of RANDOM polynomial integrated-rectifier maps through the X-program kernels (csrc/ttm_xprog.h: k_int_forward, k_int_objective,
k_int_root_x behind option int_xprog = 2) at ensemble sizes of several tiles per workgroup and component chunks on grid.y:
1-6 components, bands 1-3, orders up to 10, products of up to three factors, 0-2 conditioning columns, all six polynomial
families with and without Hermite functions, exponential / softplus rectifiers, N = 70 000 ... 260 000.
Checks per map: forward map against the oracle on 1 500 rows (tails included) 1e-11; map, bisection and Newton roots, objective
and gradient of every component against the kernels that walk the term tables per sample (option int_xprog = 0: an independent
code path over the WHOLE ensemble) 1e-11 / 1e-7 / 1e-11 (Newton: residuals); objective and gradient of one component against the oracle
on the whole ensemble 1e-10.            python tools/fuzz_int.py LO HI   (GPU box)"""
import os
import sys
import traceback
import warnings

import numpy as np

warnings.filterwarnings('ignore')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.ttm_oracle import OracleMap                                  # noqa: E402
from triangular_transport_toolbox_amd import specs                      # noqa: E402
from triangular_transport_toolbox_amd.transport_map import transport_map  # noqa: E402
from tests.util import relerr                                           # noqa: E402

FAMILIES = ['hermite function', "probabilist's hermite", 'hermite', 'power series', 'chebyshev', 'laguerre', 'legendre']


def random_spec(rng):
    D = int(rng.integers(1, 7))
    skip = int(rng.integers(0, 3)) if D >= 2 else 0
    band = int(rng.integers(1, 4))
    order = int(rng.choice([2, 3, 3, 4, 5, 5, 7, 10]))
    fam = str(rng.choice(FAMILIES, p=[0.4, 0.15, 0.09, 0.09, 0.09, 0.09, 0.09]))
    plain_ok = fam not in ('hermite function',)
    mon, non = [], []
    for k in range(D):
        kc = k + skip
        lo = max(0, kc - band)
        hf = rng.random() < 0.75 or not plain_ok
        m = []
        for o in range(1, order + 1):
            if o > 1 and rng.random() < 0.35:
                continue
            m.append([kc] * o + (['HF'] if hf else []))
        cols = list(range(lo, kc))
        for _ in range(int(rng.integers(0, 7)) if cols else 0):
            nf = int(rng.integers(1, min(3, len(cols)) + 1))
            vs = sorted(int(v) for v in rng.choice(cols, size=nf, replace=False))
            term = []
            tot = 0
            for v in vs:
                p = int(rng.integers(1, max(2, order - 1)))
                term += [v] * p
                tot += p
            ok = int(rng.integers(1, max(2, order - tot + 1)))
            if tot + ok > order + 1 or max(np.unique(term, return_counts=True)[1]) > order:
                continue
            m.append(sorted(term + [kc] * ok) + (['HF'] if hf else []))
        if rng.random() < 0.15:
            m.append([])
        seen, mm = set(), []
        for t in m:
            if repr(t) not in seen:
                seen.add(repr(t)); mm.append(t)
        mon.append(mm)
        n = [[]]
        for j in cols:
            for o in range(1, 1 + int(rng.integers(0, min(order, 4) + 1))):
                n.append([j] * o + (['HF'] if (o > 1 and hf) else []))
        if len(cols) >= 2 and rng.random() < 0.5:
            a, b = sorted(int(v) for v in rng.choice(cols, size=2, replace=False))
            n.append([a, b] + (['HF'] if hf else []))
        seen, nn = set(), []
        for t in n:
            if repr(t) not in seen:
                seen.add(repr(t)); nn.append(t)
        non.append(nn)
    return D, skip, fam, order, mon, non


def kernel(tm):
    import ctypes
    tm._lib.ttm_last_kernel.restype = ctypes.c_char_p
    return tm._lib.ttm_last_kernel().decode()


def one(seed):
    rng = np.random.default_rng(50000 + seed)
    D, skip, fam, order, mon, non = random_spec(rng)
    d = D + skip
    N = int(rng.integers(70000, 260000))
    X = specs.sample_banana(N, d=d, seed=seed) if d <= 4 else rng.standard_normal((N, d)) @ (np.eye(d) + 0.3 * np.tri(d, k=-1))
    kw = dict(monotonicity='integrated rectifier', polynomial_type=fam, rectifier_type=str(rng.choice(['exponential', 'exponential', 'softplus'])),
              quadrature_input={'order': int(rng.choice([10, 20, 25]))})
    tm = transport_map(X=X, monotone=mon, nonmonotone=non, verbose=False, **kw)
    if not all(int(f) & 16 for f in tm._cm.complex):
        return 'no X program'
    om = OracleMap(X=X, monotone=mon, nonmonotone=non, **kw)
    # (plain polynomials of high order under an exponential rectifier: small coefficients, or the map is noise at the oracle's own level)
    scale = 0.25 / max(1.0, order / 3.0) / (1.0 if fam == 'hermite function' else max(1.0, order - 2.0))
    for k in range(D):
        cm, cn = scale * rng.standard_normal(len(tm.coeffs_mon[k])), 0.3 * rng.standard_normal(len(tm.coeffs_nonmon[k]))
        tm.coeffs_mon[k], om.coeffs_mon[k] = cm.copy(), cm.copy()
        tm.coeffs_nonmon[k], om.coeffs_nonmon[k] = cn.copy(), cn.copy()
    tm.alternate_root_finding = False
    idx = np.unique(np.concatenate((rng.choice(N, 1200, replace=False), np.argsort(X, axis=0)[:8].ravel(), np.argsort(X, axis=0)[-8:].ravel())))
    Zin = rng.standard_normal((4096, D))
    Xstar = X[:4096, :skip] if skip else None

    def everything(mode):
        tm._lib.ttm_set_option(b'int_xprog', mode)
        out = {'Z': tm.map(X)}
        tm.forward_device(tm._Xs, tm._N)
        out['kf'] = kernel(tm)
        out['Xi'] = tm.inverse_map(Zin, X_star=Xstar)
        tm.root_finder = 'newton'
        out['Xn'] = tm.inverse_map(Zin, X_star=Xstar)
        tm.root_finder = 'reference'
        out['S'] = []
        for k in range(D):
            c = np.concatenate((tm.coeffs_nonmon[k], tm.coeffs_mon[k]))
            out['S'].append(tm._device_sums(k, c))
            out['ko'] = kernel(tm)
        tm._lib.ttm_reset_options()
        return out
    a, b = everything(2), everything(0)
    assert a['kf'] == 'k_int_forward' and b['kf'] == 'k_int_forward<walk>', (a['kf'], b['kf'])
    assert a['ko'] == 'k_int_objective' and b['ko'] == 'k_int_objective_walk', (a['ko'], b['ko'])
    with np.errstate(all='ignore'):
        Zo = om.map(X[idx])
        fin = np.isfinite(Zo)
        assert np.array_equal(np.isfinite(a['Z'][idx]), fin), 'finite pattern of the map'
        e_or = relerr(a['Z'][idx][fin], Zo[fin])
        finw = np.isfinite(b['Z'])
        assert np.array_equal(np.isfinite(a['Z']), finw)
        e_map = relerr(a['Z'][finw], b['Z'][finw])
        ok = np.all(np.isfinite(b['Xi']), axis=1) & np.all(np.isfinite(b['Xn']), axis=1)
        assert np.array_equal(np.all(np.isfinite(a['Xi']), axis=1) & np.all(np.isfinite(a['Xn']), axis=1), ok), 'finite pattern of the roots'
        # (targets a random map cannot attain send the reference's window doubling to 1e9 and beyond: not compared)
        ok &= (np.max(np.abs(np.nan_to_num(b['Xi'])), axis=1) < 1e3) & (np.max(np.abs(np.nan_to_num(b['Xn'])), axis=1) < 1e3)
        e_bis = relerr(a['Xi'][ok][1:], b['Xi'][ok][1:]) if ok.sum() > 1 else 0.0
        # Newton: both searches stop at |S - z| <= 1e-9 under their own evaluator - where a random map is nearly flat (or runs away)
        # their end points need not agree: at most 1 % of the rows may differ by more than 1e-6 (1 + |x|)
        e_new = 0.0
        if ok.any():
            dn = np.max(np.abs(a['Xn'][ok] - b['Xn'][ok]) / (1.0 + np.abs(b['Xn'][ok])), axis=1)
            e_new = float(np.quantile(dn, 0.99))
        e_sum = 0.0
        for sa, sb in zip(a['S'], b['S']):
            if np.all(np.isfinite(sb)):
                e_sum = max(e_sum, float(np.max(np.abs(sa - sb) / (np.abs(sb) + N))))
            else:
                assert not np.all(np.isfinite(sa))
        k = int(rng.integers(0, D))
        div = len(tm.coeffs_nonmon[k])
        c = np.concatenate((tm.coeffs_nonmon[k], tm.coeffs_mon[k]))
        e_j = e_g = 0.0
        if div > 0:
            Jo, Go = om.objective_function(c.copy(), k, div), om.objective_function_jacobian(c.copy(), k, div)
            J, G = tm.objective_function(c.copy(), k, div), tm.objective_function_jacobian(c.copy(), k, div)
            if np.isfinite(Jo) and np.all(np.isfinite(Go)):
                e_j, e_g = abs(J - Jo) / (1 + abs(Jo)), relerr(G, Go)
            else:
                assert not (np.isfinite(J) and np.all(np.isfinite(G)))
    # (against the oracle: 1e-10 for random Hermite-function maps, as tests/test_random_maps.py; random PLAIN polynomials of order 5-10
    # under an exponential rectifier are at the oracle's own noise floor well above that: 1e-8)
    tol_or = 1e-10 if fam == 'hermite function' else 1e-8
    assert e_or < tol_or and e_map < 1e-11 and e_bis < 1e-7 and e_new < 1e-6 and e_sum < 1e-11 and e_j < 10 * tol_or and e_g < 10 * tol_or, \
        'oracle map %.1e walk map %.1e bisect %.1e newton %.1e sums %.1e J %.1e G %.1e' % (e_or, e_map, e_bis, e_new, e_sum, e_j, e_g)
    return 'D %d skip %d %s order %d N %d  oracle map %.1e  walk: map %.1e bisect %.1e newton %.1e sums %.1e  oracle J %.1e G %.1e' % (
        D, skip, fam, order, N, e_or, e_map, e_bis, e_new, e_sum, e_j, e_g)


lo, hi = int(sys.argv[1]), int(sys.argv[2])
fails = skipped = 0
worst = {}
for seed in range(lo, hi):
    try:
        r = one(seed)
        if r == 'no X program':
            skipped += 1
        elif (seed - lo) % 10 == 0:
            print(seed, r, flush=True)
    except Exception as e:       # noqa: BLE001
        fails += 1
        fr = traceback.extract_tb(e.__traceback__)[-1]
        print('FAIL seed', seed, type(e).__name__, str(e)[:300], 'line', fr.lineno, flush=True)
print('seeds', lo, hi, 'fails', fails, 'without X program', skipped)
