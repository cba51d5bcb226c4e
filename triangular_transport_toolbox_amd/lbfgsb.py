"""
L-BFGS-B exactly as `scipy.optimize.minimize(method='L-BFGS-B')` runs it (the reference's call, TM:3108-3114, default
options), without the per-evaluation wrapper layers.

`minimize` reaches the compiled routine `scipy.optimize._lbfgsb.setulb` through `_minimize_lbfgsb`, a reverse-
communication loop of a dozen lines; around every objective evaluation it puts `ScalarFunction`, `MemoizeJac` and
several argument-checking wrappers - ~40 us of host Python per evaluation, as much as the device reduction and its
synchronisation together.  `minimize_lbfgsb` below is that same loop (same workspace sizes, same
task codes, same defaults: maxcor 10, ftol 2.22e-9, gtol 1e-5, maxfun = maxiter = 15000, maxls 20) calling the objective
directly, so iterates, evaluation points and stopping are those of `minimize` bit for bit
(tests/test_lbfgsb.py).  `setulb` is a private SciPy entry point: the loop is used only with the SciPy series it was
written against (1.15: the C port of L-BFGS-B 3.0) and when a self-check against `minimize` on a small problem
agrees; otherwise `minimize` itself is called.
"""
import os

import numpy as np

_STATE = {'checked': False, 'ok': False}


class Result:
    __slots__ = ('x', 'fun', 'jac', 'nfev', 'nit', 'status', 'success')


def _direct(fun, x0, bounds, args, maxcor=10, ftol=2.220446049250313e-09, gtol=1e-5, maxfun=15000, maxiter=15000,
            maxls=20):
    from scipy.optimize import _lbfgsb
    m = maxcor
    factr = ftol / np.finfo(float).eps
    x0 = np.asarray(x0, dtype=float).ravel()
    n = x0.shape[0]
    lo = np.array([-np.inf if b[0] is None else b[0] for b in bounds], dtype=float)
    hi = np.array([np.inf if b[1] is None else b[1] for b in bounds], dtype=float)
    if (lo > hi).any():
        raise ValueError("LBFGSB - one of the lower bounds is greater than an upper bound.")
    x0 = np.clip(x0, lo, hi)
    nbd = np.zeros(n, np.int32)
    low_bnd = np.zeros(n, np.float64)
    upper_bnd = np.zeros(n, np.float64)
    for i in range(n):
        L, U = not np.isinf(lo[i]), not np.isinf(hi[i])
        if L:
            low_bnd[i] = lo[i]
        if U:
            upper_bnd[i] = hi[i]
        nbd[i] = (1 if L and not U else 2 if L and U else 3 if U else 0)
    x = np.array(x0, dtype=np.float64)
    f = np.array(0.0, dtype=np.int32)
    g = np.zeros((n,), dtype=np.int32)
    wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64)
    iwa = np.zeros(3 * n, dtype=np.int32)
    task = np.zeros(2, dtype=np.int32)
    ln_task = np.zeros(2, dtype=np.int32)
    lsave = np.zeros(4, dtype=np.int32)
    isave = np.zeros(44, dtype=np.int32)
    dsave = np.zeros(29, dtype=np.float64)
    nit = nfev = 0
    while True:
        g = g.astype(np.float64)
        _lbfgsb.setulb(m, x, low_bnd, upper_bnd, nbd, f, g, factr, gtol, wa, iwa, task, lsave, isave, dsave, maxls, ln_task)
        if task[0] == 3:
            f, g = fun(np.copy(x), *args)
            nfev += 1
        elif task[0] == 1:
            nit += 1
            if nit >= maxiter:
                task[0], task[1] = 5, 504
            elif nfev > maxfun:
                task[0], task[1] = 5, 502
        else:
            break
    r = Result()
    r.x, r.fun, r.jac, r.nfev, r.nit = x, f, g, nfev, nit
    r.status = 0 if task[0] == 4 else (1 if (nfev > maxfun or nit >= maxiter) else 2)
    r.success = r.status == 0
    return r


def _self_check():
    """The direct loop against scipy.optimize.minimize on a small bound-constrained problem: identical iterates."""
    import scipy
    from scipy.optimize import minimize
    try:
        if tuple(int(v) for v in scipy.__version__.split('.')[:2]) != (1, 15):
            return False
        rng = np.random.default_rng(0)
        Q = rng.standard_normal((6, 6))
        Q = Q @ Q.T + np.eye(6)
        b = rng.standard_normal(6)
        seen_a, seen_b = [], []

        def make(seen):
            def fun(c, w):
                seen.append(c.copy())
                d = c + 1.5
                return 0.5 * c @ Q @ c + b @ c - w * np.sum(np.log(d)), Q @ c + b - w / d
            return fun
        bounds = [[0.0, np.inf]] * 5 + [[-np.inf, np.inf]]
        ref = minimize(make(seen_a), np.full(6, 0.3), jac=True, method='L-BFGS-B', bounds=bounds, args=(0.7,))
        got = _direct(make(seen_b), np.full(6, 0.3), bounds, (0.7,))
        return (len(seen_a) == len(seen_b) and all(np.array_equal(p, q) for p, q in zip(seen_a, seen_b)) and
                np.array_equal(ref.x, got.x) and float(ref.fun) == float(got.fun))
    except Exception:                          # noqa: BLE001  (any surprise in the private API: use minimize)
        return False


def minimize_lbfgsb(fun, x0, bounds, args=()):
    """fun(x, *args) -> (f, g).  Returns an object with .x and .fun (and .jac, .nfev, .nit, .status, .success)."""
    if os.environ.get('TTM_LBFGSB_MINIMIZE'):      # (tests: force the public entry point)
        from scipy.optimize import minimize
        return minimize(fun=fun, method='L-BFGS-B', x0=x0, jac=True, bounds=bounds, args=args)
    if not _STATE['checked']:
        _STATE['ok'] = _self_check()
        _STATE['checked'] = True
    if _STATE['ok']:
        return _direct(fun, x0, bounds, args)
    from scipy.optimize import minimize
    return minimize(fun=fun, method='L-BFGS-B', x0=x0, jac=True, bounds=bounds, args=args)
