"""The native L-BFGS-B loop (csrc/ttm_lbfgsb.h behind ttm_lbfgsb_minimize) against scipy.optimize.minimize
(method 'L-BFGS-B', the optimiser of the reference, TM:3108-3114) on problems of the shape optimize() poses:
J(c) = c'Ac/2 - mean log(dPsi c + delta rowsum dPsi) + c.b with c >= 0 on all but the constant's coefficient.
Same algorithm, so the iteration counts agree and the minimisers agree far inside the optimiser's own tolerance."""
import ctypes

import numpy as np
import pytest
from scipy.optimize import minimize

from triangular_transport_toolbox_amd import _capi


def native_minimize(lib, fun, x0, lb, ub, maxiter=0):
    n = len(x0)
    x = np.array(x0, dtype=float, copy=True)
    lb = np.array(lb, dtype=float)
    ub = np.array(ub, dtype=float)
    calls = []

    @_capi.OBJECTIVE_CB
    def cb(n_, xp, fp, gp, user):
        xx = np.ctypeslib.as_array(xp, shape=(n_,)).copy()
        f, g = fun(xx)
        fp[0] = f
        for i in range(n_):
            gp[i] = g[i]
        calls.append(xx)
        return 0
    res = np.zeros(5)
    rc = lib.ttm_lbfgsb_minimize(n, x.ctypes.data, lb.ctypes.data, ub.ctypes.data, ctypes.cast(cb, ctypes.c_void_p), None, maxiter,
                                 res.ctypes.data)
    assert rc == 0
    return x, dict(f=res[0], pg=res[1], nit=int(res[2]), nfev=int(res[3]), status=int(res[4])), calls


def separable_problem(seed, m, N=400):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(N)
    # derivative basis of a separable component: constant-derivative term, iRBF-like bumps, edge terms
    dpsi = np.column_stack([0.5 * (1 - np.tanh(x + 1))] + [np.exp(-0.5 * ((x - c) / 0.6) ** 2) for c in np.linspace(-1.5, 1.5, m - 2)]
                           + [0.5 * (1 + np.tanh(x - 1))])
    psi = np.cumsum(dpsi[np.argsort(x)], axis=0)[np.argsort(np.argsort(x))] / N
    A = psi.T @ psi / N + 1e-3 * np.eye(m)
    delta = 1e-8
    b = delta * A.sum(axis=1)

    def fun(c):
        dS = dpsi @ c + delta * dpsi.sum(axis=1)
        Ac = A @ c
        return c @ Ac / 2 - np.mean(np.log(dS)) + c @ b, Ac - (dpsi / dS[:, None]).mean(axis=0) + b
    lb = np.zeros(m)
    ub = np.full(m, np.inf)
    return fun, lb, ub


@pytest.mark.parametrize('seed,m', [(0, 4), (1, 5), (2, 6), (3, 9), (4, 13)])
def test_matches_scipy_on_separable_component_problems(seed, m):
    from tests.hostemu import emu
    fun, lb, ub = separable_problem(seed, m)
    x0 = np.full(m, 0.3)
    ref = minimize(fun, x0, jac=True, method='L-BFGS-B', bounds=list(zip(lb, ub)))
    x, info, _ = native_minimize(emu.lib(), fun, x0, lb, ub)
    assert info['status'] in (0, 1)
    assert abs(info['f'] - ref.fun) <= 1e-9 * (1 + abs(ref.fun))
    assert np.max(np.abs(x - ref.x)) < 2e-5
    assert abs(info['nit'] - ref.nit) <= 2 and abs(info['nfev'] - ref.nfev) <= 3


def test_same_iterates_as_scipy_on_a_bounded_quadratic():
    """Convex quadratic with active bounds: Cauchy point, subspace step and line search are all exercised; the
    sequences of evaluation points agree to rounding."""
    from tests.hostemu import emu
    rng = np.random.default_rng(7)
    n = 8
    Q = rng.standard_normal((n, n))
    Q = Q @ Q.T + 0.5 * np.eye(n)
    c = rng.standard_normal(n) * 3

    def fun(x):
        return 0.5 * x @ Q @ x - c @ x, Q @ x - c
    lb = np.full(n, -0.2)
    ub = np.full(n, 0.6)
    lb[0], ub[1] = -np.inf, np.inf
    x0 = np.zeros(n)
    pts = []

    def traced(x):
        pts.append(np.array(x, copy=True))
        return fun(x)
    ref = minimize(traced, x0, jac=True, method='L-BFGS-B', bounds=list(zip(lb, ub)))
    x, info, calls = native_minimize(emu.lib(), fun, x0, lb, ub)
    assert len(calls) == len(pts)
    for a, b in zip(calls, pts):
        assert np.max(np.abs(a - b)) < 1e-9
    assert np.max(np.abs(x - ref.x)) < 1e-10 and info['nit'] == ref.nit


def test_unconstrained_rosenbrock_and_limits():
    from tests.hostemu import emu
    from scipy.optimize import rosen, rosen_der
    x0 = np.array([-1.2, 1.0, 0.7, -0.4])
    ref = minimize(rosen, x0, jac=rosen_der, method='L-BFGS-B')
    x, info, _ = native_minimize(emu.lib(), lambda v: (rosen(v), rosen_der(v)), x0, np.full(4, -np.inf), np.full(4, np.inf))
    assert np.max(np.abs(x - ref.x)) < 1e-4 and abs(info['nit'] - ref.nit) <= 3
    x, info, _ = native_minimize(emu.lib(), lambda v: (rosen(v), rosen_der(v)), x0, np.full(4, -np.inf), np.full(4, np.inf), maxiter=3)
    assert info['status'] == 2 and info['nit'] == 3


def test_native_l2_reduction_equals_the_numpy_path():
    """ttm_separable_reduce_l2 (csrc/ttm_optim.cpp) against transport_map.separable_setup's NumPy / LAPACK arithmetic on
    random Gram matrices: A and the nonmonotone solve agree to rounding; a Gram matrix that is not positive definite is
    handed back."""
    import ctypes
    from tests.hostemu import emu
    from triangular_transport_toolbox_amd.transport_map import transport_map as T
    rng = np.random.default_rng(3)
    lib = emu.lib()
    if True:
        for n, m, lam in ((5, 4, 0.05), (9, 1, 0.05), (13, 5, 1e-3), (1, 1, 0.5), (40, 8, 0.05)):
            B = rng.standard_normal((300, n + m)) * (1 + 10 * rng.random(n + m))
            G = B.T @ B
            Gnn, Gnm, Gmm = G[:n, :n], G[:n, n:], G[n:, n:]
            Gm = T._normal_solve(Gnn, Gnm, lam)
            dd = Gmm - Gnm.T @ Gm - Gm.T @ Gnm + Gm.T @ Gnn @ Gm
            A0 = dd / 2 + lam * (Gm.T @ Gm + np.identity(m))
            A0 = (A0 + A0.T) / 2
            S0 = T._normal_solve(Gnn, Gnm, 2 * lam)
            A, S = np.empty((m, m)), np.empty((n, m))
            Gc = np.ascontiguousarray(G)
            assert lib.ttm_separable_reduce_l2(ctypes.c_void_p(Gc.ctypes.data), n, m, lam, ctypes.c_void_p(A.ctypes.data),
                                               ctypes.c_void_p(S.ctypes.data)) == 0
            assert np.max(np.abs(A - A0)) <= 1e-12 * np.max(np.abs(A0)) and np.array_equal(A, A.T)
            assert np.max(np.abs(S - S0)) <= 1e-12 * max(1.0, np.max(np.abs(S0)))
        G = -np.identity(3)
        A, S = np.empty((1, 1)), np.empty((2, 1))
        assert lib.ttm_separable_reduce_l2(ctypes.c_void_p(G.ctypes.data), 2, 1, 0.05, ctypes.c_void_p(A.ctypes.data), ctypes.c_void_p(S.ctypes.data)) != 0
