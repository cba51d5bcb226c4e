"""
The univariate form ("U-form", csrc/ttm_uform.h) of separable maps: spline geometry, verified fit errors,
agreement with the direct evaluation and with the oracle, and the fall-backs (too many spline columns, a
rejected fit).  Runs on the host test double (same builder / evaluator bodies as the kernels) and, marked
`gpu`, through libttm.so.
"""
import os

import numpy as np
import pytest

from tests.hostemu import emu
from tests.test_hostemu_vs_oracle import build
from tests.util import SEPARABLE, case_X, coeff_lists, ctor_kwargs, load_case, make_oracle, relerr
from triangular_transport_toolbox_amd import specs, termtable

U_CASES = ['c2b_sep', 'c3_sep', 'c5_sep']


@pytest.fixture(params=['hostemu', pytest.param('hip', marks=pytest.mark.gpu)])
def backend(request):
    if request.param == 'hostemu':
        with emu.install():
            yield 'hostemu'
    else:
        yield 'hip'


@pytest.mark.parametrize('name', SEPARABLE)
def test_geometry_and_eligibility(name):
    npz, desc, om, cm, em, _ = build(name)
    em.pack(om.coeffs_nonmon, om.coeffs_mon)
    if name == 'misc_sep':                      # special terms in the nonmonotone list: generic interpreter, no U-form
        assert not cm.u_static and not cm.u_enabled
        return
    assert cm.u_static and cm.u_enabled
    uc = cm.ucomp[:cm.D * termtable.UC_LEN].reshape(-1, termtable.UC_LEN)
    geo = cm.ugeo.reshape(-1, 2)
    for k in range(cm.D):
        nI = int(uc[k, 4])
        assert nI % 2 == 0 and 4 <= nI <= termtable.U_NI_MAX
        # support covers every special term, interval width <= kappa * smallest scale
        base = int(cm.dpar_off[k])
        mu = np.asarray([cm.dpar[base + p0] for p0 in cm.u_info[k]['st_p0']])
        sc = np.asarray([cm.dpar[base + p0 + 1] for p0 in cm.u_info[k]['st_p0']])
        t_lo, h = geo[k]
        t_hi = t_lo + (nI - 2) * h
        assert t_lo <= np.min(mu - termtable.U_SUPPORT * sc) + 1e-12 and t_hi >= np.max(mu + termtable.U_SUPPORT * sc) - 1e-12
        assert h <= termtable.U_KAPPA * sc.min() * (1 + 1e-12)
    # table offsets: 16-byte units, no overlap
    offs = sorted((int(uc[k, 5]), termtable.U_TSTRIDE * int(uc[k, 4])) for k in range(cm.D))
    for (o1, n1), (o2, _) in zip(offs, offs[1:]):
        assert o1 % 2 == 0 and o1 + n1 <= o2


@pytest.mark.parametrize('name', U_CASES)
def test_fit_errors_and_direct_agreement(name):
    npz, desc, om, cm, em, _ = build(name)
    coef = em.pack(om.coeffs_nonmon, om.coeffs_mon)
    err = em.uform_errors()
    assert err.shape == (cm.D, 2)
    assert err[:, 0].max() < 2e-14 and err[:, 1].max() < 2e-11           # value / derivative, relative to 1 + |exact|
    rng = np.random.default_rng(11)
    X = np.vstack((om.X[:300], rng.standard_normal((200, cm.d_cols)) * 3.0,      # far tails: the linear spline columns
                   rng.standard_normal((20, cm.d_cols)) * 30.0))
    Zu, ldu = em.forward(coef, X)
    os.environ['TTM_NO_UFORM'] = '1'
    try:
        Zd, ldd = em.forward(coef, X)
    finally:
        del os.environ['TTM_NO_UFORM']
    assert relerr(Zu, Zd) < 1e-12
    ok = np.isfinite(ldd)
    assert np.array_equal(np.isfinite(ldu), ok) and relerr(ldu[ok], ldd[ok]) < 1e-10
    # generic U-form evaluator (per-group degrees) == hot records (fixed degrees), where the map has them
    if cm.u_h_cls:
        os.environ['TTM_EMU_NO_HOT'] = '1'
        try:
            Zg, ldg = em.forward(coef, X)
        finally:
            del os.environ['TTM_EMU_NO_HOT']
        assert np.array_equal(Zg, Zu) and np.array_equal(ldg[ok], ldu[ok])


def test_monomial_conversion_all_families():
    """Row n of umono = monomial coefficients of P_n for every polynomial family the reference offers."""
    x = np.linspace(-2.5, 2.5, 41)
    for ptype, (fam, polyclass) in termtable.FAMILIES.items():
        mon = [[[0]]]
        non = [[[]]]
        cm = termtable.compile_map(mon, non, 1, ptype, 'separable monotonicity')
        M = cm.umono.reshape(termtable.U_PMAX + 1, termtable.U_PMAX + 1)
        for n in range(termtable.U_PMAX + 1):
            ref = polyclass([0.] * n + [1.])(x)
            got = sum(M[n, j] * x ** j for j in range(termtable.U_PMAX + 1))
            assert np.max(np.abs(got - ref)) < 1e-10 * (1 + np.max(np.abs(ref)))


def _narrow_map(scale_factor):
    """d = 3 separable map with special terms; ST_scale_factor controls the spline resolution needed."""
    mon, non = specs.banded_separable_spec(3, band=2)
    return mon, non, dict(monotonicity='separable monotonicity', ST_scale_factor=scale_factor, verbose=False)


def test_falls_back_when_spline_would_be_too_fine(backend):
    """Tiny special-term scales need more spline columns than TTM_U_NI_MAX: the map has no U-form and runs on the
    direct kernels, with the same results as the oracle."""
    from triangular_transport_toolbox_amd.transport_map import transport_map
    from oracle.ttm_oracle import OracleMap
    rng = np.random.default_rng(5)
    X = rng.standard_normal((600, 3)) @ np.array([[1.0, 0.4, 0.0], [0.0, 1.0, 0.5], [0.0, 0.0, 1.0]])
    mon, non, kw = _narrow_map(0.02)
    tm = transport_map(X=X, monotone=mon, nonmonotone=non, **kw)
    assert tm._cm.u_static and not tm._cm.u_enabled
    om = OracleMap(X=X, monotone=mon, nonmonotone=non, **{k: v for k, v in kw.items() if k != 'verbose'})
    for k in range(3):
        tm.coeffs_mon[k] = om.coeffs_mon[k] = 0.3 + 0.1 * rng.random(len(tm.coeffs_mon[k]))
        tm.coeffs_nonmon[k] = om.coeffs_nonmon[k] = 0.2 * rng.standard_normal(len(tm.coeffs_nonmon[k]))
    assert relerr(tm.map(X), om.map(X)) < 1e-11
    # a wider scale factor brings the U-form back
    mon, non, kw = _narrow_map(1.0)
    tm2 = transport_map(X=X, monotone=mon, nonmonotone=non, **kw)
    assert tm2._cm.u_enabled


def test_rejected_fit_disables_uform(backend, monkeypatch, ttm_opt):
    """A spline outside the host's tolerance is rejected when the coefficients are packed: the fold is redone
    without the U section and the direct kernels run (bit-identical to a map that never had a U-form)."""
    from triangular_transport_toolbox_amd.transport_map import transport_map
    npz, desc = load_case('c3_sep')
    X = case_X('c3_sep', npz)[:500]
    kw = ctor_kwargs(desc)
    tm = transport_map(X=X, monotone=desc['monotone'], nonmonotone=desc['nonmonotone'], verbose=False, **kw)
    tm.coeffs_mon, tm.coeffs_nonmon = coeff_lists(npz, tm.D)
    assert tm._cm.u_enabled
    Zu = tm.map(X)
    assert tm.uform_fit_error[:, 0].max() < 2e-14
    ttm_opt('no_uform', int('1'))
    Zd = tm.map(X)
    ttm_opt('no_uform', 0)
    assert relerr(Zu, Zd) < 1e-12
    monkeypatch.setattr(termtable, 'U_TOL_VALUE', 0.0)          # nothing passes
    tm._refresh_uform()
    assert tm._cm.u_enabled
    Zr = tm.map(X)
    assert not tm._cm.u_enabled and tm._prog.u_enabled == 0
    assert np.array_equal(Zr, Zd)
    # a new special-term placement gives the U-form another chance
    monkeypatch.undo()
    tm.reset(X)
    assert tm._cm.u_enabled


@pytest.mark.parametrize('name', U_CASES)
def test_class_results_match_oracle_through_uform(backend, name):
    from triangular_transport_toolbox_amd.transport_map import transport_map
    npz, desc = load_case(name)
    X = case_X(name, npz)
    tm = transport_map(X=X, monotone=desc['monotone'], nonmonotone=desc['nonmonotone'], verbose=False, **ctor_kwargs(desc))
    tm.coeffs_mon, tm.coeffs_nonmon = coeff_lists(npz, tm.D)
    om = make_oracle(name, npz, desc)
    assert tm._cm.u_enabled
    Xq = X[:400]
    assert relerr(tm.map(Xq), om.map(Xq)) < 1e-12
    assert relerr(tm.evaluate_pullback_density(Xq), om.evaluate_pullback_density(Xq)) < 1e-10
    Zin = npz['inv_Z']
    assert relerr(tm.inverse_map(Zin), om.inverse_map(Zin)) < 1e-10


@pytest.mark.parametrize('name', ['c5_sep', 'c2b_sep'])
def test_conditional_inverse_and_partial_sweeps_on_loader_kernels(backend, name, monkeypatch, ttm_opt):
    """Sweeps that start inside the map (conditional inverse with X_star on a map without skipped dimensions,
    s() of a single component): the hot-record kernels preload the planned cache from the entry state of that
    component.  Forced onto the loader-wave kernels (they are chosen by themselves only for large ensembles)."""
    from triangular_transport_toolbox_amd.transport_map import transport_map
    npz, desc = load_case(name)
    X = case_X(name, npz)
    tm = transport_map(X=X, monotone=desc['monotone'], nonmonotone=desc['nonmonotone'], verbose=False, **ctor_kwargs(desc))
    tm.coeffs_mon, tm.coeffs_nonmon = coeff_lists(npz, tm.D)
    om = make_oracle(name, npz, desc)
    ttm_opt('u_loader', int('1'))
    Xq = X[:333]
    E = 1 if tm.D == 2 else 7
    Zq = om.map(Xq)
    got = tm.inverse_map(Zq[:, E:], X_star=Xq[:, :E])
    ref = om.inverse_map(Zq[:, E:], X_star=Xq[:, :E])
    assert got.shape == ref.shape and relerr(got, ref) < 1e-10
    Xs = (Xq - om.X_mean) / om.X_std
    for k in sorted({0, tm.D // 2, tm.D - 1}):
        assert relerr(tm.s(Xs, k), om.s(Xs, k)) < 1e-12


def _synthetic_separable(D, band, hf_order, plain_order, n_irbf, family='hermite function', seed=0):
    """Banded separable map: per component and conditioning column j in the band one plain term of order 1..plain_order
    and Hermite-function terms of order 2..hf_order; monotone LET, n_irbf iRBF, RET (an RBF when n_irbf is odd)."""
    mon, non = [], []
    for k in range(D):
        nm = [[]]
        for j in range(max(0, k - band), k):
            for o in range(1, plain_order + 1):
                nm.append([j] * o)
            for o in range(2, hf_order + 1):
                nm.append([j] * o + ['HF'])
        non.append(nm)
        m = ['LET %d' % k] + ['iRBF %d' % k] * n_irbf + ['RET %d' % k]
        if n_irbf % 2:
            m.insert(1, 'RBF %d' % k)
        mon.append(m)
    return mon, non


@pytest.mark.parametrize('D,band,hf_order,plain_order,n_irbf,expect', [
    (6, 2, 3, 1, 2, (1, 2)),       # class (3,1), two group records  (the C5 shape)
    (7, 4, 3, 1, 1, (1, 4)),       # class (3,1), four group records
    (5, 2, 5, 3, 3, (2, 2)),       # class (5,5)
    (6, 3, 7, 6, 2, (3, 4)),       # class (7,7), four group records
    (5, 1, 4, 1, 0, (2, 2)),       # one group per component (padded records), LET + RET only
])
def test_every_hot_kernel_class_against_the_oracle(backend, monkeypatch, D, band, hf_order, plain_order, n_irbf, expect, ttm_opt):
    """Every degree class / record count of the hot-record kernels (k_forward_hl, k_inverse_hl), forced onto the
    loader-wave path at small N, against the oracle: map, pullback density, table inverse, conditional inverse."""
    from triangular_transport_toolbox_amd.transport_map import transport_map
    from oracle.ttm_oracle import OracleMap
    rng = np.random.default_rng(100 * D + 10 * band + hf_order)
    L = np.tril(rng.standard_normal((D, D)) * 0.4) + np.eye(D)
    X = rng.standard_normal((1537, D)) @ L.T + 0.3 * rng.standard_normal((1537, D)) ** 2
    mon, non = _synthetic_separable(D, band, hf_order, plain_order, n_irbf)
    kw = dict(monotonicity='separable monotonicity')
    tm = transport_map(X=X, monotone=mon, nonmonotone=non, verbose=False, **kw)
    om = OracleMap(X=X, monotone=mon, nonmonotone=non, **kw)
    for k in range(D):
        cm_ = 0.2 + 0.5 * rng.random(len(tm.coeffs_mon[k]))
        cn_ = 0.3 * rng.standard_normal(len(tm.coeffs_nonmon[k])) / (1 + np.arange(len(tm.coeffs_nonmon[k])))
        tm.coeffs_mon[k], om.coeffs_mon[k] = cm_.copy(), cm_.copy()
        tm.coeffs_nonmon[k], om.coeffs_nonmon[k] = cn_.copy(), cn_.copy()
    assert tm._cm.u_enabled
    assert (tm._cm.u_h_cls, tm._cm.u_h_ng) == expect
    ttm_opt('u_loader', int('1'))
    for ns in ('2', '4'):
        ttm_opt('hl_ns', int(ns))
        Z = tm.map(X)
        assert relerr(Z, om.map(X)) < 1e-11
        with np.errstate(all='ignore'):
            pref = om.evaluate_pullback_density(X[:300])
        pgot = tm.evaluate_pullback_density(X[:300])
        ok = np.isfinite(pref)                       # (an RBF term can make dS/dx negative: log -> NaN on both sides)
        assert np.array_equal(np.isfinite(pgot), ok) and relerr(pgot[ok], pref[ok]) < 1e-9
        Zin = rng.standard_normal((700, D))
        assert relerr(tm.inverse_map(Zin), om.inverse_map(Zin)) < 1e-9
        E = D // 2
        Zq = om.map(X[:200])
        assert relerr(tm.inverse_map(Zq[:, E:], X_star=X[:200, :E]), om.inverse_map(Zq[:, E:], X_star=X[:200, :E])) < 1e-9
    # windowed tables of the resident-table inverse (only the middle of every table in LDS, the rest searched in
    # memory): the same bits as with whole tables, whatever the window and however many rows fall outside it
    Zw = rng.standard_normal((1500, D))
    Zw[:40] *= 4.0
    Zw[40] = np.nan
    full = tm.inverse_map(Zw)
    keep = np.arange(len(Zw)) != 40
    assert relerr(full[keep], om.inverse_map(Zw[keep])) < 1e-9 and np.all(np.isnan(full[40]))
    for rows in (2, 4):
        ttm_opt('rt_ns', rows)
        for window in (0, 40, 300, 650, 990):
            ttm_opt('rt_window', window)
            assert np.array_equal(tm.inverse_map(Zw), full, equal_nan=True), (rows, window)
    assert tm.uform_fit_error[:, 0].max() < 5e-14


@pytest.mark.parametrize('case', ['dense', 'own_terms', 'skip_dims', 'other_family'])
def test_generic_uform_kernels_against_the_oracle(backend, monkeypatch, case, ttm_opt):
    """U-form maps WITHOUT hot records (cache misses of dense maps, polynomial / Hermite-function terms in the
    monotone list, skipped dimensions, another polynomial family): the generic U-form kernels (k_forward_u,
    loader-wave k_forward_ul) and the direct inverse, against the oracle."""
    from triangular_transport_toolbox_amd.transport_map import transport_map
    from oracle.ttm_oracle import OracleMap
    rng = np.random.default_rng({'dense': 1, 'own_terms': 2, 'skip_dims': 3, 'other_family': 4}[case])
    kw = dict(monotonicity='separable monotonicity')
    if case == 'dense':
        D, d = 6, 6
        mon, non = specs.dense_separable_spec(D, 3)
    elif case == 'own_terms':
        D, d = 4, 4
        mon, non = _synthetic_separable(D, 2, 3, 1, 1)
        for k in range(D):
            mon[k] = [[k]] + mon[k] + [[k, k, k, 'HF']]            # linear + a Hermite function of x_k in the monotone list
    elif case == 'skip_dims':
        D, d = 3, 4
        mon, non = specs.entf_filter_spec(3)
        kw.update(regularization='l2', regularization_lambda=0.05)
    else:
        D, d = 4, 4
        mon, non = _synthetic_separable(D, 2, 1, 3, 2)              # plain polynomial terms only
        kw.update(polynomial_type='legendre')
    X = rng.standard_normal((1200, d)) @ (np.tril(rng.standard_normal((d, d)) * 0.3) + np.eye(d)).T
    tm = transport_map(X=X, monotone=mon, nonmonotone=non, verbose=False, **kw)
    om = OracleMap(X=X, monotone=mon, nonmonotone=non, **kw)
    for k in range(D):
        cm_ = 0.2 + 0.5 * rng.random(len(tm.coeffs_mon[k]))
        cn_ = 0.3 * rng.standard_normal(len(tm.coeffs_nonmon[k])) / (1 + np.arange(len(tm.coeffs_nonmon[k])))
        tm.coeffs_mon[k], om.coeffs_mon[k] = cm_.copy(), cm_.copy()
        tm.coeffs_nonmon[k], om.coeffs_nonmon[k] = cn_.copy(), cn_.copy()
    # (no hot records, or hot records that only feed the push records of a map with a few components: the kernels here
    # are the generic ones either way)
    assert tm._cm.u_enabled and (tm._cm.u_h_cls == 0 or tm._cm.u_p_lag == 3)
    for loader in ('0', '1'):
        ttm_opt('u_loader', int(loader))
        assert relerr(tm.map(X), om.map(X)) < 1e-11
        with np.errstate(all='ignore'):
            pref = om.evaluate_pullback_density(X[:200])
        pgot = tm.evaluate_pullback_density(X[:200])
        ok = np.isfinite(pref)
        assert np.array_equal(np.isfinite(pgot), ok) and relerr(pgot[ok], pref[ok]) < 1e-9
    Zin = rng.standard_normal((500, D))
    Xstar = X[:500, :d - D] if d > D else None
    assert relerr(tm.inverse_map(Zin, X_star=Xstar), om.inverse_map(Zin, X_star=Xstar)) < 1e-9
