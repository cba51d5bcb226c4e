"""
Push-form kernels of banded U-form maps (csrc/ttm_band.hip: k_band_forward, k_band_density, k_band_inverse) against the
oracle and against the kernels they replace (k_forward_hl, k_inverse_rt), at sizes the oracle finishes in seconds:
every degree class, sweeps that start inside the map (the columns in front are pushed first), chunks of
several tiles, several blocks of resident tables, rows the resident paths hand to their exact fall-backs.
The host side of it (band detection, push-record geometry) runs on CPU.
"""
import ctypes

import numpy as np
import pytest

from tests.test_uform import _synthetic_separable
from tests.util import relerr


def _banded_with_conditioning(D, skip, band=2):
    """D components behind `skip` conditioning columns: component k lives on column k + skip and reads the `band` columns
    in front of it, conditioning columns included."""
    mon, non = [], []
    for k in range(D):
        kc = k + skip
        nm = [[]]
        for j in range(max(0, kc - band), kc):
            nm += [[j], [j, j, 'HF'], [j, j, j, 'HF']]
        non.append(nm)
        mon.append(['LET %d' % kc, 'iRBF %d' % kc, 'iRBF %d' % kc, 'RET %d' % kc])
    return mon, non


CASES = {
    'c5_shape': dict(D=6, d=6, spec=lambda: _synthetic_separable(6, 2, 3, 1, 2), cls=1),
    'class_55': dict(D=5, d=5, spec=lambda: _synthetic_separable(5, 2, 5, 3, 3), cls=2),
    'class_77': dict(D=5, d=5, spec=lambda: _synthetic_separable(5, 2, 7, 6, 2), cls=3),
    'lag_one': dict(D=5, d=5, spec=lambda: _synthetic_separable(5, 1, 4, 1, 0), cls=2),
    # maps of a few components (k_band_few, the KM = 4 inverse): C2b, C3 (a group three columns back), the filter's shape
    'few_c2b': dict(D=2, d=2, spec=lambda: _specs().temperature_spec(5), cls=2, few=True),
    'few_c3': dict(D=4, d=4, spec=lambda: _specs().dense_separable_spec(4, 4), cls=2, lag=3, few=True),
    'few_cond': dict(D=3, d=4, spec=lambda: _banded_with_conditioning(3, 1, band=3), cls=1, lag=3, few=True),
    # linear terms of the own variable next to / instead of the special terms: the filter map of example 06, the map of example 05
    'few_entf': dict(D=3, d=4, spec=lambda: _specs().entf_filter_spec(3), cls=1, lag=3, few=True),
    'few_ex05': dict(D=2, d=2, spec=lambda: _specs().density_example_spec(3), cls=1, lag=3, few=True),
    'few_cond2': dict(D=4, d=6, spec=lambda: _banded_with_conditioning(4, 2), cls=1, lag=3, few=True),   # (lag 2, but conditioning columns: records of three groups)
    # the smoother's block map of example 07 (6 columns, D = 3, skip 3: its third component reads five columns back, five groups;
    # linear monotone parts, no spline anywhere): push records of five groups
    'few_ents': dict(D=3, d=6, spec=lambda: _specs().ents_smoother_spec(3), cls=1, lag=5, few=True, kw=dict(polynomial_type="probabilist's hermite")),
}


def _specs():
    from triangular_transport_toolbox_amd import specs
    return specs


def _build(case, n=5003, seed=0):
    from triangular_transport_toolbox_amd.transport_map import transport_map
    from oracle.ttm_oracle import OracleMap
    c = CASES[case]
    rng = np.random.default_rng(seed + 17 * c['D'])
    d = c['d']
    X = rng.standard_normal((n, d)) @ (np.tril(rng.standard_normal((d, d)) * 0.4) + np.eye(d)).T + 0.3 * rng.standard_normal((n, d)) ** 2
    mon, non = c['spec']()
    kw = dict(monotonicity='separable monotonicity', **c.get('kw', {}))
    tm = transport_map(X=X, monotone=mon, nonmonotone=non, verbose=False, **kw)
    om = OracleMap(X=X, monotone=mon, nonmonotone=non, **kw)
    for k in range(c['D']):
        cm_ = 0.2 + 0.5 * rng.random(len(tm.coeffs_mon[k]))
        cn_ = 0.3 * rng.standard_normal(len(tm.coeffs_nonmon[k])) / (1 + np.arange(len(tm.coeffs_nonmon[k])))
        tm.coeffs_mon[k], om.coeffs_mon[k] = cm_.copy(), cm_.copy()
        tm.coeffs_nonmon[k], om.coeffs_nonmon[k] = cn_.copy(), cn_.copy()
    return tm, om, X, rng


@pytest.mark.parametrize('case', sorted(CASES))
def test_band_detection_and_push_record_geometry(case):
    """Host side: which maps are banded, and where their push records live in the U section."""
    from tests.hostemu import emu
    from triangular_transport_toolbox_amd import termtable
    with emu.install():
        tm, om, X, rng = _build(case, n=400)
        cm = tm._cm
        assert cm.u_enabled and cm.u_h_cls == CASES[case]['cls']
        assert cm.u_p_lag == CASES[case].get('lag', 2) and termtable.P_LAG_MAX == 5
        gp = termtable.H_DB[cm.u_h_cls] + 1 + termtable.H_DA[cm.u_h_cls]
        assert cm.u_p_stride % 8 == 0
        assert cm.u_p_off % 8 == 0 and cm.u_p_off >= cm.u_h_off + cm.D * (termtable.H_HDR + cm.u_h_ng * termtable.H_GS[cm.u_h_cls])
        assert cm.u_p_stride >= termtable.P_HDR + cm.u_p_lag * gp
        assert cm.u_size >= cm.u_p_off + (cm.D + cm.u_p_lag) * cm.u_p_stride


def test_maps_that_are_not_banded_have_no_push_records():
    from tests.hostemu import emu
    from triangular_transport_toolbox_amd import specs
    from triangular_transport_toolbox_amd.transport_map import transport_map
    with emu.install():
        rng = np.random.default_rng(2)
        for mon, non, d in (_synthetic_separable(6, 3, 3, 1, 2) + (6,),          # a group three columns back
                            specs.dense_separable_spec(5, 3) + (5,),
                            _banded_with_conditioning(5, 2) + (7,),              # conditioning columns: no hot records (cache misses), more than a few components
                            _banded_with_conditioning(5, 1, band=3) + (6,)):     # three columns back, more than a few components
            X = rng.standard_normal((300, d))
            tm = transport_map(X=X, monotone=mon, nonmonotone=non, verbose=False, monotonicity='separable monotonicity')
            assert tm._cm.u_p_lag == 0


@pytest.mark.gpu
@pytest.mark.parametrize('case', sorted(CASES))
def test_band_kernels_against_the_oracle_and_the_kernels_they_replace(case, ttm_opt):
    tm, om, X, rng = _build(case)
    lib = tm._lib
    lib.ttm_last_kernel.restype = ctypes.c_char_p
    D, d = CASES[case]['D'], CASES[case]['d']
    few = CASES[case].get('few', False)
    E = d - D
    N = len(X)
    Zo = om.map(X)
    with np.errstate(all='ignore'):
        pref = om.evaluate_pullback_density(X[:400])
    ok = np.isfinite(pref)
    Zin = rng.standard_normal((N, D))
    Zin[:40] *= 3.5                                         # (targets beyond the resident window and beyond the tables)
    Xstar = X[:, :E] if E else None
    Xo = om.inverse_map(Zin, X_star=Xstar)
    Xs = (X - om.X_mean) / om.X_std
    # the kernels the band kernels replace, forced onto their large-ensemble paths
    ttm_opt('u_loader', 1); ttm_opt('band_fwd', 0); ttm_opt('band_inv', 0)
    Zh = tm.map(X)
    Xh = tm.inverse_map(Zin, X_star=Xstar)
    ph = tm.evaluate_pullback_density(X[:400])
    for cus, block in ((-1, -1), (1, -1), (3, 2), (2, 1)):    # one tile per chunk | several tiles | several blocks
        ttm_opt('band_fwd', 1); ttm_opt('band_inv', 1); ttm_opt('band_cus', cus); ttm_opt('rt_block', block)
        Z = tm.map(X)
        tm.forward_device(tm._Xs, tm._N)
        assert lib.ttm_last_kernel().decode() == ('k_band_few' if few else 'k_band_forward')
        assert relerr(Z, Zo) < 1e-11, (cus, block)
        assert relerr(Z, Zh) < 1e-13
        pgot = tm.evaluate_pullback_density(X[:400])
        assert np.array_equal(np.isfinite(pgot), ok) and relerr(pgot[ok], pref[ok]) < 1e-10 and relerr(pgot[ok], ph[ok]) < 1e-12
        for k in sorted({0, D // 2, D - 1}):                  # sweeps that start inside the map: the columns in front are pushed first
            assert relerr(tm.s(Xs, k), om.s(Xs, k)) < 1e-12
        Xi = tm.inverse_map(Zin, X_star=Xstar)
        assert relerr(Xi, Xo) < 1e-11, (cus, block)
        # (k_inverse_rt: another summation order of the offsets, which the degree-7 groups of class (7,7) amplify to 2e-13;
        # since inverse_map() changes layout on the device - an even leading dimension for odd N too - this line compares
        # the two large-ensemble kernels; with the host transpose of round 3 both sides had silently run the generic one)
        assert relerr(Xi, Xh) < 1e-12
        tm.inverse_device(tm._cols(D, tm._N, zero=True), tm._N)
        if case != 'class_55':                                # (its RBF term makes the tables non-monotone: sorted on the host, generic lookup)
            assert lib.ttm_last_kernel().decode() == ('k_band_few_inverse' if few else 'k_band_inverse')
    # the bits do not depend on how the rows are cut into chunks and tiles or the components into blocks
    ttm_opt('band_cus', -1); ttm_opt('rt_block', -1)
    Z0, X0 = tm.map(X), tm.inverse_map(Zin, X_star=Xstar)
    for cus, block in ((1, -1), (3, 2), (2, 1)):
        ttm_opt('band_cus', cus); ttm_opt('rt_block', block)
        assert np.array_equal(tm.map(X), Z0)
        Xi = tm.inverse_map(Zin, X_star=Xstar)
        # (a tile that re-reads its columns at a block boundary takes exp(-x^2/4) there from the series, not the interval)
        assert relerr(Xi, X0) < 1e-14


@pytest.mark.gpu
def test_band_kernels_on_rows_outside_their_resident_paths(ttm_opt):
    """NaN, infinities and |x| far beyond the E table / the table windows: the forward map holds exp(-x^2/4) at 1.6e-28
    beyond 16 standard deviations, the inverse hands the row to the search in memory."""
    tm, om, X, rng = _build('c5_shape', n=3001)
    ttm_opt('u_loader', 1); ttm_opt('band_fwd', 1); ttm_opt('band_inv', 1)
    Xb = X.copy()
    Xb[5] = X[5] + 40.0 * om.X_std                          # 40 standard deviations out
    Xb[6, 2] = np.nan
    Xb[7, 1] = np.inf
    with np.errstate(all='ignore'):
        Zo = om.map(Xb)
    Z = tm.map(Xb)
    fin = np.isfinite(Zo)
    assert np.array_equal(np.isfinite(Z), fin)              # a NaN / inf reaches exactly the components that read its column
    assert relerr(Z[fin], Zo[fin]) < 1e-11
    assert np.all(np.isnan(Z[6, 2:5])) and np.all(np.isfinite(Z[6, :2])) and np.isfinite(Z[6, 5])
    Zin = rng.standard_normal((len(X), tm.D))
    Zin[3] = 50.0; Zin[4] = -50.0; Zin[8, 0] = np.nan
    with np.errstate(all='ignore'):
        Xo = om.inverse_map(Zin)
    Xi = tm.inverse_map(Zin)
    keep = np.ones(len(X), bool); keep[8] = False
    assert relerr(Xi[keep], Xo[keep]) < 1e-11
    assert np.all(np.isnan(Xi[8]))


@pytest.mark.gpu
def test_deferred_checks_give_the_same_results_and_catch_what_the_eager_checks_catch(ttm_opt, monkeypatch):
    """`deferred_checks`: no host visit behind a new coefficient vector; validate() reads the flags later.  Same bits as
    the eager path when the checks hold; when they do not (a table that is not sorted; a spline outside the
    tolerance) validate() says so and repairs the state."""
    import torch
    from triangular_transport_toolbox_amd import termtable
    tm, om, X, rng = _build('c5_shape', n=3001)
    N = tm._N
    Zin = rng.standard_normal((N, tm.D))
    Z0 = tm.forward_device(tm._Xs, N).clone()
    Zd = tm._cols(tm.D, N)
    Zd[:, :N] = tm._to_dev(np.ascontiguousarray(Zin.T))
    X0 = tm.inverse_device(Zd, N).clone()
    tm.deferred_checks = True
    tm._pack_memo = None                                      # a new coefficient vector
    coef = tm._pack_coeffs()
    assert getattr(coef, '_ttm_pending', None) is not None
    Z1 = tm.forward_device(tm._Xs, N, coef=coef).clone()
    X1 = tm.inverse_device(Zd, N, coef=coef).clone()
    assert tm.validate(coef) and tm.validate(coef)            # (a second call has nothing left to read)
    assert np.array_equal(Z1[:, :N].cpu().numpy(), Z0[:, :N].cpu().numpy())
    assert np.array_equal(X1[:, :N].cpu().numpy(), X0[:, :N].cpu().numpy())
    # a table reported as not sorted: validate() says so, later inversions take the sorted lookup of the eager path
    tm._pack_memo = None
    coef = tm._pack_coeffs()
    torch.cuda.synchronize()
    coef._ttm_pending[3][1] = 1                               # (the pinned copy of the flags: table 1 "not sorted")
    assert tm.validate(coef) is False
    assert [e[4] for e in coef._ttm_tables.values()] == [False]
    X2 = tm.inverse_device(Zd, N, coef=coef)
    lib = tm._lib
    lib.ttm_last_kernel.restype = ctypes.c_char_p
    assert lib.ttm_last_kernel().decode() == 'k_inverse_table'
    assert relerr(X2[:, :N].cpu().numpy(), X0[:, :N].cpu().numpy()) < 1e-13
    # a spline outside the tolerance: found at validate(), U-form switched off, the vector folded again
    tm, om, X, rng = _build('c5_shape', n=3001)
    N = tm._N
    Zref = om.map(X)
    tm.deferred_checks = True
    monkeypatch.setattr(termtable, 'U_TOL_VALUE', 0.0)
    tm._pack_memo = None
    coef = tm._pack_coeffs()
    tm.forward_device(tm._Xs, N, coef=coef)
    assert tm.validate(coef) is False and not tm._cm.u_enabled
    Z = tm.forward_device(tm._Xs, N, coef=coef)
    assert relerr(Z[:, :N].cpu().numpy().T, Zref) < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['c5_shape', 'few_c2b'])
def test_density_pass_with_negative_derivatives_is_nan_where_the_reference_is(case, ttm_opt):
    """The density kernels take ONE logarithm of the product of a row's derivatives: a negative derivative must still make
    the row NaN (np.log of a negative number, TM:2690-2700), also when two of them would cancel in the product."""
    tm, om, X, rng = _build(case, n=3001)
    for k in range(min(3, tm.D)):                             # LET terms with negative coefficients: dS_k/dx_k < 0 in the left tail
        tm.coeffs_mon[k][0] = om.coeffs_mon[k][0] = -0.6
    ttm_opt('u_loader', 1); ttm_opt('band_fwd', 1)
    with np.errstate(all='ignore'):
        ref = om.evaluate_pullback_density(X)
    got = tm.evaluate_pullback_density(X)
    bad = ~np.isfinite(ref)
    assert bad.sum() > 20 and (~bad).sum() > 20               # (the case has rows of both kinds)
    assert np.array_equal(~np.isfinite(got), bad)
    assert relerr(got[~bad], ref[~bad]) < 1e-10


@pytest.mark.gpu
def test_hot_record_kernels_do_not_pick_up_what_an_earlier_launch_left_in_lds(ttm_opt):
    """A column that only plain-polynomial groups read is put into the planned cache without exp(-x^2/4) (nobody needs it);
    the hot-record evaluators read both halves of a cache slot and multiply the second by the group's Hermite-function
    polynomial - zero here.  That half must be DEFINED: after a table inverse (+inf sentinels behind its resident tables) the
    same LDS bytes held +inf, and 0 * inf put NaN into a few rows of k_forward_hl's output (found by tools/fuzz_few.py,
    seed 327).  The sequence of that run: inverse, then the forward map through the hot-record kernel."""
    from triangular_transport_toolbox_amd.transport_map import transport_map
    from oracle.ttm_oracle import OracleMap
    rng = np.random.default_rng(327)
    mon = [['LET 1', 'iRBF 1', 'iRBF 1', 'iRBF 1', 'RET 1'], ['LET 2', 'iRBF 2', 'iRBF 2', 'RET 2']]
    non = [[[]], [[], [1]]]
    kw = dict(monotonicity='separable monotonicity', polynomial_type="probabilist's hermite")
    n = 5003
    X = rng.standard_normal((n, 3)) @ (np.tril(rng.standard_normal((3, 3)) * 0.4) + np.eye(3)).T
    tm = transport_map(X=X, monotone=mon, nonmonotone=non, verbose=False, **kw)
    om = OracleMap(X=X, monotone=mon, nonmonotone=non, **kw)
    for k in range(2):
        c = 0.2 + 0.5 * rng.random(len(tm.coeffs_mon[k]))
        tm.coeffs_mon[k], om.coeffs_mon[k] = c.copy(), c.copy()
        c = 0.3 * rng.standard_normal(len(tm.coeffs_nonmon[k]))
        tm.coeffs_nonmon[k], om.coeffs_nonmon[k] = c.copy(), c.copy()
    Zo = om.map(X)
    lib = tm._lib
    lib.ttm_last_kernel.restype = ctypes.c_char_p
    for rep in range(3):
        ttm_opt('u_loader', 1); ttm_opt('band_fwd', 1); ttm_opt('band_inv', 1)
        tm.inverse_map(rng.standard_normal((n, 2)), X_star=X[:, :1])      # (+inf sentinels into LDS)
        ttm_opt('band_fwd', 0); ttm_opt('band_inv', 0)
        Z = tm.map(X)
        tm.forward_device(tm._Xs, tm._N)
        assert lib.ttm_last_kernel().decode() == 'k_forward_hl'
        assert np.isfinite(Z).all() and relerr(Z, Zo) < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize('case', sorted(CASES))
def test_fused_setup_launches_write_the_same_bits_as_the_separate_ones(case, ttm_opt):
    """A new coefficient vector's device setup is two launches - k_uform with the fold as its prologue and the push records
    scattered behind the hot record, k_table_build_index - instead of five (k_fold, k_uform, k_band_records, k_table_build,
    k_table_index; options fold_fused / table_fused = 0): the folded-coefficient buffer with its whole U section (hot
    and push records, splines with their local-coordinate offsets, fit errors) and the tables, ranges, bucket indices and
    sortedness flags are identical bit for bit."""
    import torch
    tm, om, X, rng = _build(case)
    assert tm._cm.u_enabled and tm._cm.u_p_lag > 0

    def setup():
        tm._pack_memo = None
        coef = tm._pack_coeffs()
        key, tabs = tm._launch_default_tables(coef)
        torch.cuda.synchronize()
        return coef._ttm_fold.clone(), [None if t is None else t.clone() for t in tabs], tm._lib.ttm_last_kernel().decode()
    fold_f, tabs_f, last_f = setup()
    assert last_f == 'k_table_build_index'
    ttm_opt('fold_fused', 0)
    ttm_opt('table_fused', 0)
    fold_s, tabs_s, last_s = setup()
    assert last_s == ('k_table_index' if tabs_s[5] is None else 'k_table_image')     # (resident-table images: maps of more than a few components)
    assert fold_f.shape == fold_s.shape
    same = (fold_f == fold_s) | (torch.isnan(fold_f) & torch.isnan(fold_s))
    assert bool(same.all()), 'fold / U section differs at %s' % torch.nonzero(~same).flatten()[:8].tolist()
    assert fold_f.view(torch.int64).equal(fold_s.view(torch.int64))                  # (packed int32 pairs included)
    for a, b in zip(tabs_f, tabs_s):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a.view(torch.int64) if a.dtype == torch.float64 else a, b.view(torch.int64) if b.dtype == torch.float64 else b)


@pytest.mark.gpu
def test_staged_fold_with_recycled_buffers_equals_the_copying_path(ttm_opt):
    """A new coefficient vector of a U-form map reaches the device without a copy in the stream (ttm_fold_staged: the fold kernel
    reads the packed vector from page-locked host memory; fit errors and table flags come back the same way) and takes over
    the fold buffer of a vector that is gone without a zero-fill.  Twelve vectors in a row (the ring has eight slots): each
    fold buffer, the device copy of the vector, the tables and the checks are those of the copying path, bit for bit."""
    import torch
    tm, om, X, rng = _build('c5_shape')
    tm.inverse_map(rng.standard_normal((8, tm.D)))           # (tables are built with the fold from the first inversion on)
    base_mon = [c.copy() for c in tm.coeffs_mon]
    for i in range(12):
        for k in range(tm.D):
            tm.coeffs_mon[k] = base_mon[k] * (1.0 + 0.01 * (i + 1))
        tm._pack_memo = None
        coef = tm._pack_coeffs()
        assert tm._lib.ttm_last_kernel().decode() == 'k_setup'                       # (fold + U section + tables: one launch)
        torch.cuda.synchronize()
        got = (coef.clone(), coef._ttm_fold.clone(), [t.clone() if hasattr(t, 'clone') else t for t in list(coef._ttm_tables.values())[0]])
        del coef
        ttm_opt('fold_fused', 0)                              # -> ttm_fold_staged declines: H2D copy + k_fold + k_uform + k_band_records
        tm._pack_memo = None
        ref = tm._pack_coeffs()
        torch.cuda.synchronize()
        assert torch.equal(got[0], ref)
        assert got[1].view(torch.int64).equal(ref._ttm_fold.view(torch.int64))
        for a, b in zip(got[2], list(ref._ttm_tables.values())[0]):
            assert (torch.equal(a, b) if hasattr(a, 'clone') else a == b)
        del ref
        ttm_opt('fold_fused', -1)


def _ring_map(D, n=5003, seed=3, shape=(3, 1, 2), nm_scale=0.3):
    """A banded map of D components (shape: Hermite-function order, plain order, iRBF terms - (3, 1, 2) is the C5 shape) with
    moderate random coefficients (tables monotone), device map + oracle."""
    from triangular_transport_toolbox_amd.transport_map import transport_map
    from oracle.ttm_oracle import OracleMap
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, D)) * (1.0 + 0.3 * rng.random(D)) + 0.2 * rng.standard_normal((n, 1))
    mon, non = _synthetic_separable(D, 2, *shape)
    kw = dict(monotonicity='separable monotonicity')
    tm = transport_map(X=X, monotone=mon, nonmonotone=non, verbose=False, **kw)
    om = OracleMap(X=X, monotone=mon, nonmonotone=non, **kw)
    for k in range(D):
        cm_ = 0.2 + 0.5 * rng.random(len(tm.coeffs_mon[k]))
        cn_ = nm_scale * rng.standard_normal(len(tm.coeffs_nonmon[k])) / (1 + np.arange(len(tm.coeffs_nonmon[k])))
        tm.coeffs_mon[k], om.coeffs_mon[k] = cm_.copy(), cm_.copy()
        tm.coeffs_nonmon[k], om.coeffs_nonmon[k] = cn_.copy(), cn_.copy()
    return tm, om, X, rng


@pytest.mark.gpu
@pytest.mark.parametrize('D', [10, 26, 27])
def test_ring_inverse_equals_the_block_inverse_and_the_oracle(D, ttm_opt):
    """k_band_inverse_ring (resident-table images copied into a ring of LDS slots by DMA, csrc/ttm_band_image.h) against
    k_band_inverse (tables assembled per workgroup and block) and the oracle: every component resident (no refill), rings of
    24 slots refilled 8 at a time and of 12 / 16 slots refilled 4 at a time, an odd number of components, chunks of one and
    of several tiles, targets beyond the resident window and beyond the tables."""
    tm, om, X, rng = _ring_map(D)
    lib = tm._lib
    lib.ttm_last_kernel.restype = ctypes.c_char_p
    N = len(X)
    Zin = rng.standard_normal((N, D))
    Zin[:40] *= 3.5
    Zin[41] = 60.0; Zin[42] = -60.0
    Xo = om.inverse_map(Zin)
    ttm_opt('u_loader', 1); ttm_opt('band_inv', 1)

    def run(ring, cus, block):
        ttm_opt('band_ring', ring); ttm_opt('band_cus', cus); ttm_opt('rt_block', block)
        tm._pack_memo = None                                # (a fresh coefficient vector: tables and images under these options)
        Xi = tm.inverse_map(Zin)
        tm.inverse_device(tm._cols(D, tm._N, zero=True), tm._N)
        return Xi, lib.ttm_last_kernel().decode()

    X_block, name = run(0, -1, -1)
    assert name == 'k_band_inverse' and relerr(X_block, Xo) < 1e-11
    X_ring, name = run(1, -1, -1)
    assert name == 'k_band_inverse_ring'
    assert relerr(X_ring, Xo) < 1e-11
    # one tile per chunk: both kernels carry the running sums in registers through all columns - the same bits
    assert np.array_equal(X_ring, X_block)
    for cus, block in ((-1, 12), (-1, 16), (1, -1), (1, 12), (2, 16)):
        Xi, name = run(1, cus, block)
        expect_ring = D > 4
        assert name == ('k_band_inverse_ring' if expect_ring else 'k_band_inverse'), (cus, block, name)
        # the ring kernel's result does not depend on how rows are cut into chunks and tiles or on the ring's size
        assert np.array_equal(Xi, X_ring), (cus, block)
        Xb, name = run(0, cus, block)
        assert name == 'k_band_inverse' and relerr(Xb, X_ring) < 1e-14, (cus, block)
    # a sweep that starts inside the map (conditional inverse): the running sums start from the two columns in front
    E = 3
    Xc_o = om.inverse_map(Zin[:, E:], X_star=X[:, :E])
    got = {}
    for ring in (1, 0):
        ttm_opt('band_ring', ring); ttm_opt('band_cus', -1); ttm_opt('rt_block', -1)
        tm._pack_memo = None
        got[ring] = tm.inverse_map(Zin[:, E:], X_star=X[:, :E])
        assert relerr(got[ring], Xc_o) < 1e-11
    assert np.array_equal(got[1], got[0])


@pytest.mark.gpu
def test_table_images_fused_and_separate_launches_and_layout_mismatch(ttm_opt):
    """The resident-table images written by the fused table kernel equal those of the two-launch path bit for bit; an image
    laid out for another window than the lookup plans for is ignored (k_band_inverse runs), not misread."""
    import torch
    D = 26
    tm, om, X, rng = _ring_map(D, n=3001)
    lib = tm._lib
    lib.ttm_last_kernel.restype = ctypes.c_char_p
    ttm_opt('u_loader', 1); ttm_opt('band_inv', 1)
    Zin = rng.standard_normal((len(X), D))
    Xo = om.inverse_map(Zin)
    imgs = {}
    for fused in (1, 0):
        ttm_opt('table_fused', fused)
        tm._pack_memo = None
        coef = tm._pack_coeffs()
        tm._inverse_table(coef, 0, D, None, None, 0)
        entry = next(iter(coef._ttm_tables.values()))
        assert entry[5] is not None and entry[5].numel() == D * int(lib.ttm_inverse_table_image_doubles(tm._pp, 0, D, 1001, tm._inv_nb()))
        imgs[fused] = [t.clone() for t in entry[:4]] + [entry[5].clone()]
    torch.cuda.synchronize()
    for a, b in zip(imgs[1], imgs[0]):
        assert torch.equal(a.view(torch.int64) if a.dtype == torch.float64 else a, b.view(torch.int64) if b.dtype == torch.float64 else b)
    # tables and images built for the default window; the lookup then plans another one
    ttm_opt('table_fused', 1)
    tm._pack_memo = None
    Xi = tm.inverse_map(Zin)
    assert relerr(Xi, Xo) < 1e-11
    ttm_opt('rt_window', 300)
    Xw = tm.inverse_map(Zin)                                # (same coefficient vector: its cached tables and images)
    tm.inverse_device(tm._cols(D, tm._N, zero=True), tm._N)
    assert lib.ttm_last_kernel().decode() == 'k_band_inverse'
    assert relerr(Xw, Xo) < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize('shape,cls', [((5, 3, 2), 2), ((7, 6, 2), 3)])
def test_ring_inverse_of_the_higher_degree_classes(shape, cls, ttm_opt):
    """The ring kernel's refill variants (8 and 4 columns at a time) for the degree classes (5,5) and (7,7): against the block
    kernel bit for bit and the oracle."""
    D = 26
    tm, om, X, rng = _ring_map(D, n=4099, shape=shape, nm_scale=0.04)        # (tame offsets: 26 inversions in a row feed on each other)
    assert tm._cm.u_h_cls == cls and tm._cm.u_p_lag == 2
    lib = tm._lib
    lib.ttm_last_kernel.restype = ctypes.c_char_p
    Zin = rng.standard_normal((len(X), D))
    Zin[:30] *= 3.0
    Xo = om.inverse_map(Zin)
    ttm_opt('u_loader', 1); ttm_opt('band_inv', 1)
    res = {}
    for ring, block in ((0, -1), (1, -1), (1, 12), (1, 16)):
        ttm_opt('band_ring', ring); ttm_opt('rt_block', block)
        tm._pack_memo = None
        res[(ring, block)] = tm.inverse_map(Zin)
        tm.inverse_device(tm._cols(D, tm._N, zero=True), tm._N)
        assert lib.ttm_last_kernel().decode() == ('k_band_inverse_ring' if ring else 'k_band_inverse')
        # (the oracle sums the offsets in another order, which the degree-5 / degree-7 groups amplify along 26 inversions in a row whose
        # results reach |x| = 12: 2e-11 for BOTH kernels at class (5,5); test_band_kernels_… holds the 5-component maps to 1e-11)
        assert relerr(res[(ring, block)], Xo) < 1e-10, (ring, block)
    for key in ((1, -1), (1, 12), (1, 16)):
        assert np.array_equal(res[key], res[(0, -1)]), key


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['c5_shape', 'class_77', 'few_c3', 'few_entf'])
def test_one_launch_setup_equals_the_two_launches(case, ttm_opt):
    """ttm_setup_staged (k_setup: the uform workgroups and the table workgroups of a new coefficient vector side by side, the table
    workgroups folding privately into a scratch copy) against ttm_fold_staged + ttm_inverse_table_build_index: folded coefficients
    with their U section, tables, ranges, bucket indices, flags, resident-table images - bit for bit, over several vectors in a row
    (the scratch copy is reused)."""
    import torch
    tm, om, X, rng = _build(case)
    lib = tm._lib
    lib.ttm_last_kernel.restype = ctypes.c_char_p
    tm._inverse_seen = True                                   # (the default tables ride along with the fold from now on)
    tm.deferred_checks = True

    def setup(vec):
        for k in range(tm.D):
            tm.coeffs_mon[k] = vec[0][k].copy()
            tm.coeffs_nonmon[k] = vec[1][k].copy()
        tm._pack_memo = None
        coef = tm._pack_coeffs()
        name = lib.ttm_last_kernel().decode()
        assert tm.validate(coef)
        torch.cuda.synchronize()
        entry = next(iter(coef._ttm_tables.values()))
        return coef._ttm_fold.clone(), [None if (t is None or isinstance(t, bool)) else t.clone() for t in entry], name
    vecs = []
    for i in range(3):
        vecs.append(([0.2 + 0.5 * rng.random(len(c)) for c in tm.coeffs_mon],
                     [0.3 * rng.standard_normal(len(c)) / (1 + np.arange(len(c))) for c in tm.coeffs_nonmon]))
    one = [setup(v) for v in vecs]
    assert all(name == 'k_setup' for _, _, name in one), [n for _, _, n in one]
    ttm_opt('setup_fused', 0)
    two = [setup(v) for v in vecs]
    assert all(name == 'k_table_build_index' for _, _, name in two), [n for _, _, n in two]
    for (fa, ta, _), (fb, tb, _) in zip(one, two):
        assert fa.view(torch.int64).equal(fb.view(torch.int64))
        for a, b in zip(ta, tb):
            assert (a is None) == (b is None)
            if a is not None:
                assert torch.equal(a.view(torch.int64) if a.dtype == torch.float64 else a, b.view(torch.int64) if b.dtype == torch.float64 else b)
    # and the maps the vectors define
    Zin = rng.standard_normal((len(X), tm.D))
    E = CASES[case]['d'] - CASES[case]['D']
    ttm_opt('setup_fused', -1)
    tm._pack_memo = None
    Xi = tm.inverse_map(Zin, X_star=X[:, :E] if E else None)
    for k in range(tm.D):
        om.coeffs_mon[k], om.coeffs_nonmon[k] = tm.coeffs_mon[k].copy(), tm.coeffs_nonmon[k].copy()
    assert relerr(Xi, om.inverse_map(Zin, X_star=X[:, :E] if E else None)) < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize('case', [c for c in sorted(CASES) if CASES[c].get('few')])
def test_roundtrip_in_one_launch_equals_forward_then_inverse(case, ttm_opt):
    """ttm_roundtrip / roundtrip_device (k_band_few_roundtrip: the forward sweep, optionally the density terms, and the table inverse
    of the image as ONE pass over the ensemble, z in registers) against forward_device followed by inverse_device: the same
    statements in the same order, so the same bits - with and without the density terms, conditioning columns, one or several
    tiles per workgroup, an odd number of rows; and the two calls when the fused launch is switched off."""
    tm, om, X, rng = _build(case)
    lib = tm._lib
    lib.ttm_last_kernel.restype = ctypes.c_char_p
    D, d = CASES[case]['D'], CASES[case]['d']
    E = d - D
    N, Xs = tm._N, tm._Xs
    ttm_opt('u_loader', 1); ttm_opt('band_fwd', 1); ttm_opt('band_inv', 1)
    ttm_opt('roundtrip_fused', 1)             # (every shape through the fused kernel, also those the default leaves to the two calls)
    sigma = tm._to_dev(np.asarray(tm.X_std[E:E + D], dtype=float))

    def two_calls(dens):
        ld, ss = (tm._zeros(N), tm._zeros(N)) if dens else (None, None)
        Z = tm.forward_device(Xs, N, logdet=ld, sigma=sigma if dens else None, sumsq=ss)
        Xr = tm._cols(d, N, zero=True)
        if E:
            Xr[:E, :N].copy_(Xs[:E, :N])
        tm.inverse_device(Z, N, X=Xr)
        return Z, Xr, ld, ss, lib.ttm_last_kernel().decode()

    def one_call(dens):
        ld, ss = (tm._zeros(N), tm._zeros(N)) if dens else (None, None)
        Z, Xr = tm.roundtrip_device(Xs, N, logdet=ld, sigma=sigma if dens else None, sumsq=ss)
        return Z, Xr, ld, ss, lib.ttm_last_kernel().decode()

    def same(a, b):
        return np.array_equal(a[:, :N].cpu().numpy(), b[:, :N].cpu().numpy(), equal_nan=True)
    for cus in (-1, 1, 2):
        ttm_opt('band_cus', cus)
        for dens in (False, True):
            Z1, X1, l1, s1, inv_kernel = two_calls(dens)
            Z2, X2, l2, s2, rt_kernel = one_call(dens)
            if inv_kernel == 'k_band_few_inverse':
                assert rt_kernel == ('k_band_few_roundtrip<density>' if dens else 'k_band_few_roundtrip'), (case, cus, dens, rt_kernel)
            assert same(Z1, Z2) and same(X1, X2), (case, cus, dens)
            if dens:
                assert np.array_equal(l1.cpu().numpy(), l2.cpu().numpy(), equal_nan=True)
                assert np.array_equal(s1.cpu().numpy(), s2.cpu().numpy(), equal_nan=True)
    # the round trip is one: S^-1(S(x)) = x inside the tables
    xs, xr = Xs[E:, :N].cpu().numpy(), X2[E:, :N].cpu().numpy()
    inside = (np.abs(xs) < 5).all(axis=0)
    assert inside.mean() > 0.9 and np.median(np.abs(xr[:, inside] - xs[:, inside])) < 1e-3
    ttm_opt('band_cus', -1)
    ttm_opt('roundtrip_fused', 0)
    Z3, X3, _, _, k3 = one_call(False)
    assert not k3.startswith('k_band_few_roundtrip') and same(Z3, Z2) and same(X3, X2)
