"""
root_finder='newton' (SURVEY section 8a', K5 second mode; an extension, the default stays the reference's bisection):
the same roots as the reference's root search - residual |S - z| <= 1e-9 under the oracle's forward map, positions
within what that residual allows - in far fewer evaluations.  Both monotonicity modes, skipped dimensions, conditional
inversion.
"""
import numpy as np
import pytest

from tests.hostemu import emu
from tests.util import case_X, coeff_lists, ctor_kwargs, load_case, make_oracle


@pytest.fixture(params=[pytest.param('hostemu'), pytest.param('hip', marks=pytest.mark.gpu)])
def backend(request):
    if request.param == 'hostemu':
        with emu.install():
            yield 'hostemu'
    else:
        yield 'hip'


@pytest.mark.parametrize('name', ['c1_int', 'c2a_int', 'c3_int', 'misc_grid', 'c3_sep', 'misc_sep'])
def test_newton_finds_the_reference_roots(backend, name):
    from triangular_transport_toolbox_amd.transport_map import transport_map
    npz, desc = load_case(name)
    X = case_X(name, npz)
    kw = ctor_kwargs(desc)
    tm = transport_map(X=X, monotone=desc['monotone'], nonmonotone=desc['nonmonotone'], verbose=False,
                       root_finder='newton', alternate_root_finding=False, **kw)
    tm.coeffs_mon, tm.coeffs_nonmon = coeff_lists(npz, tm.D)
    om = make_oracle(name, npz, desc)
    Zin = npz['inv_Z']
    key = 'inv_X_nostar' if name == 'misc_grid' else ('inv_X' if 'inv_X' in npz else 'inv_X_bisect')
    ref = npz[key]
    got = tm.inverse_map(Zin)
    assert got.shape == ref.shape
    # the reference's own result may have run away (targets beyond a component's range: |x| ~ 1e8); compare where it did not
    sane = np.all(np.abs(ref) < 50.0, axis=1)
    sane[0] = False                                                     # (sample 0 of the reference: loop-guard quirk)
    assert sane.mean() > 0.8
    skipcols = np.zeros((len(got), om.skip_dimensions)) + om.X_mean[:om.skip_dimensions]
    res = np.abs(om.map(np.column_stack((skipcols, got))) - Zin)
    assert res[sane].max() < 2e-9                                       # the stopping rule, under the oracle's forward map
    # positions: both stop at |S - z| <= 1e-9, so they differ by <= 2e-9 / (dS/dx): 1e-6 covers slopes down to 2e-3
    close = np.abs(got[sane] - ref[sane]) <= 1e-6 * (1 + np.abs(ref[sane]))
    assert close.mean() > 0.98
    # conditional inversion (X_star) where the fixture has it
    if name == 'misc_grid':
        gotc = tm.inverse_map(Zin, X_star=npz['inv_Xstar'])
        refc = npz['inv_X']
        sane = np.all(np.abs(refc) < 50.0, axis=1)
        sane[0] = False
        assert (np.abs(gotc[sane] - refc[sane]) <= 1e-6 * (1 + np.abs(refc[sane]))).mean() > 0.98


def test_newton_needs_fewer_trial_points(backend):
    from triangular_transport_toolbox_amd.transport_map import transport_map
    npz, desc = load_case('c2a_int')
    X = case_X('c2a_int', npz)
    counts = {}
    for rf in ('reference', 'newton'):
        tm = transport_map(X=X, monotone=desc['monotone'], nonmonotone=desc['nonmonotone'], verbose=False, root_finder=rf,
                           **ctor_kwargs(desc))
        tm.coeffs_mon, tm.coeffs_nonmon = coeff_lists(npz, tm.D)
        coef = tm._pack_coeffs()
        Zs = tm._import(npz['inv_Z'], False)
        N = len(npz['inv_Z'])
        Xs = tm._cols(tm._cm.d_cols, N, zero=True)
        import ctypes
        iters = tm._zeros(tm.D, dtype=__import__('torch').int32)
        if rf == 'newton':
            rc = tm._lib.ttm_inverse_newton(tm._pp, tm._ptr(coef), tm._ptr(coef._ttm_fold), 0, tm.D, tm._ptr(Zs), Zs.shape[1],
                                            tm._ptr(Xs), Xs.shape[1], N, ctypes.c_void_p(iters.data_ptr()), tm._stream())
        else:
            rc = tm._lib.ttm_inverse_bisect(tm._pp, tm._ptr(coef), tm._ptr(coef._ttm_fold), 0, tm.D, tm._ptr(Zs), Zs.shape[1],
                                            tm._ptr(Xs), Xs.shape[1], N, ctypes.c_void_p(iters.data_ptr()), None, tm._stream())
        assert rc == 0
        counts[rf] = iters.cpu().numpy().copy()
    assert counts['reference'].max() >= 25 and counts['newton'].max() <= 25 and counts['newton'].sum() < 0.7 * counts['reference'].sum()


def test_root_finder_argument_is_checked(backend):
    from triangular_transport_toolbox_amd.transport_map import transport_map
    X = np.random.default_rng(0).standard_normal((64, 1))
    with pytest.raises(ValueError, match='root_finder'):
        transport_map(X=X, monotone=[[[0]]], nonmonotone=[[[]]], verbose=False, root_finder='secant')
