"""
'LIN' tail-linearisation modifier and linearisation thresholds (reference TM:1042-1057,
1375-1385, 1513-1541, 2364-2389), integrated rectifier.

Fixtures misc_lin_q / misc_lin_abs come from the reference (tests/golden/make_golden.py lin): thresholds as
quantiles (0.05 / 0.95) and prescribed (+-1.5, increment 1e-5); maps with 'LIN' on plain, Hermite-function and
cross-term factors in both lists; samples far outside the thresholds.

What the modifier does in the reference: the generated code blends P(x_clipped) and P(x_clipped + increment), but
the text substitution that should insert the clipped / extended variables finds nothing to replace ("__x__" was
already replaced by "x", TM:1371 before TM:1381-1385), so both sides evaluate P at the unclipped x:
    P_LIN = P(x) (1 - v/inc) + P(x) v/inc,   v = overshoot beyond the thresholds,
i.e. P(x) with rounding noise of relative size ~ |v|/inc * 1e-16 (1e-10 ... 1e-8 on the fixtures).  The oracle
restates that arithmetic operation by operation and matches the reference to the last bit; the engine evaluates the
factor once.  LIN_TOL below is therefore the reference's own noise floor, not an engine tolerance: the same fixtures
with the modifier removed from the oracle agree with the engine to the usual 1e-11.
"""
import numpy as np
import pytest

from tests.hostemu import emu
from tests.util import coeff_lists, ctor_kwargs, load_case, make_oracle, relerr

LIN_CASES = ['misc_lin_q', 'misc_lin_abs']
LIN_TOL = 2e-8          # |v| / increment * eps with |v| <~ 20 standard deviations on X_far (see module docstring)


@pytest.fixture(params=[pytest.param('hostemu'), pytest.param('hip', marks=pytest.mark.gpu)])
def backend(request):
    if request.param == 'hostemu':
        with emu.install():
            yield 'hostemu'
    else:
        yield 'hip'


def make_tm(npz, desc, **extra):
    from triangular_transport_toolbox_amd.transport_map import transport_map
    kw = ctor_kwargs(desc)
    kw.update(extra)
    tm = transport_map(X=npz['X'], monotone=desc['monotone'], nonmonotone=desc['nonmonotone'], verbose=False, **kw)
    tm.coeffs_mon, tm.coeffs_nonmon = coeff_lists(npz, tm.D)
    return tm


def strip_lin(om):
    """The oracle's plans with the modifier removed (what the blend equals in exact arithmetic)."""
    om.plan_mon = [[[(f[:4] + (False,)) if f[0] == 'poly' else f for f in t] for t in terms] for terms in om.plan_mon]
    om.plan_nonmon = [[[(f[:4] + (False,)) if f[0] == 'poly' else f for f in t] for t in terms] for terms in om.plan_nonmon]
    return om


# ---- oracle against the reference (CPU) --------------------------------------------------------------------------

@pytest.mark.parametrize('name', LIN_CASES)
def test_oracle_matches_reference_bitwise(name):
    npz, desc = load_case(name)
    om = make_oracle(name, npz, desc)
    assert np.array_equal(om.linearization_threshold, npz['linearization_threshold'])
    assert relerr(om.map(npz['X']), npz['Z']) < 1e-15
    assert relerr(om.map(npz['X_far']), npz['Z_far']) < 1e-15
    for k in range(om.D):
        assert relerr(om.fun_mon(k, om.X[:256]), npz['Psi_mon_%d' % k]) < 1e-15
        if ('Psi_nonmon_%d' % k) in npz:
            assert relerr(om.fun_nonmon(k, om.X[:256]), npz['Psi_nonmon_%d' % k]) < 1e-15
        div = len(om.coeffs_nonmon[k])
        for c, J, G in zip(npz['obj_c_%d' % k], npz['obj_J_%d' % k], npz['obj_G_%d' % k]):
            assert abs(om.objective_function(c.copy(), k, div) - J) <= 1e-13 * (1 + abs(J))
            assert relerr(om.objective_function_jacobian(c.copy(), k, div), G) < 1e-12
    om.reset(npz['X_reset'])
    assert np.array_equal(om.linearization_threshold, npz['linearization_threshold_reset'])


@pytest.mark.parametrize('name', LIN_CASES)
def test_blend_is_the_identity_up_to_its_own_rounding_noise(name):
    npz, desc = load_case(name)
    om = strip_lin(make_oracle(name, npz, desc))
    dz, dfar = relerr(om.map(npz['X']), npz['Z']), relerr(om.map(npz['X_far']), npz['Z_far'])
    assert 1e-13 < dfar < LIN_TOL and dz < LIN_TOL          # noise is there, and it is only noise


def test_lin_needs_a_linearization_value():
    from oracle.ttm_oracle import OracleMap
    X = np.random.default_rng(0).standard_normal((50, 1))
    with pytest.raises(Exception, match="'LIN' modifier specified"):
        OracleMap(X=X, monotone=[[[0, 'LIN']]], nonmonotone=[[[]]])


# ---- engine against the reference ----------------------------------------------------------------------------------

@pytest.mark.parametrize('name', LIN_CASES)
def test_thresholds(backend, name):
    # quantiles of the STANDARDISED columns: the order statistics and their interpolation are exact
    # (tests/test_quantile.py), the column mean / std come from a device reduction whose summation order differs
    # from NumPy's (1e-13, test_standardisation_and_special_terms) - same tolerance as the special-term centres
    npz, desc = load_case(name)
    tm = make_tm(npz, desc)
    assert relerr(tm.linearization_threshold, npz['linearization_threshold']) < 1e-12
    tm.reset(npz['X_reset'])
    assert relerr(tm.linearization_threshold, npz['linearization_threshold_reset']) < 1e-12
    if not desc['kwargs'].get('linearization_specified_as_quantiles', True):
        assert np.array_equal(tm.linearization_threshold, npz['linearization_threshold'])


@pytest.mark.parametrize('name', LIN_CASES)
def test_map_basis_objective_inverse(backend, name):
    npz, desc = load_case(name)
    tm = make_tm(npz, desc)
    om = strip_lin(make_oracle(name, npz, desc))
    assert [len(c) for c in tm.coeffs_mon] == desc['n_coeffs_mon']
    assert [len(c) for c in tm.coeffs_nonmon] == desc['n_coeffs_nonmon']
    # against the reference: its noise floor; against the noise-free restatement: the usual tolerances
    assert relerr(tm.map(npz['X']), npz['Z']) < LIN_TOL
    assert relerr(tm.map(npz['X_far']), npz['Z_far']) < LIN_TOL
    assert relerr(tm.map(npz['X']), om.map(npz['X'])) < 1e-11
    assert relerr(tm.map(npz['X_far']), om.map(npz['X_far'])) < 1e-11
    Xs = om.X[:256]
    for k in range(tm.D):
        assert relerr(tm.basis(k, 'mon', Xs), npz['Psi_mon_%d' % k]) < LIN_TOL
        assert relerr(tm.basis(k, 'mon', Xs), om.fun_mon(k, Xs)) < 1e-12
        if ('Psi_nonmon_%d' % k) in npz:
            assert relerr(tm.basis(k, 'nonmon', Xs), npz['Psi_nonmon_%d' % k]) < LIN_TOL
        div = len(tm.coeffs_nonmon[k])
        for c, J, G in zip(npz['obj_c_%d' % k], npz['obj_J_%d' % k], npz['obj_G_%d' % k]):
            assert abs(tm.objective_function(c.copy(), k, div) - J) <= LIN_TOL * (1 + abs(J))
            assert relerr(tm.objective_function_jacobian(c.copy(), k, div), G) < LIN_TOL
            Jo = om.objective_function(c.copy(), k, div)
            assert abs(tm.objective_function(c.copy(), k, div) - Jo) <= 1e-10 * (1 + abs(Jo))     # north_star bar
            assert relerr(tm.objective_function_jacobian(c.copy(), k, div), om.objective_function_jacobian(c.copy(), k, div)) < 1e-10
    # bisection inverse: the stopping rule is a residual (|S - z| <= 1e-9, TM:3952) and the reference's residuals carry
    # the blend's noise, so where S is flat (dS/dx ~ 1e-6) the two midpoint sequences stop a few steps apart: compare
    # residuals under the noise-free forward map for every sample, positions only where the map is not flat
    # (samples whose target lies beyond the range of a component - the window doubling then runs away to |x| ~ 1e8
    # in the reference and here alike - are left out)
    Zin = npz['inv_Z']
    X = tm.inverse_map(Zin)
    sane = np.all(np.abs(npz['inv_X']) < 50.0, axis=1)
    sane[0] = False                                                       # (sample 0: loop-guard quirk, TM:3952)
    assert sane.mean() > 0.9
    assert np.max(np.abs(om.map(X)[sane] - Zin[sane])) < 5e-9
    assert np.max(np.abs(om.map(npz['inv_X'])[sane] - Zin[sane])) < 5e-8
    close = np.abs(X[sane] - npz['inv_X'][sane]) < 1e-6 * (1 + np.abs(X[sane]))
    assert close.mean() > 0.95


def test_engine_rejects_lin_without_value_and_separable_linearization(backend):
    from triangular_transport_toolbox_amd.transport_map import transport_map
    X = np.random.default_rng(0).standard_normal((64, 1))
    with pytest.raises(Exception, match="'LIN' modifier specified"):
        transport_map(X=X, monotone=[[[0, 'LIN']]], nonmonotone=[[[]]], verbose=False)
    with pytest.raises(NotImplementedError, match='TM:2063-2080'):
        transport_map(X=X, monotone=[[[0], 'iRBF 0']], nonmonotone=[[[]]], verbose=False,
                      monotonicity='separable monotonicity', linearization=0.05)
