"""
Pins the CPU oracle (oracle/ttm_oracle.py) against outputs of the reference
itself (tests/golden/*.npz, produced by tests/golden/make_golden.py which
imports /root/reference/transport_map.py in the development container) and
against the reference's single shipped known-answer (Example 01 coefficients).

No GPU needed.
"""
import numpy as np
import pytest
import scipy.stats

from tests.util import (ALL_CASES, INTEGRATED, SEPARABLE, case_X, coeff_lists, ctor_kwargs, load_case, make_oracle, relerr)


@pytest.mark.parametrize('name', ALL_CASES)
def test_standardisation_and_special_terms(name):
    npz, desc = load_case(name)
    om = make_oracle(name, npz, desc)
    assert np.array_equal(om.X_mean, npz['X_mean'])       # same NumPy calls -> bit-exact
    assert np.array_equal(om.X_std, npz['X_std'])
    for kc, d in desc['special_terms'].items():
        for var, v in d.items():
            if var == 'cross-terms':
                for var2, v2 in v.items():
                    got = om.special_terms[int(kc)]['cross-terms'][int(var2)]
                    assert list(got['centers']) == v2['centers'] and list(got['scales']) == v2['scales']
            else:
                got = om.special_terms[int(kc)][int(var)]
                assert list(got['centers']) == v['centers'] and list(got['scales']) == v['scales']
    assert [len(c) for c in om.coeffs_mon] == desc['n_coeffs_mon']
    assert [len(c) for c in om.coeffs_nonmon] == desc['n_coeffs_nonmon']


@pytest.mark.parametrize('name', ALL_CASES)
def test_basis_matrices(name):
    npz, desc = load_case(name)
    om = make_oracle(name, npz, desc)
    Xs = om.X[:256]
    for k in range(om.D):
        assert relerr(om.fun_mon(k, Xs), npz['Psi_mon_%d' % k]) < 1e-14
        pn = om.fun_nonmon(k, Xs)
        if pn is None:
            assert 'Psi_nonmon_%d' % k not in npz
        else:
            assert relerr(pn, npz['Psi_nonmon_%d' % k]) < 1e-14
        if 'dPsi_mon_%d' % k in npz:
            assert relerr(om.der_fun_mon(k, Xs), npz['dPsi_mon_%d' % k]) < 1e-14


@pytest.mark.parametrize('name', ALL_CASES)
def test_forward_map(name):
    npz, desc = load_case(name)
    om = make_oracle(name, npz, desc)
    X = case_X(name, npz)[:npz['Z'].shape[0]]
    assert relerr(om.map(X), npz['Z']) < 1e-13


@pytest.mark.parametrize('name', INTEGRATED)
def test_objective_integrated(name):
    npz, desc = load_case(name)
    om = make_oracle(name, npz, desc)
    for k in range(om.D):
        div = len(om.coeffs_nonmon[k])
        for c, J, G in zip(npz['obj_c_%d' % k], npz['obj_J_%d' % k], npz['obj_G_%d' % k]):
            assert abs(om.objective_function(c.copy(), k, div) - J) <= 1e-13 * (1 + abs(J))
            assert relerr(om.objective_function_jacobian(c.copy(), k, div), G) < 1e-12


@pytest.mark.parametrize('name', ['c2b_sep', 'c3_sep', 'c5_sep'])
def test_objective_separable(name):
    npz, desc = load_case(name)
    om = make_oracle(name, npz, desc)
    for k in range(0, om.D, 7 if om.D > 8 else 1):
        A, aux = om.separable_setup(k)
        assert relerr(A, npz['sep_A_%d' % k]) < 1e-12
        for c, J, G in zip(npz['sep_c_%d' % k], npz['sep_J_%d' % k], npz['sep_G_%d' % k]):
            Jo, Go = om.separable_objective(c.copy(), npz['sep_A_%d' % k], k)
            assert abs(Jo - J) <= 1e-13 * (1 + abs(J))
            assert relerr(Go, G) < 1e-12
        cn = om.separable_nonmonotone(npz['coeffs_mon_%d' % k], aux)
        assert relerr(cn, npz['coeffs_nonmon_%d' % k]) < 1e-9


def test_example01_known_answer():
    """The reference's only shipped fixture: order-10 spiral coefficients."""
    npz, desc = load_case('ex01_order10')
    om = make_oracle('ex01_order10', npz, desc)
    J0 = om.objective_function(None, 0, len(om.coeffs_nonmon[0]))
    J1 = om.objective_function(None, 1, len(om.coeffs_nonmon[1]))
    assert abs(J0 - 0.22517858233600704) < 1e-13
    assert abs(J1 - (-0.7978830242276339)) < 1e-13
    assert np.allclose([J0, J1], npz['J'], rtol=0, atol=1e-13)
    for k in range(2):
        G = om.objective_function_jacobian(None, k, len(om.coeffs_nonmon[k]))
        assert np.max(np.abs(G)) < 1e-5                 # BFGS gtol of the shipped optimum
        assert relerr(G, npz['G_%d' % k]) < 1e-12
    Z = om.map(npz['X_head'])
    assert relerr(Z, npz['Z_head']) < 1e-13
    assert relerr(om.inverse_map(npz['inv_Z']), npz['inv_X']) < 1e-9


@pytest.mark.parametrize('name', ['c1_int', 'c2a_int', 'c3_int', 'c5_int'])
def test_inverse_bisection_integrated(name):
    npz, desc = load_case(name)
    om = make_oracle(name, npz, desc)
    X = om.inverse_map(npz['inv_Z'])
    assert relerr(X, npz['inv_X']) < 1e-8
    if 'inv_X_n1' in npz:   # quirk: a single sample is never refined (TM:3952)
        assert relerr(om.inverse_map(npz['inv_Z'][:1]), npz['inv_X_n1']) < 1e-12
        assert relerr(om.inverse_map(npz['inv_Z'][:2]), npz['inv_X_n2']) < 1e-8
    if 'inv_cond_X' in npz:
        Xc = om.inverse_map(npz['inv_Z'][:, 1:], X_star=npz['inv_cond_Xstar'])
        assert relerr(Xc, npz['inv_cond_X']) < 1e-8


@pytest.mark.parametrize('name', SEPARABLE)
def test_inverse_separable(name):
    npz, desc = load_case(name)
    om = make_oracle(name, npz, desc)
    assert relerr(om.inverse_map(npz['inv_Z']), npz['inv_X_table']) < 1e-11
    om.alternate_root_finding = False
    assert relerr(om.inverse_map(npz['inv_Z']), npz['inv_X_bisect']) < 1e-8
    if 'inv_X_bisect_n1' in npz:
        assert relerr(om.inverse_map(npz['inv_Z'][:1]), npz['inv_X_bisect_n1']) < 1e-12
    om.alternate_root_finding = True
    if 'inv_cond_X' in npz:
        Xc = om.inverse_map(npz['inv_Z'][:, 1:], X_star=npz['inv_cond_Xstar'])
        assert relerr(Xc, npz['inv_cond_X']) < 1e-11
    for k in range(om.D):
        if 'table_out_%d' % k in npz:
            pts = np.linspace(-10, 10, 1001)
            fakeX = np.zeros((1001, om.X.shape[1]))
            fakeX[:, om.skip_dimensions + k] = pts
            out = np.dot(om.fun_mon(k, fakeX), om.coeffs_mon[k])
            assert relerr(out, npz['table_out_%d' % k]) < 1e-14


def test_misc_grid_conditional_inverse():
    npz, desc = load_case('misc_grid')
    om = make_oracle('misc_grid', npz, desc)
    assert relerr(om.inverse_map(npz['inv_Z'], X_star=npz['inv_Xstar']), npz['inv_X']) < 1e-8
    assert relerr(om.inverse_map(npz['inv_Z']), npz['inv_X_nostar']) < 1e-8


@pytest.mark.parametrize('name', SEPARABLE)
def test_densities(name):
    npz, desc = load_case(name)
    om = make_oracle(name, npz, desc)
    X = case_X(name, npz)[:npz['pullback'].shape[0]]
    assert relerr(om.evaluate_pullback_density(X), npz['pullback']) < 1e-12
    if 'pushforward' in npz:
        def log_target_pdf(x):
            return scipy.stats.multivariate_normal.logpdf(x, mean=np.zeros(x.shape[-1]), cov=np.identity(x.shape[-1]))
        got = om.evaluate_pushforward_density(npz['inv_Z'], log_target_pdf)
        ok = np.isfinite(npz['pushforward'])
        assert np.array_equal(np.isfinite(got), ok)
        assert relerr(got[ok], npz['pushforward'][ok]) < 1e-9


def test_interp1d_restatement_matches_scipy():
    from scipy.interpolate import interp1d
    from oracle.ttm_oracle import interp1d_linear
    rng = np.random.default_rng(0)
    out = np.cumsum(np.abs(rng.standard_normal(1001)))
    pts = np.linspace(-10, 10, 1001)
    t = rng.uniform(out[0] - 1, out[-1] + 1, 500)
    assert np.array_equal(interp1d(out, pts, fill_value='extrapolate')(t), interp1d_linear(out, pts, t))


def test_optimize_reaches_reference_objective():
    """Oracle optimize() (same SciPy calls) reproduces the reference coefficients."""
    for name in ['c1_int', 'c2b_sep']:
        npz, desc = load_case(name)
        om = make_oracle(name, npz, desc)
        ref_mon, ref_non = coeff_lists(npz, om.D)
        om.coeffs_mon = [c * 0 for c in ref_mon]
        om.coeffs_nonmon = [c * 0 for c in ref_non]
        om.optimize()
        for k in range(om.D):
            assert relerr(om.coeffs_mon[k], ref_mon[k]) < 1e-6
            assert relerr(om.coeffs_nonmon[k], ref_non[k]) < 1e-6


def test_entf_first_update_and_cycles():
    """Example 06 filter map (skip_dimensions = 1, L2): first update in detail."""
    npz, desc = load_case('entf')
    from oracle.ttm_oracle import OracleMap
    om = OracleMap(X=npz['u0_map_input'], monotone=desc['monotone'], nonmonotone=desc['nonmonotone'],
                   **{**desc['kwargs'], 'quadrature_input': {'order': 5}})
    assert np.array_equal(om.X_mean, npz['u0_X_mean'])
    for k in range(om.D):
        A, aux = om.separable_setup(k)
        assert relerr(A, npz['u0_sep_A_%d' % k]) < 1e-11
    om.optimize()
    for k in range(om.D):
        assert relerr(om.coeffs_mon[k], npz['u0_coeffs_mon_%d' % k]) < 1e-6
        assert relerr(om.coeffs_nonmon[k], npz['u0_coeffs_nonmon_%d' % k]) < 1e-6
    om.coeffs_mon, om.coeffs_nonmon = coeff_lists(npz, om.D, prefix='u0_')
    Zp = om.map(npz['u0_map_input'])
    assert relerr(Zp, npz['u0_Z']) < 1e-12
    Ystar = np.repeat(npz['obs'][0][0].reshape((1, 1)), Zp.shape[0], axis=0)
    assert relerr(om.inverse_map(npz['u0_Z'], X_star=Ystar), npz['u0_ret']) < 1e-11


def test_entf_five_cycles():
    """The reference's five assimilation cycles (every random draw replayed) through the oracle and the host-loop
    harness of the product: the analysis ensemble of every cycle within 1e-6."""
    from oracle.ttm_oracle import OracleMap
    from triangular_transport_toolbox_amd import entf
    npz, desc = load_case('entf')
    ens = npz['ens0']
    om = OracleMap(X=np.random.default_rng(0).uniform(size=(ens.shape[0], 4)), monotone=desc['monotone'],
                   nonmonotone=desc['nonmonotone'], **{**desc['kwargs'], 'quadrature_input': {'order': 5}})
    assert len(npz['obs']) == 5
    for t in range(5):
        noises = [npz['noise_%d_%d' % (t, i)] for i in range(3)]
        Xa = entf.assimilate(om, ens, npz['obs'][t], noises)
        assert relerr(Xa, npz['ens_%d_2' % t]) < 1e-6
        ens = entf.rk4(Xa, 0.05, 2)
        assert relerr(ens, npz['forecast_%d' % t]) < 1e-6


def test_example05_density_grids():
    """example_05.py:146-162 and 330-400 through the oracle."""
    from oracle.ttm_oracle import OracleMap
    from triangular_transport_toolbox_amd import specs
    npz, desc = load_case('ex05_density')
    om = make_oracle('ex05_density', npz, desc['full'])
    grid = npz['grid']
    assert relerr(om.map(grid), npz['grid_Z']) < 1e-12
    assert relerr(om.evaluate_pullback_density(grid), npz['pullback']) < 1e-12
    got = om.evaluate_pushforward_density(grid, specs.logpdf_wavy)
    ok = np.isfinite(npz['pushforward'])
    assert np.array_equal(np.isfinite(got), ok) and relerr(got[ok], npz['pushforward'][ok]) < 1e-9
    d = desc['conditional']
    oc = OracleMap(X=npz['X'], monotone=d['monotone'], nonmonotone=d['nonmonotone'], **ctor_kwargs(d))
    oc.coeffs_mon, oc.coeffs_nonmon = coeff_lists(npz, oc.D, prefix='cond_')
    g = np.linspace(-3, 3, 101)[:, np.newaxis]
    Xstar = np.ones((101, 1))
    assert relerr(oc.evaluate_pullback_density(g, X_star=Xstar), npz['cond_pullback']) < 1e-12
    got = oc.evaluate_pushforward_density(g, lambda x: specs.logpdf_wavy(np.column_stack((Xstar, x))), X_star=Xstar)
    ok = np.isfinite(npz['cond_pushforward'])
    assert np.array_equal(np.isfinite(got), ok) and relerr(got[ok], npz['cond_pushforward'][ok]) < 1e-9
