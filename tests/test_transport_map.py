"""
Parity of the product class `transport_map` against the reference goldens.

Every test runs twice:
  * backend 'hip'     (marked gpu): the real path - libttm.so HIP kernels through the C ABI;
  * backend 'hostemu' (CPU)       : the same class driven through the host test double of
    the C ABI (tests/hostemu) - checks the host logic without a GPU.
Tolerances are those of SURVEY.md section 8c / BASELINE.md section 3.
"""
import numpy as np
import pytest
import scipy.stats

from tests.hostemu import emu
from tests.util import (ALL_CASES, INTEGRATED, SEPARABLE, case_X, check, coeff_lists, ctor_kwargs, load_case, make_oracle, relerr)


@pytest.fixture(params=[pytest.param('hostemu'), pytest.param('hip', marks=pytest.mark.gpu)])
def backend(request):
    if request.param == 'hostemu':
        with emu.install():
            yield 'hostemu'
    else:
        yield 'hip'


def make_tm(name, npz=None, desc=None, X=None, with_coeffs=True, **extra):
    from triangular_transport_toolbox_amd.transport_map import transport_map
    if npz is None:
        npz, desc = load_case(name)
    if X is None:
        X = case_X(name, npz)
    kw = ctor_kwargs(desc)
    kw.update(extra)
    tm = transport_map(X=X, monotone=desc['monotone'], nonmonotone=desc['nonmonotone'], verbose=False, **kw)
    if with_coeffs and 'coeffs_mon_0' in npz:
        tm.coeffs_mon, tm.coeffs_nonmon = coeff_lists(npz, tm.D)
    return tm


@pytest.mark.parametrize('name', ALL_CASES)
def test_standardisation_and_special_terms(backend, name):
    npz, desc = load_case(name)
    tm = make_tm(name, npz, desc)
    assert relerr(tm.X_mean, npz['X_mean']) < 1e-13
    assert relerr(tm.X_std, npz['X_std']) < 1e-13
    assert tm.D == desc['D'] and tm.skip_dimensions == desc['skip_dimensions']
    for kc, d in desc['special_terms'].items():
        for var, v in d.items():
            groups = v.items() if var == 'cross-terms' else [(var, v)]
            for var2, v2 in groups:
                got = tm.special_terms[int(kc)]['cross-terms'][int(var2)] if var == 'cross-terms' \
                    else tm.special_terms[int(kc)][int(var2)]
                assert relerr(got['centers'], v2['centers']) < 1e-12
                assert relerr(got['scales'], v2['scales']) < 1e-12
    assert [len(c) for c in tm.coeffs_mon] == desc['n_coeffs_mon']
    assert [len(c) for c in tm.coeffs_nonmon] == desc['n_coeffs_nonmon']
    om = make_oracle(name, npz, desc)
    assert relerr(tm.X[:64], om.X[:64]) < 1e-12


@pytest.mark.parametrize('name', ALL_CASES)
def test_map(backend, name):
    npz, desc = load_case(name)
    tm = make_tm(name, npz, desc)
    X = case_X(name, npz)[:npz['Z'].shape[0]]
    Z = tm.map(X)
    assert Z.shape == npz['Z'].shape
    check('golden/map[%s]' % name, relerr(Z, npz['Z']), 1e-11, backend)
    if npz['Z'].shape[0] == tm._N:
        assert relerr(tm.map(), npz['Z']) < 1e-11          # stored samples


@pytest.mark.parametrize('name', ['c1_int', 'c3_sep', 'misc_grid', 'misc_sep', 'misc_family_laguerre'])
def test_s_and_basis(backend, name):
    npz, desc = load_case(name)
    tm = make_tm(name, npz, desc)
    om = make_oracle(name, npz, desc)
    Xs = om.X[:256]
    for k in range(tm.D):
        assert relerr(tm.basis(k, 'mon', Xs), npz['Psi_mon_%d' % k]) < 1e-12
        pn = tm.basis(k, 'nonmon', Xs)
        if pn is not None:
            assert relerr(pn, npz['Psi_nonmon_%d' % k]) < 1e-12
        if 'dPsi_mon_%d' % k in npz:
            assert relerr(tm.basis(k, 'der_mon', Xs), npz['dPsi_mon_%d' % k]) < 1e-12
        rng = np.random.default_rng(k)
        cn, cm = rng.standard_normal(len(om.coeffs_nonmon[k])) * 0.1, np.abs(rng.standard_normal(len(om.coeffs_mon[k]))) * 0.1
        assert relerr(tm.s(Xs, k, cn, cm), om.s(Xs, k, cn, cm)) < 1e-11
        assert relerr(tm.s(None, k)[:256], om.s(Xs, k)) < 1e-11


@pytest.mark.parametrize('name', INTEGRATED)
def test_objective_integrated(backend, name):
    npz, desc = load_case(name)
    tm = make_tm(name, npz, desc)
    for k in range(tm.D):
        div = len(tm.coeffs_nonmon[k])
        for c, J, G in zip(npz['obj_c_%d' % k], npz['obj_J_%d' % k], npz['obj_G_%d' % k]):
            check('golden/objective_J[%s]' % name, abs(tm.objective_function(c.copy(), k, div) - J) / (1 + abs(J)), 1e-10, backend)   # north_star bar
            check('golden/objective_gradJ[%s]' % name, relerr(tm.objective_function_jacobian(c.copy(), k, div), G), 1e-10, backend)


def test_example01_known_answer(backend):
    """The reference's shipped order-10 spiral coefficients on seed-0 data."""
    npz, desc = load_case('ex01_order10')
    tm = make_tm('ex01_order10', npz, desc)
    J0 = tm.objective_function(None, 0, len(tm.coeffs_nonmon[0]))
    J1 = tm.objective_function(None, 1, len(tm.coeffs_nonmon[1]))
    check('golden/ex01_known_answer_J0', abs(J0 - 0.22517858233600704) / 1.3, 1e-10, backend)
    check('golden/ex01_known_answer_J1', abs(J1 - (-0.7978830242276339)) / 1.8, 1e-10, backend)
    for k in range(2):
        G = tm.objective_function_jacobian(None, k, len(tm.coeffs_nonmon[k]))
        assert np.max(np.abs(G)) < 1e-5
        assert relerr(G, npz['G_%d' % k]) < 1e-10
    assert relerr(tm.map(npz['X_head']), npz['Z_head']) < 1e-11
    X = tm.inverse_map(npz['inv_Z'])
    assert relerr(X, npz['inv_X']) < 1e-6


@pytest.mark.parametrize('name', ['c2b_sep', 'c3_sep', 'c5_sep'])
def test_separable_reduction(backend, name):
    npz, desc = load_case(name)
    tm = make_tm(name, npz, desc)
    for k in range(0, tm.D, 13 if tm.D > 8 else 1):
        A, solve_nonmon = tm.separable_setup(k)                # the engine's own Gram + Cholesky reduction ...
        check('golden/sep_A[%s]' % name, relerr(A, npz['sep_A_%d' % k]), 1e-10, backend)
        for c, J, G in zip(npz['sep_c_%d' % k], npz['sep_J_%d' % k], npz['sep_G_%d' % k]):
            Jg, Gg = tm.separable_objective(c.copy(), A, k)    # ... is what the objective is evaluated with
            check('golden/sep_J[%s]' % name, abs(Jg - J) / (1 + abs(J)), 1e-10, backend)
            check('golden/sep_gradJ[%s]' % name, relerr(Gg, G), 1e-10, backend)
        check('golden/sep_c_nonmon[%s]' % name, relerr(solve_nonmon(npz['coeffs_mon_%d' % k]), npz['coeffs_nonmon_%d' % k]), 1e-10, backend)


@pytest.mark.parametrize('name', SEPARABLE)
def test_inverse_table(backend, name):
    npz, desc = load_case(name)
    tm = make_tm(name, npz, desc)
    X = tm.inverse_map(npz['inv_Z'])
    assert X.shape == npz['inv_X_table'].shape
    check('golden/inverse_table[%s]' % name, relerr(X, npz['inv_X_table']), 1e-11, backend)
    if 'inv_cond_X' in npz:
        Xc = tm.inverse_map(npz['inv_Z'][:, 1:], X_star=npz['inv_cond_Xstar'])
        check('golden/inverse_table_conditional[%s]' % name, relerr(Xc, npz['inv_cond_X']), 1e-11, backend)


@pytest.mark.parametrize('name', ['c1_int', 'c2a_int', 'c3_int', 'c2b_sep', 'c3_sep', 'misc_sep', 'misc_grid'])
def test_inverse_bisection(backend, name):
    npz, desc = load_case(name)
    tm = make_tm(name, npz, desc, alternate_root_finding=False)
    om = make_oracle(name, npz, desc)
    Zin = npz['inv_Z']
    key = 'inv_X_nostar' if name == 'misc_grid' else ('inv_X' if 'inv_X' in npz else 'inv_X_bisect')
    X = tm.inverse_map(Zin)
    check('golden/inverse_bisection[%s]' % name, relerr(X, npz[key]), 1e-6, backend)
    # residual under the oracle's forward map (bisection stops at |S - z| <= 1e-9)
    skipcols = np.zeros((len(X), om.skip_dimensions)) + om.X_mean[:om.skip_dimensions]
    assert np.max(np.abs(om.map(np.column_stack((skipcols, X)))[1:] - Zin[1:])) < 5e-9
    n1 = npz.get('inv_X_n1', npz.get('inv_X_bisect_n1'))
    if n1 is not None:      # quirk 1: a single sample is never refined (TM:3952)
        assert relerr(tm.inverse_map(Zin[:1]), n1) < 1e-12
    if 'inv_X_n2' in npz:
        assert relerr(tm.inverse_map(Zin[:2]), npz['inv_X_n2']) < 1e-6
    if name == 'misc_grid':
        assert relerr(tm.inverse_map(Zin, X_star=npz['inv_Xstar']), npz['inv_X']) < 1e-6
    elif 'inv_cond_X' in npz and name.endswith('_int'):     # (the separable fixtures hold table-mode values)
        assert relerr(tm.inverse_map(Zin[:, 1:], X_star=npz['inv_cond_Xstar']), npz['inv_cond_X']) < 1e-6


@pytest.mark.parametrize('name', SEPARABLE)
def test_densities(backend, name):
    npz, desc = load_case(name)
    tm = make_tm(name, npz, desc)
    X = case_X(name, npz)[:npz['pullback'].shape[0]]
    got = tm.evaluate_pullback_density(X)
    check('golden/pullback[%s]' % name, relerr(got, npz['pullback']), 1e-10, backend)
    pos = npz['pullback'] > 0                  # (d = 40: densities of 1e-30 and less - the logarithm is what carries digits)
    assert np.all(got[pos] > 0)
    check('golden/log_pullback[%s]' % name, relerr(np.log(got[pos]), np.log(npz['pullback'][pos])), 1e-10, backend)
    if 'pushforward' in npz:
        def log_target_pdf(x):
            return scipy.stats.multivariate_normal.logpdf(x, mean=np.zeros(x.shape[-1]), cov=np.identity(x.shape[-1]))
        got = tm.evaluate_pushforward_density(npz['inv_Z'], log_target_pdf)
        ok = np.isfinite(npz['pushforward'])
        assert np.array_equal(np.isfinite(got), ok)
        check('golden/pushforward[%s]' % name, relerr(got[ok], npz['pushforward'][ok]), 1e-8, backend)


def test_example05_density_grids(backend):
    """The pullback and pushforward densities of example_05.py on its 101 x 101 grid (example_05.py:146-162; the grid
    reaches three standard deviations beyond the samples: the tails of the table inverse) and the conditional
    densities at x_1 = 1 of its second half."""
    from triangular_transport_toolbox_amd import specs
    npz, desc = load_case('ex05_density')
    tm = make_tm('ex05_density', npz, desc['full'])
    grid = npz['grid']
    check('golden/ex05_map', relerr(tm.map(grid), npz['grid_Z']), 1e-11, backend)
    check('golden/ex05_pullback', relerr(tm.evaluate_pullback_density(grid), npz['pullback']), 1e-10, backend)
    got = tm.evaluate_pushforward_density(grid, specs.logpdf_wavy)
    ok = np.isfinite(npz['pushforward'])
    assert np.array_equal(np.isfinite(got), ok)
    assert relerr(got[ok], npz['pushforward'][ok]) < 1e-8
    tc = make_tm('ex05_density', npz, desc['conditional'], with_coeffs=False)
    tc.coeffs_mon, tc.coeffs_nonmon = coeff_lists(npz, tc.D, prefix='cond_')
    g = np.linspace(-3, 3, 101)[:, np.newaxis]
    Xstar = np.ones((101, 1))
    assert relerr(tc.evaluate_pullback_density(g, X_star=Xstar), npz['cond_pullback']) < 1e-10
    got = tc.evaluate_pushforward_density(g, lambda x: specs.logpdf_wavy(np.column_stack((Xstar, x))), X_star=Xstar)
    ok = np.isfinite(npz['cond_pushforward'])
    assert np.array_equal(np.isfinite(got), ok)
    assert relerr(got[ok], npz['cond_pushforward'][ok]) < 1e-8


@pytest.mark.parametrize('name', ['c1_int', 'c2b_sep', 'c3_sep', 'c5_sep'])
def test_optimize(backend, name):
    """optimize() against the reference's final coefficients.  c5_sep: the 40 components of BASELINE
    config 5 on the fixture's 10^4 samples - what bench.py times at N = 10^6 (TM:2903-3172); there the objective the
    reference's own L-BFGS-B run ended on is in the fixture (sep_Jopt_k) and is compared as well."""
    npz, desc = load_case(name)
    tm = make_tm(name, npz, desc, with_coeffs=False)
    ref_mon, ref_non = coeff_lists(npz, tm.D)
    tm.optimize()
    om = make_oracle(name, npz, desc)
    for k in range(tm.D):
        if 'sep_Jopt_%d' % k in npz:
            A, _ = om.separable_setup(k)
            J_got = om.separable_objective(tm.coeffs_mon[k], A, k)[0]
            J_opt = float(npz['sep_Jopt_%d' % k])
            assert J_got <= J_opt + 1e-8 * abs(J_opt) + 1e-10
            check('golden/optimize_J_vs_reference_Jopt[%s]' % name, max(J_got - J_opt, 0.0) / (1 + abs(J_opt)), 1e-8, backend)
        if name.endswith('_int'):
            div = len(ref_non[k])
            J_ref = om.objective_function(np.concatenate((ref_non[k], ref_mon[k])), k, div)
            J_got = om.objective_function(np.concatenate((tm.coeffs_nonmon[k], tm.coeffs_mon[k])), k, div)
        else:
            A, _ = om.separable_setup(k)
            J_ref = om.separable_objective(ref_mon[k], A, k)[0]
            J_got = om.separable_objective(tm.coeffs_mon[k], A, k)[0]
        assert J_got <= J_ref + 1e-8 * abs(J_ref) + 1e-10
        assert relerr(tm.coeffs_mon[k], ref_mon[k]) < 2e-4
        assert relerr(tm.coeffs_nonmon[k], ref_non[k]) < 2e-4


@pytest.mark.parametrize('name', ['c2b_sep', 'c3_sep', 'c5_sep', 'c1_int', 'c2a_int'])
def test_batched_component_optimisation_equals_the_sequential_one(backend, name):
    """optimize() hands the components of a map to ttm_optimize_separable_batch / ttm_optimize_integrated_batch (host
    threads, a stream each - the reference's process pool over components, TM:2789-2845).  The problems are independent and every
    reduction is deterministic: the same bits as one component after the other, whatever the thread count; a subset
    of components and a single component (no batch) go through the same code."""
    npz, desc = load_case(name)
    runs = {}
    for threads in (1, 3, 8):
        tm = make_tm(name, npz, desc, with_coeffs=False)
        tm.optimizer_threads = threads
        tm.optimize()
        runs[threads] = ([np.array(c) for c in tm.coeffs_mon], [np.array(c) for c in tm.coeffs_nonmon], tm.objective_total)
    for threads in (3, 8):
        for k in range(len(runs[1][0])):
            assert np.array_equal(runs[threads][0][k], runs[1][0][k]) and np.array_equal(runs[threads][1][k], runs[1][1][k])
        assert runs[threads][2] == runs[1][2]
    if not name.endswith('_int'):
        # the memory-lean variant (derivative basis recomputed from the x_k column in every evaluation): the same bits
        tm = make_tm(name, npz, desc, with_coeffs=False)
        tm.direct_objective = True
        assert all(d is not None for d in tm._cm.sep_direct)
        tm.optimize()
        for k in range(tm.D):
            assert np.array_equal(tm.coeffs_mon[k], runs[1][0][k]) and np.array_equal(tm.coeffs_nonmon[k], runs[1][1][k])
    tm = make_tm(name, npz, desc, with_coeffs=False)
    tm.optimizer_batch_bytes = 1                      # one component per batch: still the same results
    tm.optimize(K=[tm.D - 1, 0])
    for k in (0, tm.D - 1):
        assert np.array_equal(tm.coeffs_mon[k], runs[1][0][k]) and np.array_equal(tm.coeffs_nonmon[k], runs[1][1][k])


def test_entf_update_with_reset(backend):
    """Example 06 filter map (X is N x 4, skip_dimensions 1, L2 lambda 0.05): reset -> optimize -> map -> inverse."""
    from triangular_transport_toolbox_amd.transport_map import transport_map
    npz, desc = load_case('entf')
    rng = np.random.default_rng(0)
    tm = transport_map(X=rng.uniform(size=(500, 4)), monotone=desc['monotone'], nonmonotone=desc['nonmonotone'],
                       polynomial_type='hermite function', monotonicity='separable monotonicity',
                       regularization='l2', regularization_lambda=0.05, verbose=False)
    tm.reset(npz['u0_map_input'])
    assert relerr(tm.X_mean, npz['u0_X_mean']) < 1e-13
    for k in range(tm.D):
        A, _ = tm.separable_setup(k)
        check('golden/entf_sep_A', relerr(A, npz['u0_sep_A_%d' % k]), 1e-10, backend)
    tm.optimize()
    for k in range(tm.D):
        assert relerr(tm.coeffs_mon[k], npz['u0_coeffs_mon_%d' % k]) < 1e-4
        assert relerr(tm.coeffs_nonmon[k], npz['u0_coeffs_nonmon_%d' % k]) < 1e-4
    tm.coeffs_mon, tm.coeffs_nonmon = coeff_lists(npz, tm.D, prefix='u0_')
    Zp = tm.map(npz['u0_map_input'])
    check('golden/entf_map', relerr(Zp, npz['u0_Z']), 1e-11, backend)
    Ystar = np.repeat(npz['obs'][0][0].reshape((1, 1)), Zp.shape[0], axis=0)
    check('golden/entf_conditional_inverse', relerr(tm.inverse_map(npz['u0_Z'], X_star=Ystar), npz['u0_ret']), 1e-11, backend)


def test_argument_errors(backend):
    from triangular_transport_toolbox_amd.transport_map import transport_map
    X = np.random.default_rng(0).standard_normal((50, 2))
    mon, non = [[[0]], [[1]]], [[[]], [[], [0]]]
    with pytest.raises(ValueError):
        transport_map(X=X, monotone=mon, nonmonotone=non, monotonicity='nonsense', verbose=False)
    with pytest.raises(ValueError):
        transport_map(X=X, monotone=mon, nonmonotone=non, ST_scale_mode='nonsense', verbose=False)
    with pytest.raises(Exception):
        transport_map(X=X, monotone=mon, nonmonotone=non, polynomial_type='nonsense', verbose=False)
    with pytest.raises(ValueError):
        transport_map(X=X, monotone=[[[0]], ['XYZ 1']], nonmonotone=non, verbose=False)
    tm = transport_map(X=X, monotone=mon, nonmonotone=non, verbose=False, quadrature_input={'order': 10})
    with pytest.raises(Exception):
        tm.reset(X[:, 0])
    with pytest.raises(AssertionError):
        tm.evaluate_pullback_density(X)
    # inputs are copied, never mutated (TM:311, 2413, 3669)
    X0, Z0 = X.copy(), np.random.default_rng(1).standard_normal((7, 2))
    Zc = Z0.copy()
    tm.map(X)
    tm.inverse_map(Z0)
    assert np.array_equal(X, X0) and np.array_equal(Z0, Zc)
    # quirk: with standardize_samples=False a user X is ignored by map() (TM:2410-2422)
    tm2 = transport_map(X=X, monotone=mon, nonmonotone=non, verbose=False, standardize_samples=False,
                        quadrature_input={'order': 10})
    assert tm2.map(X[:7]).shape == (50, 2)


def test_entf_cycles_match_reference(backend):
    """Five full assimilation cycles of the Example-06 filter (N = 500, all random draws replayed from the
    fixture): the ensemble after every cycle stays within 1e-6 of the reference's."""
    from triangular_transport_toolbox_amd import entf
    npz, desc = load_case('entf')
    ens = npz['ens0']
    tm = entf.make_filter_map(ens.shape[0], maxorder=3, lmbda=float(npz['lmbda']))
    assert len(npz['obs']) == 5
    for t in range(len(npz['obs'])):
        noises = [npz['noise_%d_%d' % (t, i)] for i in range(3)]
        Xa = entf.assimilate(tm, ens, npz['obs'][t], noises)
        check('golden/entf_cycles', relerr(Xa, npz['ens_%d_2' % t]), 1e-6, backend)
        ens = entf.rk4(Xa, 0.05, 2)
        assert relerr(ens, npz['forecast_%d' % t]) < 1e-6


def test_device_resident_filter_matches_reference_and_host_loop(backend):
    """entf.Filter - forecast, observation noise, map input, reset, optimisation, pushforward, conditional inverse all on
    the device, no host copy of the ensemble - replays the reference's five cycles (its own noise draws added on the
    device) within 1e-6, and agrees with the host-loop harness `assimilate` to rounding."""
    from triangular_transport_toolbox_amd import entf
    npz, desc = load_case('entf')
    ens = npz['ens0']
    flt = entf.Filter(ens.shape[0], maxorder=3, lmbda=float(npz['lmbda']))
    flt.set_ensemble(ens)
    tm_host = entf.make_filter_map(ens.shape[0], maxorder=3, lmbda=float(npz['lmbda']))
    host = ens
    for t in range(len(npz['obs'])):
        noises = np.stack([npz['noise_%d_%d' % (t, i)] for i in range(3)])
        flt.assimilate(npz['obs'][t], noises=noises)
        Xa = flt.ensemble()
        assert relerr(Xa, npz['ens_%d_2' % t]) < 1e-6
        host = entf.assimilate(tm_host, host, npz['obs'][t], list(noises))
        assert relerr(Xa, host) < 1e-9
        flt.forecast(0.05, 2)
        host = entf.rk4(host, 0.05, 2)
        assert relerr(flt.ensemble(), npz['forecast_%d' % t]) < 1e-6
    # own noise generator: reproducible, N(0, sd^2)
    a = entf.Filter(4000, seed=5)
    b = entf.Filter(4000, seed=5)
    for f in (a, b):
        f.set_ensemble(np.zeros((4000, 3)))
        f.tm.map_columns([-1] * 4, 4, f.N, out=f._inp)
        entf._check(f.tm._lib.ttm_perturb(f.tm._ptr(f.ens), None, 2.0, f.seed, 1, 0, f.N, f.tm._ptr(f._inp), f.tm._stream()))
    ya, yb = a._inp[0, :4000].cpu().numpy(), b._inp[0, :4000].cpu().numpy()
    assert np.array_equal(ya, yb) and abs(ya.mean()) < 0.15 and abs(ya.std() - 2.0) < 0.1


def test_filter_batch_without_threads_and_self_validating_sums_make_the_same_bits(backend, ttm_opt):
    """The optimiser batch of the filter map - two components with ONE monotone term and one with special terms - runs without
    host threads (the one-term components' single device evaluation launched ahead, csrc/ttm_optim.cpp), with self-validating
    partial sums instead of ticket + completion mark (k_objective_sep_cached, sentinel finish), and with ONE launch for the
    whole L-BFGS-B loop of the component with special terms (k_objective_sep_server: requests through a mailbox in device
    memory; option sep_server = 0: a launch per evaluation).  None of it changes an evaluation point or the order of a sum: the same coefficients, bit for bit, as component by component, as the
    ticket finish (option sep_sentinel = 0) and as the 17-launch order statistics / four-launch moments of the reset."""
    from triangular_transport_toolbox_amd import entf
    npz, desc = load_case('entf')
    ens = npz['ens0']
    noises = [npz['noise_0_%d' % i] for i in range(3)]

    def run(threads=3, **opts):
        for name, v in opts.items():
            ttm_opt(name, v)
        tm = entf.make_filter_map(ens.shape[0], maxorder=3, lmbda=float(npz['lmbda']))
        tm.optimizer_threads = threads
        out = entf.assimilate(tm, ens, npz['obs'][0], list(noises))
        for name in opts:
            ttm_opt(name, -1)
        return out, [np.array(c) for c in tm.coeffs_mon], [np.array(c) for c in tm.coeffs_nonmon]
    base = run()
    assert relerr(base[0], npz['ens_0_2']) < 1e-6
    for other in (run(threads=1), run(sep_sentinel=0), run(select_coop=0), run(sep_server=0)):
        assert np.array_equal(other[0], base[0])
        for a, b in zip(other[1] + other[2], base[1] + base[2]):
            assert np.array_equal(a, b)
    # the four-launch moments differ from the one-launch ones in the last bits of mean and standard deviation
    loose = run(colstats_one=0)
    assert relerr(loose[0], base[0]) < 1e-9


@pytest.mark.gpu
def test_a_finishing_workgroup_that_gives_up_fails_the_loop_and_leaves_nothing_behind(ttm_opt):
    """The evaluations with self-validating sums wait a bounded time for the other workgroups' partial sums; a wait that runs out
    writes a failure pattern where the host expects the results (option sep_sentinel = 2 forces it): optimize() raises, the scratch
    of the batch is dropped, and the next optimize() - partial sums re-armed - gives what an undisturbed run gives."""
    from triangular_transport_toolbox_amd import _capi
    npz, desc = load_case('c3_sep')
    ref = make_tm('c3_sep', npz, desc, with_coeffs=False)
    ref.optimize()
    tm = make_tm('c3_sep', npz, desc, with_coeffs=False)
    ttm_opt('sep_sentinel', 2)
    with pytest.raises(_capi.TTMError):
        tm.optimize()
    ttm_opt('sep_sentinel', -1)
    for k in range(tm.D):
        tm.coeffs_mon[k] = np.asarray(tm.coeffs_mon[k], dtype=float) * 0 + tm.coeffs_init
        tm.coeffs_nonmon[k] = np.asarray(tm.coeffs_nonmon[k], dtype=float) * 0 + tm.coeffs_init
    tm.optimize()
    for k in range(tm.D):
        assert np.array_equal(tm.coeffs_mon[k], ref.coeffs_mon[k]) and np.array_equal(tm.coeffs_nonmon[k], ref.coeffs_nonmon[k])


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['c2b_sep', 'c3_sep', 'c5_sep'])
def test_evaluation_server_equals_a_launch_per_evaluation(name, ttm_opt):
    """optimize() of separable maps at sizes whose grids have at most 128 workgroups: every component with more than one monotone
    term gets ONE launch for its whole loop (k_objective_sep_server, several of them side by side on the batch's streams, each with a
    mailbox of its own) - the same coefficients, bit for bit, as with a launch per evaluation; the servers have left when optimize()
    returns (the stream is idle again at once) and a second optimize() finds mailboxes."""
    import time
    import torch
    npz, desc = load_case(name)
    runs = {}
    for opt in (0, -1, -1):
        ttm_opt('sep_server', opt)
        tm = make_tm(name, npz, desc, with_coeffs=False)
        tm.optimize()
        t0 = time.perf_counter()
        torch.cuda.synchronize()
        assert time.perf_counter() - t0 < 0.05                  # (no server waiting for its 0.2 s to run out)
        runs.setdefault(opt, []).append(([np.array(c) for c in tm.coeffs_mon], [np.array(c) for c in tm.coeffs_nonmon], tm.objective_total))
    ref = runs[0][0]
    for got in runs[-1]:
        for a, b in zip(got[0] + got[1], ref[0] + ref[1]):
            assert np.array_equal(a, b)
        assert got[2] == ref[2]


@pytest.mark.gpu
def test_an_evaluation_server_that_was_left_waiting_is_replaced(tmp_path):
    """The resident workgroups of an evaluation server leave when no request arrives for 0.2 s (a host that went away must not leave
    waves behind).  A host thread that merely was not scheduled for that long (TTM_SRV_TEST_STALL = 5: 0.3 s in front of the fifth
    request of every loop; read when the library loads, hence a process of its own) finds the stream idle and its request
    unanswered, starts a new server and asks again: the same coefficients, bit for bit."""
    import os
    import subprocess
    import sys
    script = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from tests.test_transport_map import make_tm\n"
        "from tests.util import load_case\n"
        "npz, desc = load_case('c3_sep')\n"
        "tm = make_tm('c3_sep', npz, desc, with_coeffs=False)\n"
        "tm.optimize()\n"
        "np.save(sys.argv[1], np.concatenate([np.concatenate((tm.coeffs_nonmon[k], tm.coeffs_mon[k])) for k in range(tm.D)]))\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for tag, stall in (('plain', None), ('stalled', '5')):
        env = dict(os.environ)
        env.pop('TTM_SRV_TEST_STALL', None)
        if stall:
            env['TTM_SRV_TEST_STALL'] = stall
        path = str(tmp_path / (tag + '.npy'))
        res = subprocess.run([sys.executable, '-c', script, path], env=env, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr[-2000:]
        out[tag] = np.load(path)
    assert np.array_equal(out['plain'], out['stalled'])


def test_ents_backward_smoother_matches_reference(backend):
    """Ensemble Transport Smoother (example_07.py:368-465): the 6-column block map (skip_dimensions = 3, probabilist's
    Hermite polynomials with 'HF' terms, L2), three backward steps of reset -> optimize -> map -> inverse_map with
    X_star, against the reference's smoothing ensembles."""
    from triangular_transport_toolbox_amd import entf, specs
    npz, desc = load_case('ents')
    mon, non = specs.ents_smoother_spec(int(npz['maxorder']))
    assert mon == desc['monotone'] and non == desc['nonmonotone']
    tm = entf.make_smoother_map(npz['analyses'].shape[1], maxorder=int(npz['maxorder']), lmbda=float(npz['lmbda']))
    assert tm.D == 3 and tm.skip_dimensions == 3
    Xs = entf.smooth(tm, npz['forecasts'], npz['analyses'])
    assert Xs.shape == npz['smoothed'].shape
    for t in range(len(Xs)):
        check('golden/ents_smoothed', relerr(Xs[t], npz['smoothed'][t]), 1e-6, backend)
    for k in range(tm.D):                     # (coefficients of the last update, t = 0)
        assert relerr(tm.coeffs_mon[k], npz['last_coeffs_mon_%d' % k]) < 1e-4
        assert relerr(tm.coeffs_nonmon[k], npz['last_coeffs_nonmon_%d' % k]) < 1e-4


def test_packed_coefficients_held_across_reset_are_refolded(backend):
    """forward_device / inverse_device callers may keep a packed coefficient vector; the folded coefficients, U-form
    splines and inverse tables cached with it belong to one special-term placement.  After reset() the vector is folded
    again on its next use instead of evaluating with stale splines."""
    from triangular_transport_toolbox_amd.transport_map import transport_map
    npz, desc = load_case('c3_sep')
    X = case_X('c3_sep', npz)
    tm = transport_map(X=X[:1000], monotone=desc['monotone'], nonmonotone=desc['nonmonotone'], verbose=False, **ctor_kwargs(desc))
    mon, non = coeff_lists(npz, tm.D)
    tm.coeffs_mon, tm.coeffs_nonmon = [c.copy() for c in mon], [c.copy() for c in non]
    coef = tm._pack_coeffs()
    Z0 = tm.forward_device(tm._Xs, tm._N, coef=coef)[:, :tm._N].cpu().numpy()
    tm.reset(X[1000:2000] * 1.7 + 0.3)                   # other samples: other special-term centres / scales
    tm.coeffs_mon, tm.coeffs_nonmon = [c.copy() for c in mon], [c.copy() for c in non]
    epoch = coef._ttm_epoch
    Z1 = tm.forward_device(tm._Xs, tm._N, coef=coef)[:, :tm._N].cpu().numpy()
    assert coef._ttm_epoch != epoch
    assert np.array_equal(Z1, tm.forward_device(tm._Xs, tm._N)[:, :tm._N].cpu().numpy())      # == a fresh pack
    assert Z0.shape == Z1.shape and not np.allclose(Z0, Z1)
    Xi = tm.inverse_device(tm.forward_device(tm._Xs, tm._N), tm._N, coef=coef)
    assert np.array_equal(Xi[:, :tm._N].cpu().numpy(), tm.inverse_device(tm.forward_device(tm._Xs, tm._N), tm._N)[:, :tm._N].cpu().numpy())


def test_nearly_collinear_nonmonotone_basis_takes_the_qr_path(backend):
    """The reduced separable problem is formed from the Gram matrix (one device pass) when that is safe and by the
    reference's Householder QR otherwise: two nonmonotone terms that differ by 1e-7 of a third make the equilibrated
    Gram matrix singular to working precision; A and the nonmonotone coefficients still match the oracle's."""
    from triangular_transport_toolbox_amd.transport_map import transport_map
    from oracle.ttm_oracle import OracleMap
    rng = np.random.default_rng(2)
    x0 = rng.standard_normal(3000)
    X = np.column_stack((x0, x0 * (1 + 1e-7 * rng.standard_normal(3000)), 0.5 * x0 + rng.standard_normal(3000)))
    monotone = [['LET 0', 'RET 0'], ['LET 1', 'RET 1'], ['LET 2', 'iRBF 2', 'RET 2']]
    nonmonotone = [[[]], [[], [0]], [[], [0], [1], [0, 0, 'HF']]]
    kw = dict(polynomial_type='hermite function', monotonicity='separable monotonicity', standardize_samples=True)
    tm = transport_map(X=X, monotone=monotone, nonmonotone=nonmonotone, verbose=False, **kw)
    om = OracleMap(X=X, monotone=monotone, nonmonotone=nonmonotone, **kw)
    k = 2
    G = tm._gram(k)
    n_nm = int(tm._cm.n_nm[k])
    assert tm._normal_solve(G[:n_nm, :n_nm], G[:n_nm, n_nm:], 0.0) is None          # the normal equations are refused ...
    A, solve = tm.separable_setup(k)                                                  # ... and the QR path runs
    Ao, aux = om.separable_setup(k)
    assert relerr(A, Ao) < 1e-8
    c = np.array([0.7, 0.2, 0.9])
    assert relerr(tm.separable_objective(c, A, k)[0], om.separable_objective(c, Ao, k)[0]) < 1e-9
    # a well-conditioned component of the same map keeps the Gram path and agrees as before
    A1, _ = tm.separable_setup(1)
    assert relerr(A1, om.separable_setup(1)[0]) < 1e-10
