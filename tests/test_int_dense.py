"""The dense integrated-rectifier path (csrc/ttm_dense.h bodies, csrc/ttm_int.hip kernels: monomial form of g, Horner per
quadrature node) against the generic evaluators it replaces (option int_dense = 0) and against the reference goldens, on
both backends; on the GPU the kernels are asserted by name."""
import os

import numpy as np
import pytest

from tests.hostemu import emu
from tests.util import INTEGRATED, case_X, check, coeff_lists, ctor_kwargs, load_case, make_oracle, relerr


@pytest.fixture(params=[pytest.param('hostemu'), pytest.param('hip', marks=pytest.mark.gpu)])
def backend(request):
    if request.param == 'hostemu':
        with emu.install():
            yield 'hostemu'
    else:
        yield 'hip'


class generic_path:
    """int_dense = 0 for the duration of the block (library option on the GPU, environment switch of the test double)."""
    def __init__(self, tm, backend):
        self.tm, self.backend = tm, backend

    def __enter__(self):
        if self.backend == 'hip':
            self.tm._lib.ttm_set_option(b'int_dense', 0)
        else:
            os.environ['TTM_INT_DENSE'] = '0'

    def __exit__(self, *exc):
        if self.backend == 'hip':
            self.tm._lib.ttm_reset_options()
        else:
            del os.environ['TTM_INT_DENSE']


def make_tm(name, npz, desc, **extra):
    from triangular_transport_toolbox_amd.transport_map import transport_map
    kw = ctor_kwargs(desc)
    kw.update(extra)
    tm = transport_map(X=case_X(name, npz), monotone=desc['monotone'], nonmonotone=desc['nonmonotone'], verbose=False, **kw)
    tm.coeffs_mon, tm.coeffs_nonmon = coeff_lists(npz, tm.D)
    return tm


def last_kernel(tm):
    import ctypes
    tm._lib.ttm_last_kernel.restype = ctypes.c_char_p
    return tm._lib.ttm_last_kernel().decode()


DENSE = [c for c in INTEGRATED if c != 'misc_grid']          # (misc_grid: special-term cross grids - no dense B set)


@pytest.mark.parametrize('name', DENSE)
def test_dense_and_generic_paths_agree(backend, name):
    npz, desc = load_case(name)
    tm = make_tm(name, npz, desc, alternate_root_finding=False)
    flags = tm._cm.complex
    assert all(int(f) & 4 for f in flags), 'fixture should have polynomial B sets'
    X = case_X(name, npz)[:npz['Z'].shape[0]]
    Z = tm.map(X)
    if backend == 'hip':                      # (kernels by name on the device entry points: map() ends on the layout change)
        Zs = tm.forward_device(tm._Xs, tm._N)
        assert last_kernel(tm) == 'k_int_forward'
        tm.inverse_device(Zs, tm._N, table=False)
        assert last_kernel(tm) == 'k_int_root<bisect>'
        tm._device_sums(tm.D - 1, np.concatenate((tm.coeffs_nonmon[tm.D - 1], tm.coeffs_mon[tm.D - 1])))
        assert last_kernel(tm) == 'k_int_objective'
    check('dense/map[%s]' % name, relerr(Z, npz['Z']), 1e-11, backend)
    Zin = npz['inv_Z'] if 'inv_Z' in npz else np.random.default_rng(5).standard_normal((64, tm.D))
    Xi = tm.inverse_map(Zin)
    k = tm.D - 1
    div = len(tm.coeffs_nonmon[k])
    c = np.concatenate((tm.coeffs_nonmon[k], tm.coeffs_mon[k])) * 1.01
    J, G = tm.objective_function(c, k, div), tm.objective_function_jacobian(c, k, div)
    tm.root_finder = 'newton'
    Xn = tm.inverse_map(Zin)
    tm.root_finder = 'reference'
    with generic_path(tm, backend):
        Zg = tm.map(X)
        if backend == 'hip':
            tm.forward_device(tm._Xs, tm._N)
            assert last_kernel(tm) in ('k_forward', 'k_forward_plan')
        Xg = tm.inverse_map(Zin)
        tm.root_finder = 'newton'
        Xng = tm.inverse_map(Zin)
        tm.root_finder = 'reference'
        tm._obj_cache = None                       # (the class keeps the sums of the last coefficient vector)
        Jg, Gg = tm.objective_function(c, k, div), tm.objective_function_jacobian(c, k, div)
    check('dense/map_vs_generic[%s]' % name, relerr(Z, Zg), 1e-12, backend)
    # (the two searches follow the same midpoint sequence unless a residual within rounding of +-1e-9 or of 0 falls the
    # other way: positions then differ by less than the final bracket)
    check('dense/bisection_vs_generic[%s]' % name, relerr(Xi, Xg), 1e-7, backend)
    check('dense/objective_vs_generic[%s]' % name, abs(J - Jg) / (1 + abs(Jg)), 1e-12, backend)
    check('dense/gradient_vs_generic[%s]' % name, relerr(G, Gg), 1e-12, backend)
    # Newton lands on the root of the same map: residual under the oracle's forward map (random targets of the fixtures
    # without inverse data may lie where exp(polynomial) overflows: both searches then end on the same non-finite rows)
    ok = np.all(np.isfinite(Xng), axis=1)
    assert np.array_equal(np.all(np.isfinite(Xn), axis=1), ok) and ok.sum() >= len(ok) // 2
    check('dense/newton_vs_generic[%s]' % name, relerr(Xn[ok], Xng[ok]), 1e-7, backend)
    om = make_oracle(name, npz, desc)
    skipcols = np.zeros((len(Xn), om.skip_dimensions)) + om.X_mean[:om.skip_dimensions]
    with np.errstate(all='ignore'):
        res = np.abs(om.map(np.column_stack((skipcols, Xn))) - Zin)[ok]
    if 'inv_Z' in npz:              # (random targets need not be attainable: the search then ends where the generic one does)
        check('dense/newton_residual[%s]' % name, float(np.max(res[np.isfinite(res)])), 2e-9, backend)


def test_nan_and_infinite_samples_stay_nan(backend):
    """The node's exp clamps its argument (a NaN would come out as a number): the dense bodies add a probe term, so a NaN or
    infinite entry makes exactly the components that read it NaN, as the reference's arithmetic does."""
    npz, desc = load_case('c3_int')
    tm = make_tm('c3_int', npz, desc)
    X = case_X('c3_int', npz)[:64].copy()
    X[3, 1] = np.nan
    X[7, 2] = np.inf
    Z = tm.map(X)
    om = make_oracle('c3_int', npz, desc)
    with np.errstate(all='ignore'):
        Zo = om.map(X)
    ok = np.isfinite(Zo)
    assert np.array_equal(np.isfinite(Z), ok)
    assert relerr(Z[ok], Zo[ok]) < 1e-11


def test_example01_order10_map_takes_the_dense_kernels(backend):
    """Example 01's order-10 map (example_01.py:121-170: Hermite-function orders 1..10, the reference's shipped configuration):
    order class (10, 0) of the monomial-form kernels (round 4 stopped at order 8 and sent it to the generic interpreter);
    map against the reference's values, objective and gradient through the X program against the generic kernels."""
    from tests.test_transport_map import make_tm as make_any
    npz, desc = load_case('ex01_order10')
    tm = make_any('ex01_order10', npz, desc)
    Z = tm.map(npz['X_head'])
    if backend == 'hip':
        tm.forward_device(tm._Xs, tm._N)
        assert last_kernel(tm) == 'k_int_forward'
    check('dense/map[ex01_order10]', relerr(Z, npz['Z_head']), 1e-11, backend)
    for k in range(tm.D):
        div = len(tm.coeffs_nonmon[k])
        c = np.concatenate((tm.coeffs_nonmon[k], tm.coeffs_mon[k]))
        tm._obj_cache = None
        J, G = tm.objective_function(c, k, div), tm.objective_function_jacobian(c, k, div)
        if backend == 'hip':
            tm._device_sums(k, c)
            assert last_kernel(tm) == 'k_int_objective'
        with generic_path(tm, backend):
            tm._obj_cache = None
            Jg, Gg = tm.objective_function(c, k, div), tm.objective_function_jacobian(c, k, div)
            if backend == 'hip':
                tm._device_sums(k, c)
                assert last_kernel(tm) == 'k_objective'
        check('dense/objective_vs_generic[ex01_order10]', abs(J - Jg) / (1 + abs(Jg)), 1e-12, backend)
        check('dense/gradient_vs_generic[ex01_order10]', relerr(G, Gg), 1e-11, backend)


class walk_path:
    """int_xprog = 0 for the duration of the block: the kernels that walk the term tables per sample."""
    def __init__(self, tm, backend):
        self.tm, self.backend = tm, backend

    def __enter__(self):
        if self.backend == 'hip':
            self.tm._lib.ttm_set_option(b'int_xprog', 0)
        else:
            os.environ['TTM_INT_XPROG'] = '0'

    def __exit__(self, *exc):
        if self.backend == 'hip':
            self.tm._lib.ttm_reset_options()
        else:
            del os.environ['TTM_INT_XPROG']


@pytest.mark.parametrize('name', DENSE + ['ex01_order10'])
def test_x_program_and_term_table_walk_agree(backend, name):
    """Everything that goes through the components' X programs (csrc/ttm_xprog.h: forward map, bisection and Newton root
    searches, objective / gradient sums) against the kernels that walk the term tables per sample (option int_xprog = 0)."""
    npz, desc = load_case(name)
    if name == 'ex01_order10':
        from tests.test_transport_map import make_tm as make_any
        tm = make_any(name, npz, desc)
        tm.alternate_root_finding = False
        X = npz['X_head']
    else:
        tm = make_tm(name, npz, desc, alternate_root_finding=False)
        X = case_X(name, npz)[:npz['Z'].shape[0]]
    assert all(int(f) & 16 for f in tm._cm.complex), 'fixture should have X programs'
    Zin = npz['inv_Z'] if 'inv_Z' in npz else np.random.default_rng(5).standard_normal((64, tm.D))

    def everything(names):
        out = {'Z': tm.map(X)}
        if backend == 'hip':
            Zs = tm.forward_device(tm._Xs, tm._N)
            assert last_kernel(tm) == names[0]
            tm.inverse_device(Zs, tm._N, table=False)
            assert last_kernel(tm) == names[1]
        out['Xi'] = tm.inverse_map(Zin)
        tm.root_finder = 'newton'
        out['Xn'] = tm.inverse_map(Zin)
        tm.root_finder = 'reference'
        out['J'], out['G'] = [], []
        for k in range(tm.D):
            div = len(tm.coeffs_nonmon[k])
            c = np.concatenate((tm.coeffs_nonmon[k], tm.coeffs_mon[k])) * 0.97 + 0.005
            tm._obj_cache = None
            out['J'].append(tm.objective_function(c, k, div))
            out['G'].append(tm.objective_function_jacobian(c, k, div))
            if backend == 'hip' and len(c) <= 128:
                tm._device_sums(k, c)
                assert last_kernel(tm) == names[2]
        return out
    if backend == 'hip':
        tm._lib.ttm_set_option(b'int_xprog', 2)          # (the root searches through the X programs as well: off by default)
    else:
        os.environ['TTM_INT_XPROG'] = '2'
    try:
        a = everything(('k_int_forward', 'k_int_root_x<bisect>', 'k_int_objective'))
    finally:
        if backend == 'hip':
            tm._lib.ttm_reset_options()
        else:
            del os.environ['TTM_INT_XPROG']
    with walk_path(tm, backend):
        b = everything(('k_int_forward<walk>', 'k_int_root<bisect>', 'k_int_objective_walk'))
    with np.errstate(all='ignore'):
        check('xprog/map_vs_walk[%s]' % name, relerr(a['Z'], b['Z']), 1e-12, backend)
        ok = np.all(np.isfinite(b['Xi']), axis=1) & np.all(np.isfinite(b['Xn']), axis=1)
        assert np.array_equal(np.all(np.isfinite(a['Xi']), axis=1) & np.all(np.isfinite(a['Xn']), axis=1), ok)
        check('xprog/bisection_vs_walk[%s]' % name, relerr(a['Xi'][ok], b['Xi'][ok]), 1e-7, backend)
        check('xprog/newton_vs_walk[%s]' % name, relerr(a['Xn'][ok], b['Xn'][ok]), 1e-7, backend)
        for k in range(tm.D):
            check('xprog/objective_vs_walk[%s]' % name, abs(a['J'][k] - b['J'][k]) / (1 + abs(b['J'][k])), 1e-12, backend)
            check('xprog/gradient_vs_walk[%s]' % name, float(np.max(np.abs(a['G'][k] - b['G'][k]) / (1.0 + np.abs(b['G'][k])))), 1e-11, backend)
