"""
Kernel-level checks through the C ABI (backend 'hip' on the GPU, 'hostemu' on CPU):
order statistics (K9), layout change + moments (K0/K1), the several-samples-per-thread kernel variants,
and size-independent properties at benchmark-like sizes.
"""
import ctypes
import os

import numpy as np
import pytest

from tests.hostemu import emu
from tests.util import case_X, coeff_lists, ctor_kwargs, load_case, make_oracle, relerr


@pytest.fixture(params=[pytest.param('hostemu'), pytest.param('hip', marks=pytest.mark.gpu)])
def backend(request):
    if request.param == 'hostemu':
        with emu.install():
            yield 'hostemu'
    else:
        yield 'hip'


def small_map(**kw):
    from triangular_transport_toolbox_amd.transport_map import transport_map
    X = np.random.default_rng(0).standard_normal((64, 2))
    return transport_map(X=X, monotone=[[[0]], [[1]]], nonmonotone=[[[]], [[], [0]]], verbose=False,
                         quadrature_input={'order': 5}, **kw)


@pytest.mark.parametrize('coop', [-1, 0, 2])
def test_order_statistics_exact(backend, ttm_opt, coop):
    # coop -1: columns of up to 131 072 rows through the one-launch select (k_select_coop: keys in registers, a grid barrier
    # per pass), longer ones through the 17 launches; 0: the launches for every length; 2: the one-launch select with every
    # wait given up at once - what happens when the grid is not co-resident: workgroup 0 selects by itself (select_solo)
    ttm_opt('select_coop', coop)
    tm = small_map()
    rng = np.random.default_rng(1)
    for n in (1, 2, 17, 1000, 2048, 2049, 100003, 131072, 131073):
        x = rng.standard_normal(n) * 5
        x[rng.integers(0, n, max(n // 10, 1))] = x[0]               # ties
        if n > 10:
            x[:3] = [0.0, -0.0, 1e-310]
            x[3:7] = [np.inf, -np.inf, -1e-310, 1e300]
        if n == 2048:
            x[:] = 1.25                                             # one value only
        xs = np.sort(x)
        ranks = np.unique(np.clip(rng.integers(0, n, 20), 0, n - 1))
        ranks = np.unique(np.concatenate((ranks, [0, n - 1])))
        got, nn = tm._order_statistics(tm._to_dev(x), ranks)
        assert nn == n
        assert np.array_equal(got, xs[ranks])
        for q in ([0.5], [0.2, 0.4, 0.6, 0.8]):
            assert np.array_equal(tm._device_quantile(tm._to_dev(x), q), np.quantile(x, q))


def test_import_export_and_moments(backend):
    tm = small_map()
    rng = np.random.default_rng(2)
    for (N, d) in ((2, 1), (63, 3), (257, 65), (5000, 40)):
        X = rng.standard_normal((N, d)) * rng.uniform(0.5, 3, d) + rng.uniform(-2, 2, d)
        tm.standardization = 'standard'
        tm.standardize(X)
        assert relerr(tm.X_mean, X.mean(axis=0)) < 1e-13 and relerr(tm.X_std, X.std(axis=0)) < 1e-13
        Xs = tm._import(X, True)
        ref = (X - tm.X_mean) / tm.X_std
        assert Xs.shape[1] % 2 == 0 and Xs.shape[1] >= N            # padded leading dimension (16-byte aligned columns)
        assert np.array_equal(Xs[:, :N].cpu().numpy(), ref.T)       # same two IEEE operations per element
        back = tm._export(Xs, N, 0, d, True)
        assert np.array_equal(back, ref * tm.X_std + tm.X_mean)
        assert np.array_equal(tm._export(Xs, N, 1 if d > 1 else 0, d - (1 if d > 1 else 0), False), ref[:, (1 if d > 1 else 0):])


def test_moments_and_standardisation_of_device_resident_columns(backend, ttm_opt):
    """reset_device standardises column-major samples without a row-major copy (ttm_colstats_cols: up to 8 columns and 131 072
    rows in ONE launch - per-workgroup means and squared deviations, Chan's combination in workgroup order -, then
    ttm_standardize_cols).  Against NumPy, against the four-launch moments of the exported rows (option colstats_one = 0), and - on
    the device library - bit for bit against ttm_colstats of the row-major copy, whatever the offset of the data."""
    import torch
    tm = small_map()
    rng = np.random.default_rng(3)
    for (N, d, offset) in ((1, 1, 0.0), (5, 2, 0.0), (1023, 4, 1e6), (4097, 8, -3.0), (100003, 4, 25.0), (131072, 3, 0.0)):
        X = rng.standard_normal((N, d)) * rng.uniform(0.5, 3, d) + rng.uniform(-2, 2, d) + offset
        if N > 100:
            X[0] += 40.0                                                 # an outlier in the first row
        Xc = tm._cols(d, N, zero=True)
        Xc[:, :N].copy_(torch.from_numpy(np.ascontiguousarray(X.T)))
        mean, sd = tm._empty(d), tm._empty(d)
        work = tm._workspace(tm._lib.ttm_colstats_work_size(N, d))
        rc = tm._lib.ttm_colstats_cols(tm._ptr(Xc), Xc.shape[1], N, d, tm._ptr(mean), tm._ptr(sd), tm._ptr(work), tm._stream())
        assert rc == 0
        m_, s_ = mean.cpu().numpy(), sd.cpu().numpy()
        sd_ref = X.std(axis=0)
        assert relerr(m_, X.mean(axis=0)) < 1e-13
        assert np.all(np.abs(s_ - sd_ref) <= 1e-13 * np.maximum(sd_ref, 1e-300)) or N == 1
        Xs = tm._cols(d, N, zero=True)
        assert tm._lib.ttm_standardize_cols(tm._ptr(Xc), Xc.shape[1], N, d, tm._ptr(mean), tm._ptr(sd), tm._ptr(Xs), Xs.shape[1],
                                            tm._stream()) == 0
        if N > 1:
            assert np.array_equal(Xs[:, :N].cpu().numpy(), ((X - m_) / s_).T)
        # the row-major entry point: the same sums
        Xr = tm._to_dev(X)
        mean2, sd2 = tm._empty(d), tm._empty(d)
        assert tm._lib.ttm_colstats(tm._ptr(Xr), N, d, tm._ptr(mean2), tm._ptr(sd2), tm._ptr(work), tm._stream()) == 0
        if backend == 'hip':
            assert np.array_equal(mean2.cpu().numpy(), m_) and np.array_equal(sd2.cpu().numpy(), s_)
        ttm_opt('colstats_one', 0)
        assert tm._lib.ttm_colstats(tm._ptr(Xr), N, d, tm._ptr(mean2), tm._ptr(sd2), tm._ptr(work), tm._stream()) == 0
        assert relerr(mean2.cpu().numpy(), m_) < 1e-13
        assert np.all(np.abs(sd2.cpu().numpy() - s_) <= 1e-12 * np.maximum(s_, 1e-300)) or N == 1
        ttm_opt('colstats_one', -1)
    # outside the one-launch range the column-major entry declines (the caller exports and takes ttm_colstats)
    Xc = tm._cols(9, 64, zero=True)
    mean, sd = tm._empty(9), tm._empty(9)
    work = tm._workspace(tm._lib.ttm_colstats_work_size(64, 9))
    assert tm._lib.ttm_colstats_cols(tm._ptr(Xc), Xc.shape[1], 64, 9, tm._ptr(mean), tm._ptr(sd), tm._ptr(work), tm._stream()) == -4


def test_gram_matrices_of_a_batch_equal_component_by_component(backend):
    """ttm_gram_many (one launch + one reduction for the components of an optimiser batch) against ttm_gram per component: the
    same tiles and the same order of the partial sums - bit for bit on the device library."""
    from triangular_transport_toolbox_amd.transport_map import transport_map
    npz, desc = load_case('c3_sep')
    X = case_X('c3_sep', npz)
    tm = transport_map(X=X, monotone=desc['monotone'], nonmonotone=desc['nonmonotone'], verbose=False, **ctor_kwargs(desc))
    K = list(range(tm.D))
    many = tm._gram_many(K)
    for k in K:
        one = tm._gram(k)
        assert many[k].shape == one.shape
        if backend == 'hip':
            assert np.array_equal(many[k], one)
        else:
            assert relerr(many[k], one) < 1e-14


@pytest.mark.parametrize('name,ns', [('c3_sep', '2'), ('c5_sep', '2'), ('c2a_int', '2'), ('misc_grid', '2'), ('c3_sep', '4')])
def test_multi_sample_kernel_variants(backend, name, ns, monkeypatch, ttm_opt):
    """The 2- and 4-samples-per-thread kernels (chosen automatically for large ensembles) give the same
    results as the one-sample kernels, including ragged tails."""
    from triangular_transport_toolbox_amd.transport_map import transport_map
    npz, desc = load_case(name)
    X = case_X(name, npz)[:777]
    tm = transport_map(X=X, monotone=desc['monotone'], nonmonotone=desc['nonmonotone'], verbose=False, **ctor_kwargs(desc))
    tm.coeffs_mon, tm.coeffs_nonmon = coeff_lists(npz, tm.D)
    ttm_opt('forward_ns', int('1'))
    ttm_opt('inverse_ns', int('1'))
    ttm_opt('u_ns', int('1'))
    ttm_opt('u_loader', int('0'))
    Z1 = tm.map(X)
    sep = desc['kwargs']['monotonicity'] == 'separable monotonicity'
    if sep:
        p1 = tm.evaluate_pullback_density(X)
        I1 = tm.inverse_map(npz['inv_Z'])
    ttm_opt('forward_ns', int(ns))
    ttm_opt('inverse_ns', int('2'))
    ttm_opt('u_ns', int(ns))
    assert np.array_equal(tm.map(X), Z1)
    if sep:
        assert np.array_equal(tm.evaluate_pullback_density(X), p1)
        assert np.array_equal(tm.inverse_map(npz['inv_Z']), I1)
        # the loader-wave kernels (U-form maps; chosen automatically for large ensembles) - odd N, ragged last tile
        # (hot-record kernels take exp(-x^2/4) from the 2^(j/32) table: same values to rounding, not bit for bit)
        ttm_opt('u_loader', int('1'))
        ZL = tm.map(X)
        assert relerr(ZL, Z1) < 1e-12
        assert relerr(tm.evaluate_pullback_density(X), p1) < 1e-10
        assert np.array_equal(tm.map(X[:1]), ZL[:1])
        assert np.array_equal(tm.map(X[:513]), ZL[:513])
        ttm_opt('hl_ns', int('4'))                       # four samples per evaluating thread (1024-row tiles)
        assert np.array_equal(tm.map(X), ZL)
        assert np.array_equal(tm.map(X[:1025]), ZL[:1025])
        I4 = tm.inverse_map(npz['inv_Z'])
        ttm_opt('hl_ns', int('2'))
        assert np.array_equal(tm.inverse_map(npz['inv_Z']), I4)
        # (the loader-wave inverse evaluates the offsets in U-form: same values to rounding, not bit for bit)
        assert relerr(tm.inverse_map(npz['inv_Z']), I1) < 1e-12
        assert relerr(tm.inverse_map(npz['inv_Z'][:1]), I1[:1]) < 1e-12


def test_large_ensemble_properties(backend):
    """Benchmark-shaped run (C3 map, N = 3e5 on the GPU / 2e4 on the CPU double): forward values agree with
    the oracle on a strided subset; the table inverse reproduces the oracle's; linearity of the map in its
    coefficients; checksum of the ensemble is permutation-equivariant."""
    from triangular_transport_toolbox_amd import specs
    from triangular_transport_toolbox_amd.transport_map import transport_map
    N = 300000 if backend == 'hip' else 20000
    cfg = specs.config('C3')
    X = cfg['sampler'](N)
    npz, desc = load_case('c3_sep')
    tm = transport_map(X=X, monotone=cfg['monotone'], nonmonotone=cfg['nonmonotone'], verbose=False, **cfg['kwargs'])
    tm.coeffs_mon, tm.coeffs_nonmon = coeff_lists(npz, tm.D)
    from oracle.ttm_oracle import OracleMap
    om = OracleMap(X=X, monotone=cfg['monotone'], nonmonotone=cfg['nonmonotone'], **cfg['kwargs'])
    om.coeffs_mon, om.coeffs_nonmon = coeff_lists(npz, tm.D)
    assert relerr(tm.X_mean, om.X_mean) < 1e-12 and relerr(tm.X_std, om.X_std) < 1e-12
    Z = tm.map(X)
    idx = np.arange(0, N, max(N // 1500, 1))
    assert relerr(Z[idx], om.map(X[idx])) < 1e-10
    Zin = specs.reference_samples(N, 4)
    Xinv = tm.inverse_map(Zin)
    assert relerr(Xinv[idx], om.inverse_map(Zin[idx])) < 1e-9
    # permutation equivariance (exact): every sample is processed independently
    perm = np.random.default_rng(0).permutation(N)
    assert np.array_equal(tm.map(X[perm]), Z[perm])
    # linearity in the coefficients (separable maps): S(2c) = 2 S(c)
    tm.coeffs_mon = [2 * c for c in tm.coeffs_mon]
    tm.coeffs_nonmon = [2 * c for c in tm.coeffs_nonmon]
    assert relerr(tm.map(X[idx]), 2 * Z[idx]) < 1e-13


@pytest.mark.parametrize('name', ['c3_sep', 'c1_int', 'misc_sep'])
def test_nan_and_inf_samples_propagate(backend, name):
    """A NaN in a sample makes exactly the components that read it NaN (as NumPy does in the reference); the
    other samples are untouched.  +-inf inputs stay finite or inf, never crash."""
    from triangular_transport_toolbox_amd.transport_map import transport_map
    npz, desc = load_case(name)
    X = case_X(name, npz)[:300].copy()
    tm = transport_map(X=case_X(name, npz), monotone=desc['monotone'], nonmonotone=desc['nonmonotone'], verbose=False,
                       **ctor_kwargs(desc))
    tm.coeffs_mon, tm.coeffs_nonmon = coeff_lists(npz, tm.D)
    om = make_oracle(name, npz, desc)
    Zc = tm.map(X)
    Xn = X.copy()
    Xn[5, 0] = np.nan
    Xn[9, -1] = np.nan
    Z = tm.map(Xn)
    with np.errstate(all='ignore'):
        Zo = om.map(Xn)
    assert np.array_equal(np.isnan(Z), np.isnan(Zo))
    ok = ~np.isnan(Zo)
    assert relerr(Z[ok], Zo[ok]) < 1e-11
    rows = np.ones(len(X), dtype=bool)
    rows[[5, 9]] = False
    assert np.array_equal(Z[rows], Zc[rows])
    Xi = X.copy()
    Xi[7, 0] = np.inf
    Zi = tm.map(Xi)
    assert np.array_equal(Zi[rows & (np.arange(len(X)) != 7)], Zc[rows & (np.arange(len(X)) != 7)])


@pytest.mark.parametrize('name', ['c3_sep', 'c5_sep'])
def test_separable_objective_from_cached_basis(backend, name):
    """The one-launch evaluation on the cached derivative basis (ttm_objective_sep_cached) gives the sums of the
    basis-recomputing reduction (ttm_objective), for odd N and several components."""
    from triangular_transport_toolbox_amd.transport_map import transport_map
    npz, desc = load_case(name)
    X = case_X(name, npz)[:2001]
    tm = transport_map(X=X, monotone=desc['monotone'], nonmonotone=desc['nonmonotone'], verbose=False, **ctor_kwargs(desc))
    rng = np.random.default_rng(4)
    for k in sorted({0, 1, tm.D - 1}):
        A, _ = tm.separable_setup(k)
        c = 0.05 + rng.random(int(tm._cm.n_mon[k]))
        J0, G0 = tm.separable_objective(c, A, k)
        tm._sep_cache_begin(k)
        assert tm._sep_cache is not None
        J1, G1 = tm.separable_objective(c, A, k)
        J2, G2 = tm.separable_objective(2 * c, A, k)
        tm._sep_cache_end()
        assert abs(J1 - J0) <= 1e-12 * (1 + abs(J0)) and relerr(G1, G0) < 1e-12
        J3, G3 = tm.separable_objective(2 * c, A, k)
        assert abs(J2 - J3) <= 1e-12 * (1 + abs(J3)) and relerr(G2, G3) < 1e-12


@pytest.mark.gpu
def test_full_size_c5_properties():
    """BASELINE configuration C5 at its full size (d = 40, N = 10^6) through the loader-wave kernels: oracle
    agreement on a strided subset (forward, table inverse, pullback density), permutation equivariance (exact: samples
    are independent), the round trip S(S^-1(z)) against the table resolution, and the fused sum of squares."""
    from triangular_transport_toolbox_amd import specs
    from triangular_transport_toolbox_amd.transport_map import transport_map
    from oracle.ttm_oracle import OracleMap
    N = 1000000
    cfg = specs.config('C5')
    X = cfg['sampler'](N)
    npz, desc = load_case('c5_sep')
    tm = transport_map(X=X, monotone=cfg['monotone'], nonmonotone=cfg['nonmonotone'], verbose=False, **cfg['kwargs'])
    tm.coeffs_mon, tm.coeffs_nonmon = coeff_lists(npz, tm.D)
    assert tm._cm.u_enabled and tm._cm.u_h_cls > 0
    idx = np.arange(0, N, 4001)
    om = OracleMap(X=X, monotone=cfg['monotone'], nonmonotone=cfg['nonmonotone'], **cfg['kwargs'])   # (moments, quantiles)
    om.coeffs_mon, om.coeffs_nonmon = coeff_lists(npz, tm.D)
    assert relerr(tm.X_mean, om.X_mean) < 1e-12 and relerr(tm.X_std, om.X_std) < 1e-12
    Z = tm.map(X)
    assert relerr(Z[idx], om.map(X[idx])) < 1e-11
    perm = np.random.default_rng(3).permutation(N)
    assert np.array_equal(tm.map(X[perm]), Z[perm])
    Xi = tm.inverse_map(Z)
    assert relerr(Xi[idx], om.inverse_map(Z[idx])) < 1e-11
    # round trip: the table inverse interpolates linearly between 1001 points on [-10, 10] (TM:4039-4082)
    assert np.max(np.abs(Xi - X) / tm.X_std) < 5e-4
    assert relerr(tm.evaluate_pullback_density(X[idx]), om.evaluate_pullback_density(X[idx])) < 1e-11
    # fused sum of squares of the device entry point == row norms of Z
    ss = tm._empty(N)
    tm.forward_device(tm._Xs, N, sumsq=ss)
    assert relerr(ss.cpu().numpy(), np.sum(Z * Z, axis=1)) < 1e-12


@pytest.mark.gpu
def test_c_abi_reports_bad_arguments_instead_of_launching():
    """Every entry point validates its arguments on the host and returns a negative TTM_E_* code with a message
    (include/ttm.h): no kernel is launched on a null pointer, an empty ensemble, a leading dimension shorter than N or
    a component range outside the map."""
    import ctypes
    from triangular_transport_toolbox_amd import _capi
    tm = small_map()
    lib = tm._lib
    coef = tm._pack_coeffs()
    N = tm._N
    Z = tm._cols(tm.D, N)
    st = tm._stream()
    p, c, f, X = tm._pp, tm._ptr(coef), tm._ptr(coef._ttm_fold), tm._ptr(tm._Xs)
    ldx = tm._Xs.shape[1]

    def expect(rc, code):
        assert rc == code, (rc, lib.ttm_last_error_string().decode())
        assert len(lib.ttm_last_error_string().decode()) > 0

    expect(lib.ttm_forward(p, c, f, None, ldx, N, 0, tm.D, tm._ptr(Z), Z.shape[1], None, None, None, st), -1)       # null X
    expect(lib.ttm_forward(p, c, f, X, ldx, 0, 0, tm.D, tm._ptr(Z), Z.shape[1], None, None, None, st), -1)          # N = 0
    expect(lib.ttm_forward(p, c, f, X, N - 1, N, 0, tm.D, tm._ptr(Z), Z.shape[1], None, None, None, st), -1)        # ldx < N
    expect(lib.ttm_forward(p, c, f, X, ldx, N, 0, tm.D + 1, tm._ptr(Z), Z.shape[1], None, None, None, st), -1)      # k1 > D
    expect(lib.ttm_forward(p, c, f, X, ldx, N, 2, 1, tm._ptr(Z), Z.shape[1], None, None, None, st), -1)             # k0 > k1
    iters = tm._zeros(tm.D, dtype=__import__('torch').int32)
    expect(lib.ttm_inverse_bisect(p, c, f, 0, tm.D, None, Z.shape[1], X, ldx, N, ctypes.c_void_p(iters.data_ptr()), None, st), -1)
    expect(lib.ttm_inverse_newton(p, c, f, 0, tm.D, tm._ptr(Z), Z.shape[1], X, ldx, N, None, st), -1)
    expect(lib.ttm_basis(p, tm.D, 0, X, ldx, N, tm._ptr(Z), Z.shape[1], st), -1)                                     # component out of range
    with pytest.raises(_capi.TTMError, match='libttm error -1'):
        _capi.check(lib.ttm_forward(p, c, f, None, ldx, N, 0, tm.D, tm._ptr(Z), Z.shape[1], None, None, None, st))
    # ... and a valid call still works afterwards
    _capi.check(lib.ttm_forward(p, c, f, X, ldx, N, 0, tm.D, tm._ptr(Z), Z.shape[1], None, None, None, st))


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['c5_sep', 'c3_sep', 'c2b_sep', 'c2a_int'])
def test_gram_on_the_matrix_cores_equals_the_pairwise_kernel(name, ttm_opt):
    """G = Psi' Psi (TM:2966-2975) by v_mfma_f64_16x16x4f64 - one tile for up to 16 basis functions, three for up to 32 -
    against the pairwise FMA kernel and against NumPy on the basis matrices: the same sums up to their order."""
    from triangular_transport_toolbox_amd.transport_map import transport_map
    npz, desc = load_case(name)
    X = case_X(name, npz)
    X = np.concatenate([X + 0.01 * i for i in range(5)])[:4099]          # several tiles and a ragged last one
    tm = transport_map(X=X, monotone=desc['monotone'], nonmonotone=desc['nonmonotone'], verbose=False, **ctor_kwargs(desc))
    seen = set()
    for k in range(tm.D):
        ttm_opt('gram_mfma', 0)
        G0 = tm._gram(k)
        assert tm._lib.ttm_last_kernel().decode() == 'k_gram'
        ttm_opt('gram_mfma', 1)
        G1 = tm._gram(k)
        m = G0.shape[0]
        if m <= 32:
            assert tm._lib.ttm_last_kernel().decode() == 'k_gram_mfma'
            seen.add(m > 16)
        assert np.array_equal(G1, G1.T)
        assert np.max(np.abs(G1 - G0)) <= 1e-14 * np.max(np.abs(G0))             # (entries that cancel to ~0 differ by rounding only)
    assert seen                                                          # (the matrix-core kernel ran)
