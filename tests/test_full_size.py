"""BASELINE.json's configurations at their FULL sizes on the GPU (all `-m gpu`): the oracle on a subset of >= 10^4
samples that includes the tails of every column, plus size-independent properties on the whole ensemble
(permutation equivariance, round trips, the fused reductions).  C5 (d = 40, N = 1e6), C2b / C2a (spiral, N = 1e6),
C4 (Lorenz-63 filter update, N = 1e5)."""
import ctypes

import numpy as np
import pytest

from tests.util import coeff_lists, load_case, record_parity, relerr

pytestmark = pytest.mark.gpu


def subset_with_tails(X, n_random=10000, n_tail=16, seed=3):
    """n_random random rows + for every column the n_tail smallest and largest entries."""
    rng = np.random.default_rng(seed)
    idx = set(rng.choice(len(X), size=n_random, replace=False).tolist())
    for j in range(X.shape[1]):
        order = np.argsort(X[:, j])
        idx.update(order[:n_tail].tolist())
        idx.update(order[-n_tail:].tolist())
    return np.array(sorted(idx))


def build(cfgname, fixture, N):
    from triangular_transport_toolbox_amd import specs
    from triangular_transport_toolbox_amd.transport_map import transport_map
    from oracle.ttm_oracle import OracleMap
    cfg = specs.config(cfgname)
    X = cfg['sampler'](N)
    npz, desc = load_case(fixture)
    tm = transport_map(X=X, monotone=cfg['monotone'], nonmonotone=cfg['nonmonotone'], verbose=False, **cfg['kwargs'])
    tm.coeffs_mon, tm.coeffs_nonmon = coeff_lists(npz, tm.D)
    om = OracleMap(X=X, monotone=cfg['monotone'], nonmonotone=cfg['nonmonotone'], **cfg['kwargs'])
    om.coeffs_mon, om.coeffs_nonmon = coeff_lists(npz, tm.D)
    return tm, om, X


def _last_kernel(tm):
    tm._lib.ttm_last_kernel.restype = ctypes.c_char_p
    return tm._lib.ttm_last_kernel().decode()


def test_c5_ring_inverse_with_several_tiles_per_workgroup(ttm_opt):
    """C5 at N = 2 100 003 (an odd tail; 8 204 rows per workgroup = three tiles, the first two full): the ring of resident-table
    images keeps turning across the tiles of a chunk.  Against k_band_inverse (which re-reads two columns at its block boundary when
    a chunk has several tiles: last bits) and the oracle on a subset with tails."""
    import torch
    N = 2100003
    tm, om, X = build('C5', 'c5_sep', N)
    Z = tm.forward_device(tm._Xs, tm._N)
    res = {}
    for ring in (1, 0):
        ttm_opt('band_ring', ring)
        Xi = tm.inverse_device(Z, tm._N)
        torch.cuda.synchronize()
        res[ring] = Xi[:, :N].clone()
        assert _last_kernel(tm) == ('k_band_inverse_ring' if ring else 'k_band_inverse')
    assert (res[1] - res[0]).abs().max().item() < 1e-12
    assert (res[1] - tm._Xs[:, :N]).abs().max().item() < 1e-3               # (round trip: the tables' interpolation error)
    idx = subset_with_tails(X, 3000)
    sel = torch.from_numpy(idx).to(Z.device)
    Xo = (om.inverse_map(Z[:, :N].T[sel].cpu().numpy()) - om.X_mean) / om.X_std
    err = relerr(res[1].T[sel].cpu().numpy(), Xo)
    record_parity('c5_2.1e6/table_inverse(k_band_inverse_ring, three tiles per workgroup)_vs_oracle', err, 1e-11)
    assert err < 1e-11


def test_c5_full_size_against_the_oracle_on_1e4_samples_with_tails(ttm_opt):
    """C5 (d = 40, band 2, order 3, N = 1e6) through the kernels the benchmark times - asserted by name on the
    device-resident entry points - against the oracle: map 1e-11, table inverse 1e-11."""
    N = 1000000
    tm, om, X = build('C5', 'c5_sep', N)
    assert tm._cm.u_enabled and tm._cm.u_h_cls > 0 and tm._cm.u_p_lag == 2
    idx = subset_with_tails(X)
    assert len(idx) >= 10000
    assert relerr(tm.X_mean, om.X_mean) < 1e-12 and relerr(tm.X_std, om.X_std) < 1e-12
    Zdev = tm.forward_device(tm._Xs, tm._N)
    assert _last_kernel(tm) == 'k_band_forward'
    tm.inverse_device(Zdev, tm._N)
    assert _last_kernel(tm) == 'k_band_inverse_ring'
    Z = tm.map(X)
    Zo = om.map(X[idx])
    record_parity('c5_full/map(k_band_forward)_vs_oracle', relerr(Z[idx], Zo), 1e-11)
    assert relerr(Z[idx], Zo) < 1e-11
    Xi = tm.inverse_map(Z)
    Xio = om.inverse_map(Z[idx])
    record_parity('c5_full/table_inverse(k_band_inverse_ring)_of_pushed_samples_vs_oracle', relerr(Xi[idx], Xio), 1e-11)
    assert relerr(Xi[idx], Xio) < 1e-11
    # reference samples (not pushed forward ones): standard normal z, including |z| > 4
    Zr = np.random.default_rng(1).standard_normal((N, tm.D))
    Zr[:50] *= 2.5
    jdx = np.concatenate((np.arange(50), subset_with_tails(Zr, 10000)))
    Xr = tm.inverse_map(Zr)
    Xro = om.inverse_map(Zr[jdx])
    record_parity('c5_full/table_inverse(k_band_inverse_ring)_of_reference_samples_vs_oracle', relerr(Xr[jdx], Xro), 1e-11)
    assert relerr(Xr[jdx], Xro) < 1e-11
    ttm_opt('rt_window', 0)                                                   # whole tables resident: the same bits as the planned window
    assert np.array_equal(tm.inverse_map(Zr), Xr)
    ttm_opt('rt_window', -1)
    perm = np.random.default_rng(3).permutation(N)
    assert np.array_equal(tm.inverse_map(Zr[perm]), Xr[perm])                 # samples are independent: exact
    assert np.array_equal(tm.map(X[perm]), Z[perm])
    assert np.max(np.abs(Xi - X) / tm.X_std) < 5e-4                           # round trip at the table's resolution
    ld_only, ld_fused = tm._empty(tm._N), tm._empty(tm._N)
    tm.density_device(tm._Xs, tm._N, logdet=ld_only)
    assert _last_kernel(tm) == 'k_band_logdet'
    tm.density_device(tm._Xs, tm._N, logdet=ld_fused, sumsq=tm._empty(tm._N))
    assert _last_kernel(tm) == 'k_band_density'
    record_parity('c5_full/logdet(k_band_logdet)_vs_fused_density_pass', relerr(ld_only[:N].cpu().numpy(), ld_fused[:N].cpu().numpy()), 1e-13)
    assert relerr(ld_only[:N].cpu().numpy(), ld_fused[:N].cpu().numpy()) < 1e-13
    pd, pdo = tm.evaluate_pullback_density(X[idx]), om.evaluate_pullback_density(X[idx])
    ok = pdo > 0                                                              # (40 dimensions: the density itself underflows towards the tails -
    record_parity('c5_full/log_pullback_density(k_band_density+k_band_logdet)_vs_oracle', relerr(np.log(pd[ok]), np.log(pdo[ok])), 1e-10)    # compare its logarithm)
    assert relerr(pd, pdo) < 1e-10 and np.array_equal(pd > 0, ok) and relerr(np.log(pd[ok]), np.log(pdo[ok])) < 1e-10
    # the kernels the band kernels replaced stay selectable and agree with them
    ttm_opt('band_fwd', 0); ttm_opt('band_inv', 0)
    Zh = tm.forward_device(tm._Xs, tm._N)
    assert _last_kernel(tm) == 'k_forward_hl'
    Xh = tm.inverse_device(Zh, tm._N)
    assert _last_kernel(tm) == 'k_inverse_rt<band>'
    assert relerr(Zh[:, :N].cpu().numpy(), Zdev[:, :N].cpu().numpy()) < 1e-13
    record_parity('c5_full/k_inverse_rt_vs_oracle', relerr(tm.inverse_map(Zr)[jdx], Xro), 1e-11)
    assert relerr(tm.inverse_map(Zr)[jdx], Xro) < 1e-11


def test_c3_full_size_map_inverse_pullback_optimize():
    """C3 (d = 4 dense, order 4, N = 5e5): map, table inverse, pullback against the oracle on >= 1e4 samples with tails;
    optimize() from the initial coefficients ends at or below the reference's objective (fixture c3_sep, N = 2000)."""
    N = 500000
    tm, om, X = build('C3', 'c3_sep', N)
    idx = subset_with_tails(X)
    Z = tm.map(X)
    tm.forward_device(tm._Xs, tm._N)
    assert _last_kernel(tm) == 'k_band_few'                  # (a group three columns back: lag-3 push records)
    record_parity('c3_full/map(k_band_few)_vs_oracle', relerr(Z[idx], om.map(X[idx])), 1e-11)
    assert relerr(Z[idx], om.map(X[idx])) < 1e-11
    Xi = tm.inverse_map(Z)
    tm.inverse_device(tm._cols(tm.D, tm._N, zero=True), tm._N)
    assert _last_kernel(tm) == 'k_band_few_inverse'
    ld_only = tm._empty(tm._N)
    tm.density_device(tm._Xs, tm._N, logdet=ld_only)
    assert _last_kernel(tm) == 'k_band_logdet'               # (the derivative of a separable component reads its own column only)
    ld_fused = tm._empty(tm._N)
    tm.density_device(tm._Xs, tm._N, logdet=ld_fused, sumsq=tm._empty(tm._N))
    assert _last_kernel(tm) == 'k_band_few<density>'
    record_parity('c3_full/logdet(k_band_logdet)_vs_fused_density_pass', relerr(ld_only[:N].cpu().numpy(), ld_fused[:N].cpu().numpy()), 1e-13)
    assert relerr(ld_only[:N].cpu().numpy(), ld_fused[:N].cpu().numpy()) < 1e-13
    record_parity('c3_full/table_inverse(k_band_few_inverse)_vs_oracle', relerr(Xi[idx], om.inverse_map(Z[idx])), 1e-11)
    assert relerr(Xi[idx], om.inverse_map(Z[idx])) < 1e-11
    pd, pdo = tm.evaluate_pullback_density(X[idx]), om.evaluate_pullback_density(X[idx])
    record_parity('c3_full/pullback_density_vs_oracle', relerr(pd, pdo), 1e-10)
    assert relerr(pd, pdo) < 1e-10
    perm = np.random.default_rng(7).permutation(N)
    assert np.array_equal(tm.map(X[perm]), Z[perm])
    # optimize() on the full ensemble: the objective at the optimum found is at or below the objective of the
    # reference-optimised coefficients of the fixture evaluated on the SAME ensemble (KL objective, per component)
    ref_mon, ref_non = [c.copy() for c in tm.coeffs_mon], [c.copy() for c in tm.coeffs_nonmon]
    for k in range(tm.D):
        tm.coeffs_mon[k] = tm.coeffs_mon[k] * 0 + tm.coeffs_init
        tm.coeffs_nonmon[k] = tm.coeffs_nonmon[k] * 0 + tm.coeffs_init
    tm.optimize()

    def objective(mon, non):
        # sample KL objective of the whole map, sum_k mean(S_k^2 / 2 - log dS_k/dx_k), from the fused density pass
        tm.coeffs_mon, tm.coeffs_nonmon = [c.copy() for c in mon], [c.copy() for c in non]
        ld, ss = tm._empty(tm._N), tm._empty(tm._N)
        tm.forward_device(tm._Xs, tm._N, logdet=ld, sumsq=ss)
        return float((0.5 * ss[:tm._N] - ld[:tm._N]).mean().item())
    opt_mon, opt_non = [c.copy() for c in tm.coeffs_mon], [c.copy() for c in tm.coeffs_nonmon]
    J_opt = objective(opt_mon, opt_non)
    J_ref = objective(ref_mon, ref_non)
    tm.coeffs_mon, tm.coeffs_nonmon = opt_mon, opt_non
    record_parity('c3_full/optimize_J_minus_J_of_reference_coefficients', J_opt - J_ref, 1e-8 * abs(J_ref))
    assert J_opt <= J_ref + 1e-8 * abs(J_ref)
    Zopt = tm.map(X[:20000])
    assert abs(Zopt.mean()) < 0.02 and abs(Zopt.std() - 1.0) < 0.02


def test_c2b_full_size_forward_inverse_pullback():
    N = 1000000
    tm, om, X = build('C2b', 'c2b_sep', N)
    idx = subset_with_tails(X)
    Z = tm.map(X)
    tm.forward_device(tm._Xs, tm._N)
    assert _last_kernel(tm) == 'k_band_few'
    record_parity('c2b_full/map(k_band_few)_vs_oracle', relerr(Z[idx], om.map(X[idx])), 1e-11)
    assert relerr(Z[idx], om.map(X[idx])) < 1e-11
    Xi = tm.inverse_map(Z)
    tm.inverse_device(tm._cols(tm.D, tm._N, zero=True), tm._N)
    assert _last_kernel(tm) == 'k_band_few_inverse'
    record_parity('c2b_full/table_inverse(k_band_few_inverse)_vs_oracle', relerr(Xi[idx], om.inverse_map(Z[idx])), 1e-11)
    assert relerr(Xi[idx], om.inverse_map(Z[idx])) < 1e-11
    pd, pdo = tm.evaluate_pullback_density(X[idx]), om.evaluate_pullback_density(X[idx])
    record_parity('c2b_full/pullback_density_vs_oracle', relerr(pd, pdo), 1e-10)
    assert relerr(pd, pdo) < 1e-10
    perm = np.random.default_rng(5).permutation(N)
    assert np.array_equal(tm.map(X[perm]), Z[perm])
    # bisection semantics of the same map (alternate_root_finding = False), oracle on a smaller subset
    tm.alternate_root_finding = False
    om.alternate_root_finding = False
    sub = idx[::5]
    Xb = tm.inverse_map(Z[sub])
    Xo = om.inverse_map(Z[sub])
    assert np.max(np.abs(Xb[1:] - Xo[1:])) < 1e-6                           # (row 0: the reference's loop-guard quirk depends on the batch)


def test_c2a_full_size_integrated_forward_and_bisection_inverse():
    N = 1000000
    tm, om, X = build('C2a', 'c2a_int', N)
    idx = subset_with_tails(X, n_random=10000)
    Z = tm.map(X)
    assert relerr(Z[idx], om.map(X[idx])) < 1e-10
    Xi = tm.inverse_map(Z)                                                  # 1e6 bisection searches, reference midpoint sequence
    # S(S^-1(z)) = z to the root search's own threshold (|S - z| <= 1e-9 at the last midpoint), under the ORACLE's map
    sub = idx[::4]
    res = om.map(Xi[sub]) - Z[sub]
    assert np.max(np.abs(res[1:])) < 5e-9
    Xo = om.inverse_map(Z[sub])
    assert np.max(np.abs(Xi[sub][1:] - Xo[1:])) < 1e-6
    assert np.max(np.abs(Xi[1:] - X[1:])) < 1e-6                            # round trip of the whole ensemble


def test_c4_filter_update_at_1e5_against_the_oracle():
    """One assimilation update of the Example-06 filter at N = 1e5 (reset -> optimize -> map -> conditional inverse)
    on the device-resident filter against the oracle's update of the same ensemble with the same noise."""
    from triangular_transport_toolbox_amd import entf, specs
    from oracle.ttm_oracle import OracleMap
    N = 100000
    rng = np.random.default_rng(0)
    ens = rng.standard_normal((N, 3)) * [8, 9, 8] + [0, 0, 25]
    ens = entf.rk4(ens, 0.05, 20)                                           # onto the attractor
    noise = 2.0 * rng.standard_normal((3, N))
    obs = ens.mean(axis=0) + np.array([1.0, -2.0, 0.5])
    flt = entf.Filter(N)
    flt.set_ensemble(ens)
    flt.assimilate(obs, noises=noise)
    got = flt.ensemble()
    # oracle: the same three updates with SciPy on the host
    mon, non = specs.entf_filter_spec(3)
    Xa = ens.copy()
    om = None
    for idx, perm in enumerate(entf.PERMUTATIONS):
        Yt = Xa[:, idx] + noise[idx]
        inp = np.column_stack((Yt[:, None], Xa[:, perm]))
        if om is None:
            om = OracleMap(X=inp, monotone=mon, nonmonotone=non, polynomial_type='hermite function',
                           monotonicity='separable monotonicity', regularization='l2', regularization_lambda=0.05)
        else:
            om.reset(inp)
        om.optimize()
        Zp = om.map(inp)
        ret = om.inverse_map(Zp, X_star=np.full((N, 1), obs[idx]))
        Xa = ret[:, perm]
    assert relerr(got, Xa) < 1e-5
    sub = subset_with_tails(ens, 10000)
    assert relerr(got[sub], Xa[sub]) < 1e-5
