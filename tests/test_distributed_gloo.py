"""
N > 1 path on CPU: world_size-2 `gloo` process groups driving the class through the host test double.

* sample sharding (`shard_samples=True`): each rank owns half of the ensemble; column moments, Gram
  matrices, objective/gradient sums and the bisection iteration caps are all-reduced; results must equal
  the single-process ones (<= 1e-13 rel: only the reduction order differs);
* component sharding (`shard_components=True`): each rank optimises a subset of components on its replica,
  coefficients are exchanged and the summed objective is all-reduced.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, outdir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from tests.hostemu import emu
        from tests.util import case_X, coeff_lists, ctor_kwargs, load_case
        from triangular_transport_toolbox_amd.transport_map import transport_map
        npz, desc = load_case(case)
        X = case_X(case, npz)
        kw = ctor_kwargs(desc)
        out = {}
        with emu.install():
            if case.startswith('shardcomp:'):
                pass
            N = X.shape[0]
            lo, hi = (0, N // 2 + 3) if rank == 0 else (N // 2 + 3, N)       # uneven shards
            tm = transport_map(X=X[lo:hi], monotone=desc['monotone'], nonmonotone=desc['nonmonotone'], verbose=False,
                               shard_samples=True, **kw)
            out['X_mean'], out['X_std'] = tm.X_mean, tm.X_std
            out['centers'] = np.concatenate([np.ravel(v['centers']) for d in tm.special_terms.values()
                                             for key, v in d.items() if key != 'cross-terms'] + [np.zeros(0)])
            tm.coeffs_mon, tm.coeffs_nonmon = coeff_lists(npz, tm.D)
            out['Z'] = tm.map(X[lo:hi])
            if kw['monotonicity'] == 'integrated rectifier':
                k = tm.D - 1
                div = len(tm.coeffs_nonmon[k])
                c = npz['obj_c_%d' % k][2]
                out['J'] = tm.objective_function(c.copy(), k, div)
                out['G'] = tm.objective_function_jacobian(c.copy(), k, div)
                tm.alternate_root_finding = False
            else:
                k = tm.D - 1
                A, solve = tm.separable_setup(k)
                out['A'] = A
                out['Jsep'], out['Gsep'] = tm.separable_objective(npz['sep_c_%d' % k][0].copy(), A, k)
            Zin = npz['inv_Z']
            M = Zin.shape[0]
            zlo, zhi = (0, M // 2) if rank == 0 else (M // 2, M)
            tm.alternate_root_finding = False                   # bisection: needs the all-reduced iteration caps
            out['Xinv'] = tm.inverse_map(Zin[zlo:zhi])
            # exact order statistics of a column sharded over the ranks (duplicates, both signs, uneven shards):
            # the radix select with all-reduced bin counts, no gather
            g = np.random.default_rng(11).standard_normal(1501).round(2)
            g[::7] = -g[::7]
            mine = g[:600] if rank == 0 else g[600:]
            before = emu.lib().ttm_hostemu_allreduce_calls()
            ranks_ = np.array([0, 1, 375, 750, 751, 1499, 1500], dtype=np.int64)
            vals, n_tot = tm._order_statistics(torch.from_numpy(mine.copy()), ranks_)
            out['os'], out['os_n'] = vals, n_tot
            out['os_ref'] = np.sort(g)[ranks_]
            out['os_allreduces'] = emu.lib().ttm_hostemu_allreduce_calls() - before
            out['allreduce_calls'] = emu.lib().ttm_hostemu_allreduce_calls()
        np.savez(os.path.join(outdir, 'rank%d.npz' % rank), **out)
    finally:
        dist.destroy_process_group()


def _worker_components(rank, world, port, case, outdir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from tests.hostemu import emu
        from tests.util import case_X, ctor_kwargs, load_case
        from triangular_transport_toolbox_amd.transport_map import transport_map
        npz, desc = load_case(case)
        X = case_X(case, npz)
        with emu.install():
            tm = transport_map(X=X, monotone=desc['monotone'], nonmonotone=desc['nonmonotone'], verbose=False,
                               shard_components=True, **ctor_kwargs(desc))
            tm.optimize()
            out = {'J': tm.objective_total, 'allreduce_calls': emu.lib().ttm_hostemu_allreduce_calls()}
            for k in range(tm.D):
                out['mon_%d' % k] = tm.coeffs_mon[k]
                out['non_%d' % k] = tm.coeffs_nonmon[k]
        np.savez(os.path.join(outdir, 'rank%d.npz' % rank), **out)
    finally:
        dist.destroy_process_group()


def _single(case):
    from tests.hostemu import emu
    from tests.util import case_X, coeff_lists, ctor_kwargs, load_case
    from triangular_transport_toolbox_amd.transport_map import transport_map
    npz, desc = load_case(case)
    X = case_X(case, npz)
    kw = ctor_kwargs(desc)
    with emu.install():
        tm = transport_map(X=X, monotone=desc['monotone'], nonmonotone=desc['nonmonotone'], verbose=False, **kw)
    return tm, npz, X, kw


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b) / (np.abs(b) + 1.0))) if a.size else 0.0


@pytest.mark.parametrize('case', ['c1_int', 'c3_sep'])
def test_sample_sharding_world2(case, tmp_path):
    from tests.hostemu import emu
    from tests.util import coeff_lists
    port = _free_port()
    mp.spawn(_worker, args=(2, port, case, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / 'rank0.npz'), np.load(tmp_path / 'rank1.npz')
    tm, npz, X, kw = _single(case)
    # the data-path reductions went through the C ABI's ttm_allreduce_* (moments, sums / Gram matrix, iteration caps)
    assert int(r0['allreduce_calls']) >= 4 and int(r0['allreduce_calls']) == int(r1['allreduce_calls'])
    assert np.array_equal(r0['os'], r0['os_ref']) and np.array_equal(r1['os'], r1['os_ref']) and int(r0['os_n']) == 1501
    assert int(r0['os_allreduces']) == 8                   # one all-reduce of the bin counts per radix pass
    with emu.install():
        assert rel(r0['X_mean'], tm.X_mean) < 1e-13 and rel(r0['X_std'], tm.X_std) < 1e-13
        assert np.array_equal(r0['X_mean'], r1['X_mean'])
        centers = np.concatenate([np.ravel(v['centers']) for d in tm.special_terms.values()
                                  for key, v in d.items() if key != 'cross-terms'] + [np.zeros(0)])
        assert rel(r0['centers'], centers) < 1e-12
        tm.coeffs_mon, tm.coeffs_nonmon = coeff_lists(npz, tm.D)
        Z = tm.map(X)
        assert rel(np.vstack((r0['Z'], r1['Z'])), Z) < 1e-12
        k = tm.D - 1
        if kw['monotonicity'] == 'integrated rectifier':
            div = len(tm.coeffs_nonmon[k])
            c = npz['obj_c_%d' % k][2]
            assert abs(r0['J'] - tm.objective_function(c.copy(), k, div)) < 1e-13 * (1 + abs(r0['J']))
            assert rel(r0['G'], tm.objective_function_jacobian(c.copy(), k, div)) < 1e-13
            assert np.array_equal(r0['G'], r1['G'])
        else:
            A, _ = tm.separable_setup(k)
            assert rel(r0['A'], A) < 1e-12
            J, G = tm.separable_objective(npz['sep_c_%d' % k][0].copy(), A, k)
            assert abs(r0['Jsep'] - J) < 1e-12 * (1 + abs(J)) and rel(r0['Gsep'], G) < 1e-12
        tm.alternate_root_finding = False
        Xinv = tm.inverse_map(npz['inv_Z'])
        got = np.vstack((r0['Xinv'], r1['Xinv']))
        # sample 0 is capped by the maximum iteration count over BOTH ranks (all-reduce max)
        assert rel(got, Xinv) < 1e-12


def test_component_sharding_world2(tmp_path):
    from tests.hostemu import emu
    case = 'c3_sep'
    port = _free_port()
    mp.spawn(_worker_components, args=(2, port, case, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / 'rank0.npz'), np.load(tmp_path / 'rank1.npz')
    tm, npz, X, kw = _single(case)
    assert int(r0['allreduce_calls']) == 2           # coefficient exchange + the one scalar objective all-reduce
    with emu.install():
        tm.optimize()
        for k in range(tm.D):
            assert np.array_equal(r0['mon_%d' % k], r1['mon_%d' % k])          # every rank ends with all coefficients
            assert rel(r0['mon_%d' % k], tm.coeffs_mon[k]) < 1e-12
            assert rel(r0['non_%d' % k], tm.coeffs_nonmon[k]) < 1e-12
        assert abs(float(r0['J']) - tm.objective_total) < 1e-12 * (1 + abs(tm.objective_total))
        assert float(r0['J']) == float(r1['J'])


def _worker_entf(rank, world, port, outdir):
    """One assimilation cycle of the device-resident filter with the ensemble SAMPLE-SHARDED over the ranks."""
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from tests.hostemu import emu
        from triangular_transport_toolbox_amd import entf
        ens, noise, obs = _entf_case()
        N = len(ens)
        lo, hi = (0, N // 2 + 7) if rank == 0 else (N // 2 + 7, N)            # uneven shards
        with emu.install():
            before = emu.lib().ttm_hostemu_allreduce_calls()
            flt = entf.Filter(hi - lo, seed=3, row0=lo, shard_samples=True)
            flt.set_ensemble(ens[lo:hi])
            flt.assimilate(obs, noises=noise[:, lo:hi])
            a = flt.ensemble()
            flt.forecast(0.05, 2)
            flt.assimilate(obs + 0.5)                                       # generator noise: a function of the GLOBAL row
            b = flt.ensemble()
            calls = emu.lib().ttm_hostemu_allreduce_calls() - before
        np.savez(os.path.join(outdir, 'entf%d.npz' % rank), a=a, b=b, calls=calls)
    finally:
        dist.destroy_process_group()


def _entf_case(N=1201):
    from triangular_transport_toolbox_amd import entf
    rng = np.random.default_rng(5)
    ens = entf.rk4(rng.standard_normal((N, 3)) * [8, 9, 8] + [0, 0, 25], 0.05, 20)
    return ens, 2.0 * rng.standard_normal((3, N)), ens.mean(axis=0) + np.array([1.0, -2.0, 0.5])


def test_sample_sharded_entf_update_equals_the_single_rank_one(tmp_path):
    """BASELINE configs[3] with the ensemble sharded over two ranks: column moments, order statistics
    and every optimiser evaluation's fused objective + gradient sums go through ttm_allreduce_*; the updated ensemble
    equals the single-rank filter's (only the order of the partial sums differs)."""
    from tests.hostemu import emu
    from triangular_transport_toolbox_amd import entf
    port = _free_port()
    mp.spawn(_worker_entf, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / 'entf0.npz'), np.load(tmp_path / 'entf1.npz')
    ens, noise, obs = _entf_case()
    with emu.install():
        flt = entf.Filter(len(ens), seed=3)
        flt.set_ensemble(ens)
        flt.assimilate(obs, noises=noise)
        a = flt.ensemble()
        flt.forecast(0.05, 2)
        flt.assimilate(obs + 0.5)
        b = flt.ensemble()
    assert int(r0['calls']) == int(r1['calls']) and int(r0['calls']) > 6 * 10       # per update: moments, selects, one per evaluation
    # Every reduction of the update agrees with the single-rank one to rounding (1e-13, test_sample_sharding_world2); what
    # the ensembles differ by is what ~35 L-BFGS-B iterations per component make of that rounding: 5e-10 measured after
    # one cycle (the optimiser itself stops at a projected gradient of 1e-5, three orders above it)
    assert rel(np.vstack((r0['a'], r1['a'])), a) < 1e-8
    assert rel(np.vstack((r0['b'], r1['b'])), b) < 1e-7
