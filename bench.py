#!/usr/bin/env python3
"""
bench.py - headline benchmark of the MI355X transport-map engine.

Metric (BASELINE.json): map-evals/s, forward + inverse, N samples x D components,
with the ensemble resident in HBM (column-major, the engine's native layout).
One "step" = one forward map of the whole ensemble followed by one inverse map
of the result (table inverse for separable maps, reference bisection semantics
for integrated-rectifier maps).  value = N * D * steps / time, summed over ranks
(weak scaling: every rank owns its own N-sample ensemble; the path has no
data-path collective).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C5|C2a|C2b|C3]

prints ONE JSON line (rank 0) with the `roofline` object of the dominant kernel
(forward map: algorithmic bytes 8 N (d_used + D)) and a
`cpu_baseline` object (the CPU oracle, one process per host core on a bounded sample each, rank 0, N=1).
The headline workload is C5 (BASELINE.json configs[4], the one north_star's target is quoted on; it fits one GPU);
at N=1 the same line carries, under `other_configs`, the secondary numbers of the other single-GPU configurations
(configs[1] = C2b / C2a spiral d=2 order 5 N=1e6 forward + inverse + pullback, configs[2] = C3 d=4 order 4 N=5e5 with
optimize()).  Timing discipline (pre-warm, no pause before the timed steps): DESIGN.md section 6, "Clock state".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # name: (config, N, golden coefficient fixture, description)
    'C5': ('C5', 1000000, 'c5_sep', 'd=40 banded separable map, order 3, N=1e6 (Gaussian-mixture target)'),
    'C3': ('C3', 500000, 'c3_sep', 'd=4 dense separable map, order 4, N=5e5 (banana target)'),
    'C2b': ('C2b', 1000000, 'c2b_sep', 'spiral d=2 separable map, order 5, N=1e6'),
    'C2a': ('C2a', 1000000, 'c2a_int', 'spiral d=2 integrated-rectifier map, order 5, Q=25, N=1e6'),
}


def load_coeffs(fixture, D):
    npz = np.load(os.path.join(ROOT, 'tests', 'golden', fixture + '.npz'))
    return ([npz['coeffs_mon_%d' % k] for k in range(D)], [npz['coeffs_nonmon_%d' % k] for k in range(D)])


def build_map(workload, rank, n_override=None, **extra_kwargs):
    from triangular_transport_toolbox_amd import specs
    from triangular_transport_toolbox_amd.transport_map import transport_map
    cfgname, N, fixture, _ = WORKLOADS[workload]
    if n_override:
        N = n_override
    cfg = specs.config(cfgname)
    seed = {'C5': 12345, 'C3': 0, 'C2b': 0, 'C2a': 0}[workload] + 1000 * rank
    X = cfg['sampler'](N, seed=seed)
    tm = transport_map(X=X, monotone=cfg['monotone'], nonmonotone=cfg['nonmonotone'], verbose=False, **cfg['kwargs'],
                       **extra_kwargs)
    tm.coeffs_mon, tm.coeffs_nonmon = load_coeffs(fixture, tm.D)
    return tm, X, cfg


def d_used(tm):
    """number of distinct sample columns the forward map reads"""
    cols = set()
    for k in range(tm.D):
        cols.add(k + tm.skip_dimensions)
        for entry in list(tm.monotone[k]) + list(tm.nonmonotone[k]):
            if isinstance(entry, str):
                cols.add(int(entry.split(' ')[1]))
            else:
                cols.update(int(e) for e in entry if not isinstance(e, str))
    return len(cols)


def _cpu_worker(args):
    """One host core: the CPU oracle on its own n-sample ensemble of the workload (forward + inverse)."""
    workload, n, seed = args
    for v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
        os.environ[v] = '1'
    from oracle.ttm_oracle import OracleMap      # timed baseline only
    from triangular_transport_toolbox_amd import specs
    cfgname, _, fixture, _ = WORKLOADS[workload]
    cfg = specs.config(cfgname)
    Xc = cfg['sampler'](n, seed=seed)
    om = OracleMap(X=Xc, monotone=cfg['monotone'], nonmonotone=cfg['nonmonotone'], **cfg['kwargs'])
    om.coeffs_mon, om.coeffs_nonmon = load_coeffs(fixture, om.D)
    t0 = time.perf_counter()
    Z = om.map(Xc)
    t1 = time.perf_counter()
    om.inverse_map(Z)
    t2 = time.perf_counter()
    return n * om.D, t1 - t0, t2 - t1


def cpu_baseline(workload, n_cpu, cores):
    """The CPU oracle (NumPy restatement of the reference CPU path), one process per host core, every process on
    its own n_cpu-sample ensemble - the sample-parallel use of the host the reference's `workers` pool
    (TM:2789-2874) aims at.  Must run BEFORE this process initialises the GPU (the pool forks)."""
    import multiprocessing as mp
    seeds = [7000 + 13 * i for i in range(cores)]
    t0 = time.perf_counter()
    if cores > 1:
        with mp.get_context('fork').Pool(cores) as pool:
            res = pool.map(_cpu_worker, [(workload, n_cpu, sd) for sd in seeds])
    else:
        res = [_cpu_worker((workload, n_cpu, seeds[0]))]
    wall = time.perf_counter() - t0
    evals = sum(r[0] for r in res)
    busy = max(r[1] + r[2] for r in res)
    return dict(value=evals / busy, unit='map-evals/s', cores=cores, kind='port',
                sample='oracle/ttm_oracle.py (NumPy restatement of the reference CPU path), %d processes x %d samples of '
                       'the workload each, forward+inverse; slowest process %.2f s (forward %.2f s, inverse %.2f s on '
                       'average), %.1f s wall including start-up; one process alone: %.3g map-evals/s'
                       % (cores, n_cpu, busy, float(np.mean([r[1] for r in res])), float(np.mean([r[2] for r in res])), wall,
                          float(np.mean([r[0] / (r[1] + r[2]) for r in res]))),
                host_cores_available=os.cpu_count())


def other_configs(torch, names, steps=40):
    """The other single-GPU configurations of BASELINE.json (secondary numbers of the same JSON line): per workload
    forward / inverse launch times with HIP events (back to back behind a short untimed run), map-evals/s of forward +
    inverse over `steps` steps, the fused pullback pass for separable maps, optimize() wall-clock from coeffs_init."""
    out = {}
    for name in names:
        tm, X, cfg = build_map(name, 0)
        N, D, d = tm._N, tm.D, tm._cm.d_cols
        separable = tm.monotonicity == 'separable monotonicity'
        coef = tm._pack_coeffs()
        Xs, Z, Xinv = tm._Xs, tm._cols(D, N), tm._cols(d, N, zero=True)

        def step():
            tm.forward_device(Xs, N, coef=coef, Z=Z)
            tm.inverse_device(Z, N, coef=coef, X=Xinv)
        n_warm = 200 if separable else 5            # (a bisection inverse of 1e6 samples is ~17 ms)
        n = steps if separable else max(3, steps // 8)
        for _ in range(n_warm):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)]
        for a, b, c in ev:
            a.record(); tm.forward_device(Xs, N, coef=coef, Z=Z); b.record(); tm.inverse_device(Z, N, coef=coef, X=Xinv); c.record()
        torch.cuda.synchronize()
        r = dict(workload=WORKLOADS[name][3], N=N, D=D, steps=n, ms_per_step=1e3 * el / n, value=N * D * n / el,
                 forward_ms=float(np.mean([a.elapsed_time(b) for a, b, c in ev])),
                 inverse_ms=float(np.mean([b.elapsed_time(c) for a, b, c in ev])),
                 inverse='table' if separable else 'bisection (reference sequence)',
                 roundtrip_max_abs_err=float((Xinv[:, :N] - Xs[:, :N]).abs().max().item()),
                 # (the max sits in the tails, where the reference's 1001-point table inverse clips / interpolates
                 # coarsely or the bisection window runs away; the median is the inverse's working accuracy)
                 roundtrip_median_abs_err=float((Xinv[:, :N] - Xs[:, :N]).abs().median().item()))
        r['forward_GBps_algorithmic'] = 8.0 * N * (d_used(tm) + D) / (r['forward_ms'] * 1e-3) / 1e9
        if separable:
            ld, ss = tm._empty(N), tm._empty(N)
            sigma = tm._to_dev(np.asarray(tm.X_std[:D], dtype=float))
            for _ in range(100):
                tm.forward_device(Xs, N, coef=coef, Z=Z, logdet=ld, sigma=sigma, sumsq=ss)
            evp = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
            for a, b in evp:
                a.record(); tm.forward_device(Xs, N, coef=coef, Z=Z, logdet=ld, sigma=sigma, sumsq=ss); b.record()
            torch.cuda.synchronize()
            r['pullback_fused_ms'] = float(np.mean([a.elapsed_time(b) for a, b in evp]))
        if not separable:
            # the same inversion with the engine's safeguarded Newton root search (root_finder='newton', an extension:
            # same roots to |S - z| <= 1e-9, not the reference's midpoint sequence)
            tm.root_finder = 'newton'
            for _ in range(3):
                tm.inverse_device(Z, N, coef=coef, X=Xinv)
            evn = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
            for a, b in evn:
                a.record(); tm.inverse_device(Z, N, coef=coef, X=Xinv); b.record()
            torch.cuda.synchronize()
            r['inverse_newton_ms'] = float(np.mean([a.elapsed_time(b) for a, b in evn]))
            r['roundtrip_median_abs_err_newton'] = float((Xinv[:, :N] - Xs[:, :N]).abs().median().item())
            tm.root_finder = 'reference'
        Nopt = N if separable else 100000            # (integrated-rectifier optimize(): BASELINE.md quotes N = 1e5)
        if Nopt != N:
            del tm, Xs, Z, Xinv
            tm, _, _ = build_map(name, 0, Nopt)
        for k in range(tm.D):
            tm.coeffs_mon[k] = tm.coeffs_mon[k] * 0 + tm.coeffs_init
            tm.coeffs_nonmon[k] = tm.coeffs_nonmon[k] * 0 + tm.coeffs_init
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tm.optimize()
        torch.cuda.synchronize()
        r['optimize_s'], r['optimize_N'] = time.perf_counter() - t0, Nopt
        out[name] = r
        del tm
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--prewarm-seconds', type=float, default=2.0,
                    help='untimed back-to-back steps before the warm-up, until the chip holds its clock under load '
                         '(MI355X_MICROARCH.md, DVFS item 6: >= 2 s); 0 = none')
    ap.add_argument('--workload', default='C5', choices=sorted(WORKLOADS))
    ap.add_argument('--n', '--samples', dest='n', type=int, default=0, help='override the ensemble size (testing only)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="torch.distributed backend for --gpus > 1 ('nccl' = RCCL; 'gloo' only to rehearse on one GPU)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-optimize', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the secondary numbers of the other single-GPU BASELINE configurations (C2b, C2a, C3)')
    ap.add_argument('--cpu-samples', type=int, default=0, help='samples per host process of the CPU baseline')
    ap.add_argument('--cpu-cores', type=int, default=0, help='host processes of the CPU baseline (default 1)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus and world > 1:
        raise SystemExit('WORLD_SIZE (%d) != --gpus (%d)' % (world, args.gpus))
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        # first, before anything touches the GPU: the host baseline forks one process per core
        n_cpu = args.cpu_samples or {'C5': 300000, 'C3': 500000, 'C2b': 500000, 'C2a': 5000}[args.workload]
        # one process by default: on the one-GPU boxes of this pool 8 (64) concurrent oracle processes took 7x (47x)
        # longer each - the aggregate stayed at 1.3e6 (1.5e6) map-evals/s against 1.1e6 for one process alone
        cores = args.cpu_cores or 1
        cpu = cpu_baseline(args.workload, n_cpu, cores)
    import torch
    local_dev = local_rank % max(1, torch.cuda.device_count())       # (rehearsals put several ranks on one GPU)
    torch.cuda.set_device(local_dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_dev))
        else:
            dist.init_process_group(args.backend)

    tm, X, cfg = build_map(args.workload, rank, args.n or None)
    N, D, d = tm._N, tm.D, tm._cm.d_cols
    du = d_used(tm)
    separable = tm.monotonicity == 'separable monotonicity'
    coef = tm._pack_coeffs()
    Xs = tm._Xs
    Z = tm._cols(D, N)
    Xinv = tm._cols(d, N, zero=True)

    def step():
        tm.forward_device(Xs, N, coef=coef, Z=Z)
        tm.inverse_device(Z, N, coef=coef, X=Xinv)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # After any idle gap of a millisecond or more (and from a cold start) the chip runs this workload 10-30 % slower
    # for the next ~50 launches (20 ms) before it holds its clock again (tools/ramp_probe.py: forward launches of
    # 0.17-0.20 ms after a 1-100 ms pause against 0.153 ms back to back; a bare stream synchronisation costs nothing).
    # So the card is first kept busy with the same steps, untimed, for --prewarm-seconds, the W warm-up steps follow
    # without a pause, and only the barrier + synchronisation of the contract separates them from the K timed steps.
    # What is timed is unchanged: exactly K full steps between barriers.
    prewarm_steps = 0
    if args.prewarm_seconds > 0:
        t_pw = time.perf_counter()
        while time.perf_counter() - t_pw < args.prewarm_seconds:
            for _ in range(50):
                step()
            torch.cuda.synchronize()
            prewarm_steps += 50
        if dist is not None:
            # the ranks leave the time-based loop up to one batch apart; the one that waits at a barrier idles and would
            # start its timed steps on the ramp.  Align them once, then give every rank the same number of steps, so
            # that all arrive at the barrier in front of the timed region within a fraction of a millisecond
            sync()
            for _ in range(150):
                step()
            prewarm_steps += 150
    for _ in range(args.warmup):
        step()
    sync()                                   # (nothing else between the warm-up and the timed steps: see above)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    # per-kernel timing of the dominant kernel (forward map) with HIP events on the launch stream, directly behind
    # the timed steps (same clock state)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    ev_inv = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for (a, b), (c, e) in zip(ev, ev_inv):
        a.record()
        tm.forward_device(Xs, N, coef=coef, Z=Z)
        b.record()
        c.record()
        tm.inverse_device(Z, N, coef=coef, X=Xinv)
        e.record()
    torch.cuda.synchronize()
    fwd_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    inv_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_inv]))
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # round trip sanity inside the bench: S^{-1}(S(x)) == x up to the inverse's own accuracy
    err = float((Xinv[:, :N] - Xs[:, :N]).abs().max().item())
    extra = {}
    if separable:
        ld = tm._empty(N)
        ss = tm._empty(N)
        sigma = tm._to_dev(np.asarray(tm.X_std[:D], dtype=float))
        n_pb = max(5, min(args.steps, 50))
        evp = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_pb)]
        for _ in range(150 if args.prewarm_seconds > 0 else 0):             # (the allocations above were an idle gap)
            tm.forward_device(Xs, N, coef=coef, Z=Z, logdet=ld, sigma=sigma, sumsq=ss)
        for a, b in evp:
            a.record()
            tm.forward_device(Xs, N, coef=coef, Z=Z, logdet=ld, sigma=sigma, sumsq=ss)   # fused S, log det, |S|^2
            b.record()
        torch.cuda.synchronize()
        extra['pullback_fused_ms'] = float(np.mean([a.elapsed_time(b) for a, b in evp]))
    if world == 1 and not args.no_optimize:
        # secondary metric of BASELINE.json: optimize() wall-clock on the resident ensemble (from coeffs_init)
        saved = ([c.copy() for c in tm.coeffs_mon], [c.copy() for c in tm.coeffs_nonmon])
        for k in range(D):
            tm.coeffs_mon[k] = tm.coeffs_mon[k] * 0 + tm.coeffs_init
            tm.coeffs_nonmon[k] = tm.coeffs_nonmon[k] * 0 + tm.coeffs_init
        torch.cuda.synchronize()
        t0o = time.perf_counter()
        tm.optimize()
        torch.cuda.synchronize()
        extra['optimize_s'] = time.perf_counter() - t0o
        tm.coeffs_mon, tm.coeffs_nonmon = saved
    if world == 1 and not args.no_other_configs and args.workload == 'C5':
        try:
            extra['other_configs'] = other_configs(torch, ['C2b', 'C2a', 'C3'])
        except Exception as exc:                       # noqa: BLE001  (never fatal for the headline line)
            extra['other_configs_error'] = repr(exc)
    if world > 1 and not args.no_optimize:
        # optimize() with the COMPONENTS partitioned over the ranks (SURVEY section 8e / BASELINE config 5): every rank
        # holds the same ensemble (seed of rank 0), optimises a strided subset of the components, coefficients are
        # exchanged once and the summed objective is the one RCCL all-reduce.  Never fatal for the scaling run.
        try:
            del tm, Xs, Z, Xinv
            tm2, _, _ = build_map(args.workload, 0, args.n or None, shard_components=True)
            for k in range(tm2.D):
                tm2.coeffs_mon[k] = tm2.coeffs_mon[k] * 0 + tm2.coeffs_init
                tm2.coeffs_nonmon[k] = tm2.coeffs_nonmon[k] * 0 + tm2.coeffs_init
            sync()
            t0o = time.perf_counter()
            tm2.optimize()
            sync()
            t = torch.tensor([time.perf_counter() - t0o], dtype=torch.float64, device='cuda')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            extra['optimize_component_sharded_s'] = float(t.item())
            extra['optimize_objective_total'] = float(tm2.objective_total)
        except Exception as exc:                       # noqa: BLE001
            extra['optimize_component_sharded_error'] = repr(exc)

    if rank == 0:
        fwd_bytes = 8.0 * N * (du + D)
        inv_bytes = 8.0 * N * (2 * D)
        achieved = fwd_bytes / (fwd_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            traffic = tj.get(args.workload, {}).get('k_forward_hbm_bytes_per_launch')
        out = {
            'metric': 'map-evals/sec (forward+inverse, N samples x D comps)',
            'value': world * N * D * args.steps / elapsed,
            'unit': 'map-evals/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': args.workload + ': ' + WORKLOADS[args.workload][3],
                       'N_per_gpu': N, 'D': D, 'columns': d, 'inverse': 'table' if separable else 'bisection',
                       'layout': 'column-major resident in HBM', 'coefficients': 'reference-optimised (tests/golden)'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'kernel': 'k_forward_hl (forward map; k_forward_u / k_forward_plan for maps without hot records)',
                         'algorithmic_bytes_per_launch': fwd_bytes, 'avg_launch_ms': fwd_ms},
            'forward_ms': fwd_ms, 'inverse_ms': inv_ms,
            'inverse_GBps_algorithmic': inv_bytes / (inv_ms * 1e-3) / 1e9,
            'roundtrip_max_abs_err': err,
            'prewarm': {'seconds': args.prewarm_seconds, 'steps': prewarm_steps,
                        'why': 'untimed steps before the warm-up so that the timed steps run at the clock the chip holds under '
                               'sustained load (after an idle gap >= 1 ms the next ~50 launches run 10-30 % slower)'},
        }
        out.update(extra)
        if cpu is not None:
            out['cpu_baseline'] = cpu
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
