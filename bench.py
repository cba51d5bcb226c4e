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

prints ONE JSON line (rank 0) with the `roofline` object of the dominant kernel - whichever of the two map
kernels has the longer measured launch (today the table inverse; algorithmic bytes 8 N (d_used + D) forward,
8 N (2 D + E) inverse) - the fractions of the other kernel and of the pair, an `fp64` block
(arithmetic rate next to the byte rate) and a `cpu_baseline` object (rank 0, N=1).

`--gpus N` with N > 1 and no launcher in the environment (WORLD_SIZE unset) starts the N ranks itself: N fresh
child processes, one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, before this
process has touched the GPU; rank 0's JSON line is relayed.  Under torchrun (WORLD_SIZE set) every process is a rank.
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
COPY_PEAK_GBS = 6290.0       # measured float4 copy of the same guide (79 % of the spec)
FP64_PEAK_TFLOPS = 78.6      # vector fp64: half the 157.3 TFLOP/s fp32 vector peak of the same guide (16 lanes/clk/SIMD)

WORKLOADS = {
    # name: (config, N, golden coefficient fixture, description)
    'C5': ('C5', 1000000, 'c5_sep', 'd=40 banded separable map, order 3, N=1e6 (Gaussian-mixture target)'),
    'C3': ('C3', 500000, 'c3_sep', 'd=4 dense separable map, order 4, N=5e5 (banana target)'),
    'C2b': ('C2b', 1000000, 'c2b_sep', 'spiral d=2 separable map, order 5, N=1e6'),
    'C2a': ('C2a', 1000000, 'c2a_int', 'spiral d=2 integrated-rectifier map, order 5, Q=25, N=1e6'),
    # the integrated-rectifier variants SURVEY.md section 8 names as canonical (FP64-bound: secondary numbers)
    'C3int': ('C3int', 500000, 'c3_int', 'd=4 band-1 integrated-rectifier map, order 4, Q=25, N=5e5 (banana target)'),
    'C5int': ('C5int', 200000, 'c5_int', 'd=40 band-2 integrated-rectifier map, order 3, Q=25, N=2e5 (Gaussian-mixture target)'),
    # the reference's own shipped examples at their shipped orders (example_01.py:126, example_03.py:103: maxorder = 10)
    'EX01': ('EX01', 1000000, 'ex01_order10', 'Example 01 as shipped: spiral d=2 integrated-rectifier map, order 10 (66 coefficients in the second '
                                              'component, the shipped pickle), Q=25, N=1e6'),
    'EX03': ('EX03', 1000000, 'ex03_order10', 'Example 03 as shipped: d=2 separable map, order 10 (LET + 9 iRBF + RET, Hermite functions 1..10), N=1e6 (spiral target)'),
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
    seed = {'C5': 12345, 'C5int': 12345}.get(workload, 0) + 1000 * rank
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


def _oracle_for(workload, n, seed):
    from oracle.ttm_oracle import OracleMap      # timed baseline only
    from triangular_transport_toolbox_amd import specs
    cfgname, _, fixture, _ = WORKLOADS[workload]
    cfg = specs.config(cfgname)
    Xc = cfg['sampler'](n, seed=seed)
    om = OracleMap(X=Xc, monotone=cfg['monotone'], nonmonotone=cfg['nonmonotone'], **cfg['kwargs'])
    om.coeffs_mon, om.coeffs_nonmon = load_coeffs(fixture, om.D)
    return om, Xc


def _numpy_leg(args):
    """One host core: the NumPy oracle on its own n-sample ensemble of the workload (forward + inverse)."""
    workload, n, seed = args
    for v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
        os.environ[v] = '1'
    om, Xc = _oracle_for(workload, n, seed)
    t0 = time.perf_counter()
    Z = om.map(Xc)
    t1 = time.perf_counter()
    om.inverse_map(Z)
    t2 = time.perf_counter()
    return n * om.D, t1 - t0, t2 - t1


def cpu_baseline(workload, seconds):
    """The CPU baseline beside the GPU number, two legs, each bounded to about `seconds`:
    (1) the C++/OpenMP restatement of the reference's forward map + table inverse (oracle/ttm_oracle_omp.cpp) on ALL
        usable host cores - `value`; separable workloads only;
    (2) the NumPy oracle (oracle/ttm_oracle.py, the reference's own vectorised NumPy formulation) on one core, in a
        forked worker so that its thread pools are pinned to one thread.
    Must run BEFORE this process initialises the GPU (leg 2 forks)."""
    import multiprocessing as mp
    from oracle import omp
    cores, core_info = omp.usable_cores()
    out = {'unit': 'map-evals/s', 'kind': 'port', 'host': core_info}
    # leg 2: NumPy, one core; sample sized from a short calibration run
    with mp.get_context('fork').Pool(1) as pool:
        n0 = {'C5': 20000, 'C3': 50000, 'C2b': 100000, 'C2a': 500, 'C3int': 500, 'C5int': 100, 'EX01': 200, 'EX03': 50000}[workload]
        ev, tf, ti = pool.map(_numpy_leg, [(workload, n0, 7000)])[0]
        n1 = int(max(n0, min(20 * n0, n0 * seconds / max(tf + ti, 1e-3))))
        ev, tf, ti = pool.map(_numpy_leg, [(workload, n1, 7013)])[0]
    out['numpy_1core'] = {'value': ev / (tf + ti), 'cores': 1, 'samples': n1, 'forward_s': tf, 'inverse_s': ti,
                          'what': 'oracle/ttm_oracle.py (NumPy restatement of the reference CPU path), one process, one thread'}
    if workload in ('C2a', 'C3int', 'C5int', 'EX01'):
        out.update(value=out['numpy_1core']['value'], cores=1,
                   sample='NumPy oracle, %d samples of the workload, forward + bisection inverse, one core '
                          '(the OpenMP leg restates the separable path only)' % n1)
        return out
    # leg 1: C++ / OpenMP on all usable cores
    n0 = 200000
    om, Xc = _oracle_for(workload, n0, 7026)
    m = omp.OmpMap(om)
    Xs = (Xc - om.X_mean) / om.X_std

    def run(Xs_):
        t0 = time.perf_counter()
        Z = m.forward_std(Xs_, cores)
        t1 = time.perf_counter()
        m.inverse_std(Z, cores)
        return t1 - t0, time.perf_counter() - t1
    tf, ti = run(Xs)                                  # calibration (also warms the thread pool)
    reps = int(max(1, min(200, seconds / max(tf + ti, 1e-4))))
    t_f = t_i = 0.0
    for _ in range(reps):
        a, b = run(Xs)
        t_f += a
        t_i += b
    out.update(value=reps * n0 * om.D / (t_f + t_i), cores=cores,
               sample='oracle/ttm_oracle_omp.cpp (C++/OpenMP restatement of the reference forward map TM:2391-2567 + table '
                      'inverse TM:3987-4084, term by term, libm erf/exp, fresh 1001-point tables per call), %d threads, '
                      '%d passes over a %d-sample ensemble of the workload: forward %.2f s, inverse %.2f s'
                      % (cores, reps, n0, t_f, t_i),
               forward_s=t_f, inverse_s=t_i, samples=reps * n0)
    return out


def counted_fp64(workload, N, forward_ms, inverse_ms, newton_ms=None):
    """fp64 rate of the integrated-rectifier kernels from COUNTED operations (profiles/fp64_counts.json, written by
    tools/fp64_counts.py from rocprofv3 PMC passes of this bench on the same workload); None without that file."""
    path = os.path.join(ROOT, 'profiles', 'fp64_counts.json')
    if not os.path.exists(path):
        return None
    c = json.load(open(path)).get(workload)
    if not c:
        return None
    out = {'peak_TFLOPs': FP64_PEAK_TFLOPS, 'estimate': False, 'source': 'profiles/fp64_counts.json (rocprofv3 --pmc, counted)',
           'definition': c.get('definition')}
    scale = N / float(c['N'])
    for key, kern, ms in (('forward', c.get('forward_kernel'), forward_ms), ('inverse', c.get('inverse_kernel'), inverse_ms),
                          ('inverse_newton', c.get('newton_kernel'), newton_ms)):
        k = c['kernels'].get(kern) if kern else None
        if k is None or not ms:
            continue
        # per CALL: the reference-sequence inversion is two launches (samples 1..N-1, then the replay of sample 0) and `ms`
        # times both - tools/fp64_counts.py sums the counted operations over the launches of one call
        flop = k.get('flop_per_call', k['flop_per_launch']) * scale
        out[key] = {'kernel': kern, 'flop_per_call': flop, 'launches_per_call': k.get('launches_per_call', 1),
                    'TFLOPs': flop / (ms * 1e-3) / 1e12, 'frac': flop / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                    'valu_instructions_per_call': k.get('valu_per_call', k.get('valu_per_launch', 0)) * scale,
                    'salu_per_valu': (k['salu_per_call'] / k['valu_per_call'] if k.get('valu_per_call') else None)}
    return out


def _last_kernel(tm):
    """name of the kernel the library's most recent launching entry point dispatched (include/ttm.h: ttm_last_kernel)"""
    import ctypes
    tm._lib.ttm_last_kernel.restype = ctypes.c_char_p
    return tm._lib.ttm_last_kernel().decode()


def graph_ms(torch, fn, launches=20, reps=10):
    """ms per call of `fn` (a function that only launches kernels on the current stream), its launches replayed from a
    captured HIP graph: the launches of the small configurations are shorter than the Python call that makes them (a
    12 us call around a 9 us kernel), so a Python loop between two events times the host.  None when capture fails."""
    try:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=side):
                for _ in range(launches):
                    fn()
        torch.cuda.synchronize()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / (launches * reps)
    except Exception:                                   # noqa: BLE001
        torch.cuda.synchronize()
        return None


def _events_ms(torch, fn, n, warm=3):
    """mean ms per call of `fn` (launches only) over n calls, HIP events on the launch stream around EACH call"""
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev]))


def objective_roofline(torch, tm, workload, passes=3):
    """Roofline entries of the objective / gradient reductions behind optimize() (TM:3300-3635 integrated, TM:2978-3018 +
    TM:2966-2975 separable) - the third kernel family north_star names.  One PASS = one evaluation of every component at
    its current coefficients on the resident ensemble.
    Integrated maps: `k_int_objective` (or the generic `k_objective`), FP64-bound: counted fp64 operations of a pass
    (profiles/fp64_counts.json: rocprofv3 --pmc passes of THIS function under `--no-optimize`, so every dispatch of the
    kernel in the profile belongs to a pass) over the event-timed pass.
    Separable maps: the Gram pass (`k_gram_mfma`: algorithmic bytes 8 N d_k; counted fp64 beside it) and one
    L-BFGS-B evaluation on the cached derivative basis (`k_objective_sep_cached`: the kernel streams the m cached columns,
    8 N m bytes; SURVEY's figure for the evaluation, 8 N d_k with d_k = 1, beside it)."""
    N, D = tm._N, tm.D
    separable = tm.monotonicity == 'separable monotonicity'
    counts = {}
    path = os.path.join(ROOT, 'profiles', 'fp64_counts.json')
    if os.path.exists(path):
        counts = json.load(open(path)).get(workload, {})
    scale = N / float(counts.get('N', N))

    def counted(prefix):
        c = [(k, v) for k, v in counts.get('kernels', {}).items() if k.startswith(prefix + '<') or k == prefix]
        return max(c, key=lambda kv: kv[1]['dispatches'] or 0) if c else (None, None)
    out = {}
    if not separable:
        cs = [np.ascontiguousarray(np.concatenate((tm.coeffs_nonmon[k], tm.coeffs_mon[k])), dtype=float) for k in range(D)]
        if any(len(c) > 128 for c in cs):
            return None

        def one_pass():
            for k in range(D):
                tm._objective_launch(k, cs[k])
        one_pass()
        kern = _last_kernel(tm)
        ms_loop = _events_ms(torch, one_pass, passes)
        # the launches of a pass replayed from a captured HIP graph: the evaluations of the small ensembles (C5-int: 40 launches
        # of ~35 us) are not much longer than the Python call that launches them
        ms = graph_ms(torch, one_pass, launches=max(1, 40 // D), reps=5)
        how = 'HIP events around graph replays of whole passes'
        if ms is None:
            ms, how = ms_loop, 'HIP events around each pass of a Python loop'
        e = {'bound': 'fp64 valu', 'kernel': kern, 'unit': 'TFLOP/s', 'peak': FP64_PEAK_TFLOPS, 'pass_ms': ms, 'timing': how,
             'python_loop_pass_ms': ms_loop, 'evaluations_per_pass': D, 'ms_per_evaluation': ms / D, 'N': N,
             'what': 'one objective + gradient evaluation of every component (sums over the whole ensemble), one launch each'}
        name, c = counted(kern)
        if c is not None:
            # a pass launches one template instance per component (chunk count by coefficient count): the flop of a pass is the
            # dispatch-weighted sum over the instances of the counting run, which ran whole passes only
            inst = [v for k, v in counts.get('kernels', {}).items() if k.startswith(kern + '<') or k == kern]
            launches = sum(v['dispatches'] or 0 for v in inst)
            flop = (sum(v['flop_per_launch'] * (v['dispatches'] or 0) for v in inst) / launches * D if launches
                    else c['flop_per_launch'] * D) * scale
            e.update(counted_kernel=name, flop_per_pass=flop, achieved=flop / (ms * 1e-3) / 1e12,
                     frac=flop / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, wait_any_frac=c.get('wait_any_frac'),
                     salu_per_valu=(c['salu_per_launch'] / c['valu_per_launch'] if c.get('valu_per_launch') else None),
                     source='profiles/fp64_counts.json (rocprofv3 --pmc, counted; dispatch-weighted over the template instances of a pass)')
        out['objective'] = e
        return out
    # separable: Gram pass
    m_tot = [int(tm._cm.n_nm[k] + tm._cm.n_mon[k]) for k in range(D)]
    gout = tm._empty(int(sum(m * m for m in m_tot)))
    offs = np.concatenate(([0], np.cumsum([m * m for m in m_tot]))).astype(int)
    work = tm._workspace(tm._lib.ttm_reduce_work_size(max(m * m for m in m_tot)))
    from triangular_transport_toolbox_amd import _capi

    def gram_pass():
        for k in range(D):
            _capi.check(tm._lib.ttm_gram(tm._pp, int(k), tm._ptr(tm._Xs), tm._Xs.shape[1], N, tm._ptr(work), tm._ptr(gout, int(offs[k])),
                                         tm._stream()))
    gram_pass()
    kern = _last_kernel(tm)
    ms = _events_ms(torch, gram_pass, passes)
    cols = []
    for k in range(D):
        c = {k + tm.skip_dimensions}
        for entry in list(tm.monotone[k]) + list(tm.nonmonotone[k]):
            c.update([int(entry.split(' ')[1])] if isinstance(entry, str) else [int(e) for e in entry if not isinstance(e, str)])
        cols.append(len(c))
    gb = 8.0 * N * sum(cols)
    e = {'bound': 'hbm', 'kernel': kern, 'unit': 'GB/s', 'peak': HBM_PEAK_GBS, 'pass_ms': ms, 'launches_per_pass': D,
         'algorithmic_bytes_per_pass': gb, 'achieved': gb / (ms * 1e-3) / 1e9, 'frac': gb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
         'what': 'Gram matrices [Psi_nonmon | Psi_mon]^T [Psi_nonmon | Psi_mon] of every component (TM:2966-2975, 3031-3050), '
                 '8 N d_k bytes each; the launch is bound by evaluating the basis, not by the stream: fp64 beside it'}
    name, c = counted(kern)
    if c is not None:
        flop = c['flop_per_launch'] * D * scale
        e['fp64'] = {'counted_kernel': name, 'flop_per_pass': flop, 'achieved_TFLOPs': flop / (ms * 1e-3) / 1e12,
                     'frac': flop / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 'peak_TFLOPs': FP64_PEAK_TFLOPS}
    out['gram'] = e
    # ... and one evaluation of the reduced objective on the cached derivative basis (component D // 2)
    k = D // 2
    tm._sep_cache_begin(k)
    if tm._sep_cache is not None:
        c_k = np.ascontiguousarray(tm.coeffs_mon[k], dtype=float)
        m = len(c_k)
        # the evaluation as the native L-BFGS-B loops launch it: self-validating sums, ONE launch whatever the grid
        # (ttm_objective_sep_cached_sent); beside it the ticket / two-launch evaluation of rounds 3-5 (`two_launch_ms`), and the two
        # checked against each other bit for bit (the sums land in the same page-locked vector)
        ev_launch = tm._sep_objective_launch
        two_ms = graph_ms(torch, lambda: tm._sep_objective_launch(c_k))
        tm._sep_objective_launch(c_k)
        torch.cuda.synchronize()
        ref_sums = tm._obj_out[:1 + m].numpy().copy()
        sent_equal = None
        if tm._sep_objective_launch_sent(c_k):
            torch.cuda.synchronize()
            sent_equal = bool(np.array_equal(tm._obj_out[:1 + m].numpy(), ref_sums))
            ev_launch = tm._sep_objective_launch_sent
        ev_launch(c_k)
        kern = _last_kernel(tm)
        ms = graph_ms(torch, lambda: ev_launch(c_k))
        how = 'HIP graph replay (20 evaluations per graph)'
        if ms is None:
            ms, how = _events_ms(torch, lambda: ev_launch(c_k), 50), 'HIP events around each evaluation'
        out['objective_separable'] = {
            'bound': 'hbm', 'kernel': kern, 'unit': 'GB/s', 'peak': HBM_PEAK_GBS, 'ms_per_evaluation': ms, 'timing': how, 'component': k,
            'm': m, 'algorithmic_bytes': 8.0 * N * m, 'achieved': 8.0 * N * m / (ms * 1e-3) / 1e9,
            'frac': 8.0 * N * m / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            'survey_8d_bytes': 8.0 * N, 'survey_8d_frac': 8.0 * N / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            'self_validating_sums': ev_launch is not tm._sep_objective_launch, 'equals_two_launch_sums': sent_equal, 'two_launch_ms': two_ms,
            'what': 'one L-BFGS-B evaluation of the reduced separable objective (TM:2978-3018): the kernel streams the m cached '
                    'columns of dPsi_mon (8 N m bytes - what the reference keeps as der_Psi_mon); SURVEY section 8d counts an evaluation as '
                    '8 N d_k with the basis recomputed from the d_k = 1 column it depends on'}
    tm._sep_cache_end()
    return out


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
        tm.forward_device(Xs, N, coef=coef, Z=Z)
        fwd_kernel = _last_kernel(tm)
        tm.inverse_device(Z, N, coef=coef, X=Xinv)
        inv_kernel = _last_kernel(tm)
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
                 forward_kernel=fwd_kernel, inverse_kernel=inv_kernel,
                 forward_ms=float(np.mean([a.elapsed_time(b) for a, b, c in ev])),
                 inverse_ms=float(np.mean([b.elapsed_time(c) for a, b, c in ev])),
                 inverse='table' if separable else 'bisection (reference sequence)',
                 roundtrip_max_abs_err=float((Xinv[:, :N] - Xs[:, :N]).abs().max().item()),
                 # (the max sits in the tails, where the reference's 1001-point table inverse clips / interpolates
                 # coarsely or the bisection window runs away; the median is the inverse's working accuracy)
                 roundtrip_median_abs_err=float((Xinv[:, :N] - Xs[:, :N]).abs().median().item()))
        if separable:
            # launches replayed from a graph (see graph_ms): the launch times themselves, and the step back to back
            gf = graph_ms(torch, lambda: tm.forward_device(Xs, N, coef=coef, Z=Z))
            gi = graph_ms(torch, lambda: tm.inverse_device(Z, N, coef=coef, X=Xinv))
            gs = graph_ms(torch, step)
            if gf is not None and gi is not None and gs is not None:
                r.update(timing='HIP graph replay (20 launches per graph)', python_loop_ms_per_step=r['ms_per_step'],
                         python_loop_forward_ms=r['forward_ms'], python_loop_inverse_ms=r['inverse_ms'],
                         forward_ms=gf, inverse_ms=gi, ms_per_step=gs, value=N * D / (gs * 1e-3))
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
            gp = graph_ms(torch, lambda: tm.forward_device(Xs, N, coef=coef, Z=Z, logdet=ld, sigma=sigma, sumsq=ss))
            gl = graph_ms(torch, lambda: tm.density_device(Xs, N, coef=coef, logdet=ld, sigma=sigma))
            if gp is not None:
                r['pullback_fused_ms'] = gp
            # roofline of the configuration's launches on their algorithmic bytes: forward and
            # inverse 8 N (d + D) each, the log-determinant-only pullback pass 8 N (d + 1)
            fb, ib = 8.0 * N * (d_used(tm) + D), 8.0 * N * (d_used(tm) + D)
            roof = {'bound': 'hbm', 'unit': 'GB/s', 'peak': HBM_PEAK_GBS, 'copy_peak': COPY_PEAK_GBS,
                    'forward': {'kernel': fwd_kernel, 'bytes': fb, 'achieved': fb / (r['forward_ms'] * 1e-3) / 1e9},
                    'inverse': {'kernel': inv_kernel, 'bytes': ib, 'achieved': ib / (r['inverse_ms'] * 1e-3) / 1e9},
                    'step': {'bytes': fb + ib, 'achieved': (fb + ib) / (r['ms_per_step'] * 1e-3) / 1e9}}
            if gl is not None:
                lb = 8.0 * N * (d_used(tm) + 1)
                r['pullback_logdet_only_ms'] = gl
                roof['pullback_logdet_only'] = {'kernel': _last_kernel(tm), 'bytes': lb, 'achieved': lb / (gl * 1e-3) / 1e9}
            # the step - and the step with the density terms, "forward + inverse + pullback" of BASELINE configs[1] - as ONE launch
            # (ttm_roundtrip: maps of at most four components; the columns of a tile are read once, z stays in registers).  The same
            # ALGORITHMIC bytes as the launches it stands for; checked against them bit for bit before it is timed.
            Xr = tm._cols(d, N, zero=True)
            tm.roundtrip_device(Xs, N, coef=coef, Z=Z, Xr=Xr)
            rt_kernel = _last_kernel(tm)
            if rt_kernel.startswith('k_band_few_roundtrip'):
                step()
                same = bool(torch.equal(Xr[:, :N], Xinv[:, :N]))
                gr = graph_ms(torch, lambda: tm.roundtrip_device(Xs, N, coef=coef, Z=Z, Xr=Xr))
                tm.roundtrip_device(Xs, N, coef=coef, Z=Z, Xr=Xr, logdet=ld, sigma=sigma, sumsq=ss)
                # (with the density terms the library leaves the step to two launches - the fused kernel runs out of registers there)
                gd = graph_ms(torch, lambda: tm.roundtrip_device(Xs, N, coef=coef, Z=Z, Xr=Xr, logdet=ld, sigma=sigma, sumsq=ss)) \
                    if _last_kernel(tm).startswith('k_band_few_roundtrip') else None
                if gr is not None:
                    r.update(roundtrip_fused_ms=gr, roundtrip_fused_equals_two_launches=same, value_fused=N * D / (gr * 1e-3))
                    roof['step_fused'] = {'kernel': rt_kernel, 'bytes': fb + ib, 'achieved': (fb + ib) / (gr * 1e-3) / 1e9,
                                          'hbm_bytes_moved': 8.0 * N * (d_used(tm) + 2 * D)}
                if gd is not None:
                    db = fb + ib + 16.0 * N
                    r['roundtrip_fused_with_density_ms'] = gd
                    roof['step_fused_with_density'] = {'kernel': 'k_band_few_roundtrip<density>', 'bytes': db, 'achieved': db / (gd * 1e-3) / 1e9,
                                                       'hbm_bytes_moved': 8.0 * N * (d_used(tm) + 2 * D + 2)}
            for v in roof.values():
                if isinstance(v, dict):
                    v['frac'] = v['achieved'] / HBM_PEAK_GBS
            r['roofline'] = roof
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
            # FP64-bound path: arithmetic rate next to the byte rate.  Executed fp64 operations per
            # launch are COUNTED (SQ_INSTS_VALU_FMA/ADD/MUL/TRANS_F64 of tools/fp64_counts.sh, profiles/fp64_counts.json:
            # 64 lanes x (2 FMA + ADD + MUL + TRANS) wave-instructions per launch at the profiled N, scaled to this N)
            r['fp64'] = counted_fp64(name, N, r['forward_ms'], r['inverse_ms'], r.get('inverse_newton_ms'))
            r['roundtrip_median_abs_err_newton'] = float((Xinv[:, :N] - Xs[:, :N]).abs().median().item())
            tm.root_finder = 'reference'
        try:
            r['objective_roofline'] = objective_roofline(torch, tm, name)
        except Exception as exc:                       # noqa: BLE001
            r['objective_roofline_error'] = repr(exc)
        def timed_optimize(t):
            for k in range(t.D):
                t.coeffs_mon[k] = t.coeffs_mon[k] * 0 + t.coeffs_init
                t.coeffs_nonmon[k] = t.coeffs_nonmon[k] * 0 + t.coeffs_init
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            t.optimize()
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        r['optimize_first_call_s'] = timed_optimize(tm)
        r['optimize_s'], r['optimize_N'] = timed_optimize(tm), N
        if not separable:
            # (integrated-rectifier optimize(): BASELINE.md quotes the reference at N = 1e5 - that size beside the full one)
            r['optimize_full_N_s'] = r['optimize_s']
            del tm, Xs, Z, Xinv
            tm, _, _ = build_map(name, 0, 100000)
            timed_optimize(tm)
            r['optimize_s'], r['optimize_N'] = timed_optimize(tm), 100000
        out[name] = r
        del tm
    return out


def entf_config(torch, N=100000, cycles=2000):
    """BASELINE configs[3] (C4): Lorenz-63 Ensemble Transport Filter, Example-06 map (4 columns, D = 3), N = 1e5:
    `cycles` assimilation cycles timed end to end, each = three one-observation updates
    (reset -> optimize -> map -> inverse_map with the observation as X_star) + the RK4 forecast."""
    from triangular_transport_toolbox_amd import entf
    rng = np.random.default_rng(0)
    truth = np.array([1.0, 1.0, 25.0])
    ens = rng.standard_normal((N, 3)) * [8, 9, 8] + [0, 0, 25]
    flt = entf.Filter(N, seed=0) if hasattr(entf, 'Filter') else None
    if flt is not None:
        return flt.benchmark(ens, truth, cycles)
    tm = entf.make_filter_map(N)

    def cycle(ens, truth):
        truth = entf.rk4(truth[None, :], 0.05, 2)[0]
        obs = truth + 2.0 * rng.standard_normal(3)
        ens = entf.rk4(ens, 0.05, 2)
        noises = [2.0 * rng.standard_normal(N) for _ in range(3)]
        return entf.assimilate(tm, ens, obs, noises), truth
    for _ in range(3):
        ens, truth = cycle(ens, truth)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(cycles):
        ens, truth = cycle(ens, truth)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return dict(workload='C4: Lorenz-63 EnTF, Example-06 map (4 columns, D = 3, order 3, L2 0.05), host-driven loop', N=N,
                cycles=cycles, ms_per_cycle=1e3 * el / cycles, updates_per_cycle=3,
                rmse_last=float(np.sqrt(np.mean((ens.mean(axis=0) - truth) ** 2))))


def ents_block_config(torch, N=100000, steps=20, check=True):
    """BASELINE configs[3], second half: the 6-column block map of the Ensemble Transport Smoother (example_07.py:368-465:
    X is N x 6, skip_dimensions 3, D = 3, dense nonmonotone blocks over all earlier columns, linear monotone terms, L2 0.05),
    N = 1e5.  One backward step = reset -> optimize -> map -> inverse_map with X_star through the class (host arrays in and
    out, as entf.smooth drives it); the filtering ensembles are Lorenz-63 forecasts / analyses of a synthetic run.
    `check`: the CPU leg of this configuration - the oracle's backward step on the same ensembles (NumPy + SciPy L-BFGS-B on
    the host, timed once: `cpu_step_s`) - and the step that was timed is compared with it (1e-5, the optimiser's tolerance, as
    tests/test_full_size.py::test_c4_block_map_backward_step_at_1e5_against_the_oracle)."""
    from triangular_transport_toolbox_amd import entf
    rng = np.random.default_rng(0)
    ana = rng.standard_normal((N, 3)) * [8.0, 9.0, 8.0] + [0.0, 0.0, 25.0]
    fc = entf.rk4(ana, 0.05, 2)
    fc_next = entf.rk4(fc, 0.05, 2)
    tm = entf.make_smoother_map(N, maxorder=3, lmbda=0.05)
    import copy
    map_input = np.column_stack((fc_next, fc))

    def step(Xnext):
        tm.reset(copy.copy(map_input))
        tm.optimize()
        Zp = tm.map(map_input)
        return tm.inverse_map(X_star=Xnext, Z=Zp)
    Xs = fc_next + 0.1 * rng.standard_normal((N, 3))
    for _ in range(3):
        out = step(Xs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step(Xs)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    tm.forward_device(tm._Xs, tm._N)
    fk = _last_kernel(tm)
    tm.inverse_device(tm._cols(tm.D, tm._N, zero=True), tm._N)
    ik = _last_kernel(tm)
    r = dict(workload='C4 block map: Ensemble Transport Smoother backward step, Example-07 map (6 columns, D = 3, skip 3, order 3, '
                      'L2 0.05), reset -> optimize -> map -> inverse_map with X_star through the class (host arrays)',
             N=N, steps=steps, ms_per_step=1e3 * el / steps, forward_kernel=fk, inverse_kernel=ik,
             finite=bool(np.all(np.isfinite(out))))
    if check:
        from oracle.ttm_oracle import OracleMap      # CPU leg of this configuration: timed, and the checker of the step above
        from triangular_transport_toolbox_amd import specs
        mon, non = specs.ents_smoother_spec(3)
        t0 = time.perf_counter()
        om = OracleMap(X=map_input.copy(), monotone=mon, nonmonotone=non, polynomial_type="probabilist's hermite",
                       monotonicity='separable monotonicity', regularization='l2', regularization_lambda=0.05)
        om.optimize()
        want = om.inverse_map(X_star=Xs.copy(), Z=om.map(map_input))
        r['cpu_step_s'] = time.perf_counter() - t0
        r['cpu_step'] = 'oracle/ttm_oracle.py (NumPy + SciPy L-BFGS-B), one process, the same backward step'
        err = float(np.max(np.abs(out - want) / (np.abs(want) + 1.0)))
        r['max_rel_err_vs_oracle_step'] = err
        if not err < 1e-5:
            raise RuntimeError('C4 block map: the timed backward step differs from the oracle by %.3e' % err)
    return r


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def spawn_ranks(args):
    """`--gpus N` without a launcher: start N fresh rank processes (one per GPU) and relay rank 0's JSON line.
    Runs before this process has touched the GPU (torch.cuda.device_count() does not initialise it); the children are
    new interpreters, never an exec of a process that holds the device."""
    import subprocess
    import torch
    ndev = torch.cuda.device_count()
    if args.backend == 'nccl' and ndev < args.gpus:
        raise SystemExit('bench.py --gpus %d: only %d HIP device(s) visible - RCCL needs one GPU per rank '
                         "('--backend gloo' rehearses the multi-rank path with several ranks per GPU)" % (args.gpus, ndev))
    if ndev < 1:
        raise SystemExit('bench.py: no HIP device visible')
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    # exactly one JSON line on stdout: rank 0's; anything else it printed (gloo's connection banner) goes to stderr
    lines = out0.splitlines()
    js = [ln for ln in lines if ln.startswith('{"metric"')]
    for ln in lines:
        if ln not in js[-1:]:
            print(ln, file=sys.stderr)
    if js:
        print(js[-1])
    sys.stdout.flush()
    if any(rcs):
        raise SystemExit('bench.py: rank exit codes %s' % rcs)


def multi_gpu_extras(torch, dist, args, rank, world, backend):
    """What north_star asks of the node beside the weak-scaling headline, every rank taking part:
    (a) STRONG scaling of the fixed ensemble of the workload (C5: N = 1e6) split over the ranks - no collective on the
        data path, `value` = N_total D steps / the slowest rank's time;
    (b) BASELINE configs[3]: the Lorenz-63 EnTF (N = 1e5) with the ensemble SAMPLE-SHARDED over the ranks - column moments,
        order statistics and, per L-BFGS-B evaluation, ONE fused objective + gradient all-reduce (ttm_allreduce_f64 over
        RCCL inside ttm_optimize_separable; torch.distributed under `--backend gloo`) - ms per cycle and the latency of
        that all-reduce alone;
    (c) the width of the RCCL communicator the class created (`rccl_ranks`; None when the process group is not RCCL).
    Never fatal for the headline line."""
    import ctypes
    from triangular_transport_toolbox_amd import comm, entf
    if os.environ.get('TTM_BENCH_ABORT_IN_EXTRAS') == str(rank):          # (rehearsal of the guardian in main())
        os.abort()
    out = {}

    def sync():
        dist.barrier()
        torch.cuda.synchronize()

    def tmax(x):
        t = torch.tensor([x], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    try:
        n_total = args.n or WORKLOADS[args.workload][1]
        n_loc = n_total // world + (1 if rank < n_total % world else 0)
        tm, _, _ = build_map(args.workload, rank, n_loc)
        N, D, d = tm._N, tm.D, tm._cm.d_cols
        coef = tm._pack_coeffs()
        Xs, Z, Xinv = tm._Xs, tm._cols(D, N), tm._cols(d, N, zero=True)

        def step():
            tm.forward_device(Xs, N, coef=coef, Z=Z)
            tm.inverse_device(Z, N, coef=coef, X=Xinv)
        t_pw = time.perf_counter()
        while time.perf_counter() - t_pw < min(1.0, args.prewarm_seconds):
            for _ in range(50):
                step()
            torch.cuda.synchronize()
        sync()
        for _ in range(150 + args.warmup):
            step()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        el = tmax(time.perf_counter() - t0)
        out['strong_scaling'] = {'workload': args.workload, 'N_total': n_total, 'N_per_gpu': n_loc, 'steps': args.steps,
                                 'ms_per_step': 1e3 * el / args.steps, 'value': n_total * D * args.steps / el,
                                 'what': 'the fixed ensemble split over the ranks; no data-path collective'}
        del tm, Xs, Z, Xinv
    except Exception as exc:                           # noqa: BLE001
        out['strong_scaling_error'] = repr(exc)
    try:
        n_total = 100000
        base, rem = divmod(n_total, world)
        n_loc = base + (1 if rank < rem else 0)
        row0 = rank * base + min(rank, rem)
        rng = np.random.default_rng(0)
        ens = (rng.standard_normal((n_total, 3)) * [8, 9, 8] + [0, 0, 25])[row0:row0 + n_loc]
        flt = entf.Filter(n_loc, seed=0, row0=row0, shard_samples=True)
        cycles = 50
        r = flt.benchmark(ens, np.array([1.0, 1.0, 25.0]), cycles)
        tmf = flt.tm
        handle = comm.get(tmf._lib, force=False)
        nr = None
        if handle is not None:
            a, b = ctypes.c_int32(-1), ctypes.c_int32(-1)
            tmf._lib.ttm_comm_size(handle, ctypes.byref(a), ctypes.byref(b))
            nr = int(b.value)
        out['rccl_ranks'] = nr
        # the all-reduce of an optimiser evaluation by itself: 1 + m = 10 doubles, stream order, synchronised once at the end
        buf = torch.ones(10, dtype=torch.float64, device='cuda')
        for _ in range(50):
            tmf._allreduce_world(buf)
            buf.fill_(1.0)
        sync()
        n_ar = 500
        t0 = time.perf_counter()
        for _ in range(n_ar):
            tmf._allreduce_world(buf)
        sync()
        r['allreduce_us'] = 1e6 * tmax(time.perf_counter() - t0) / n_ar
        r['allreduce'] = ('ttm_allreduce_f64 over RCCL (%d ranks)' % nr) if nr else 'torch.distributed (%s): no RCCL communicator' % backend
        r['evaluations_last_update'] = getattr(tmf, 'last_optimize_evaluations', None)
        out['entf_sample_sharded'] = r
    except Exception as exc:                           # noqa: BLE001
        out['entf_sample_sharded_error'] = repr(exc)
    return out


def component_sharded_optimize(torch, dist, args):
    """optimize() with the COMPONENTS partitioned over the ranks (SURVEY section 8e / BASELINE config 5): every rank
    holds the same ensemble (seed of rank 0), optimises its share of the components, coefficients are exchanged once and
    the summed objective is the one RCCL all-reduce."""
    out = {}
    try:
        tm2, _, _ = build_map(args.workload, 0, args.n or None, shard_components=True)
        for k in range(tm2.D):
            tm2.coeffs_mon[k] = tm2.coeffs_mon[k] * 0 + tm2.coeffs_init
            tm2.coeffs_nonmon[k] = tm2.coeffs_nonmon[k] * 0 + tm2.coeffs_init
        dist.barrier(); torch.cuda.synchronize()
        t0o = time.perf_counter()
        tm2.optimize()
        dist.barrier(); torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0o], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out['optimize_component_sharded_s'] = float(t.item())
        out['optimize_objective_total'] = float(tm2.objective_total)
    except Exception as exc:                           # noqa: BLE001
        out['optimize_component_sharded_error'] = repr(exc)
    return out


def flops_per_eval(tm):
    """fp64 operations per component evaluation of the two large-ensemble map kernels, counted from the arithmetic they
    execute (FMA = 2).  Banded maps (csrc/ttm_band.hip): forward = spline (degree-11 Horner 22 + column / local
    coordinate 4) + exp(-x^2/4) from the pair table (argument 7, degree-7 Taylor 14, factor 1) + per group the two Horner
    passes of its degree class and the accumulation (2 DB + 2 DA + 2) + 1; inverse = the same pushes + target and clip 3 +
    bucket 2 + two compares 2 + interpolation (difference, floor, reciprocal with one Newton step, slope, abscissa) 11 +
    the interval exponential (argument 3, Taylor 14, factor 1).  Other U-form maps with hot records: k_forward_hl /
    k_inverse_rt as counted in round 2.  Returns (forward, inverse) averaged over the components, or None."""
    cm = tm._cm
    if not getattr(cm, 'u_enabled', False) or not getattr(cm, 'u_h_cls', 0):
        return None
    db, da = {1: (3, 1), 2: (5, 5), 3: (7, 7)}[int(cm.u_h_cls)]
    ngrp = float(np.mean(np.asarray(cm.ucomp).reshape(-1)[:cm.D * 8].reshape(cm.D, 8)[:, 2]))
    if getattr(cm, 'u_p_lag', 0):
        grp = ngrp * (2 * db + 2 * da + 2)
        return grp + 26 + 22 + 1, grp + 3 + 2 + 2 + 11 + 18
    grp = ngrp * (2 * db + 2 * da + 3)
    fwd = grp + (2 * 11 + 4) + 30
    inv = grp + 6 + 2 + (5 + 6) + (14 + 4)
    return fwd, inv


def guardian_main():
    """`bench.py --guardian` (started by rank 0 of a multi-rank run at the very top of main(), BEFORE that process imports
    torch or touches the GPU; it never touches the GPU itself): reads lines from its standard input until it closes and
    prints the LAST complete one on the standard output it inherited.  Rank 0 sends the headline line (marked
    `multi_gpu_extras_error`) before the node-level extras and the full line after them, so exactly one JSON line comes
    out whether the rank finishes, is stopped by its watchdog or dies inside a collective."""
    import signal
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, signal.SIG_IGN)           # (it ends when rank 0's end of the pipe closes, not before)
    last = None
    for line in sys.stdin.buffer:
        if line.endswith(b'\n'):
            last = line
    if last is not None:
        os.write(1, last)
    return 0


def main():
    if '--guardian' in sys.argv[1:]:
        return guardian_main()
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--prewarm-seconds', type=float, default=2.0,
                    help='untimed back-to-back steps before the warm-up, until the chip holds its clock under load '
                         '(MI355X_MICROARCH.md, DVFS item 6: >= 2 s); 0 = none.  The same K steps timed WITHOUT it are '
                         'reported as cold_ms_per_step')
    ap.add_argument('--workload', default='C5', choices=sorted(WORKLOADS))
    ap.add_argument('--n', '--samples', dest='n', type=int, default=0, help='override the ensemble size (testing only)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="torch.distributed backend for --gpus > 1 ('nccl' = RCCL; 'gloo' only to rehearse on one GPU)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-optimize', action='store_true')
    ap.add_argument('--no-api', action='store_true', help='skip the host-boundary numbers (ctor_s, api_map_ms, ...)')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the secondary numbers of the other single-GPU BASELINE configurations (C2b, C2a, C3, C4)')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='target duration of each CPU baseline leg')
    ap.add_argument('--extras-timeout', type=float, default=240.0,
                    help='N > 1: seconds the node-level extras may take before the headline line is printed without them')
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return spawn_ranks(args)                 # (before anything touches the GPU)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE (%d) != --gpus (%d)' % (world, args.gpus))
    guard = None
    if world > 1 and rank == 0:
        # the guardian of the JSON line exists before this process imports torch (see guardian_main)
        import subprocess
        guard = subprocess.Popen([sys.executable, os.path.abspath(__file__), '--guardian'], stdin=subprocess.PIPE,
                                 start_new_session=True)

    def emit(obj, last=False):
        line = json.dumps(obj) + '\n'
        if guard is None:
            sys.stdout.write(line)
            sys.stdout.flush()
            return
        try:
            guard.stdin.write(line.encode())
            guard.stdin.flush()
            if last:
                guard.stdin.close()
                guard.wait(timeout=30)
        except (OSError, ValueError, subprocess.TimeoutExpired):
            pass
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        # first, before anything touches the GPU (the NumPy leg forks a worker)
        cpu = cpu_baseline(args.workload, args.cpu_seconds)
    import torch
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit('bench.py: no HIP device visible')
    if world > 1 and args.backend == 'nccl' and ndev < world:
        raise SystemExit('bench.py: %d ranks over RCCL need %d GPUs, %d visible' % (world, world, ndev))
    local_dev = local_rank % ndev                                     # (gloo rehearsals put several ranks on one GPU)
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

    n_steps_launched = [0]                          # (steps launched so far: which dispatches of a kernel trace are the timed ones)

    def step():
        n_steps_launched[0] += 1
        tm.forward_device(Xs, N, coef=coef, Z=Z)
        tm.inverse_device(Z, N, coef=coef, X=Xinv)

    def step_uncached():
        # what a caller pays who changes the coefficients between calls (and what the reference does on every
        # inverse_map, TM:4047-4058): pack + fold + U-form build + table build + index, then the two lookups
        tm._pack_memo = None                         # (the class keeps the packed vector of unchanged coefficients)
        c = tm._pack_coeffs()
        tm.forward_device(Xs, N, coef=c, Z=Z)
        tm.inverse_device(Z, N, coef=c, X=Xinv)

    unc_prev = [None]

    def step_uncached_deferred():
        # the same with `deferred_checks`: the two per-vector checks (spline fit errors, table sortedness) are read ONE STEP
        # LATER from pinned copies made in stream order (transport_map.validate) instead of behind a synchronisation in
        # front of the lookups; a failed check would mean computing that step again (it never fails for this map)
        tm._pack_memo = None
        c = tm._pack_coeffs()
        tm.forward_device(Xs, N, coef=c, Z=Z)
        tm.inverse_device(Z, N, coef=c, X=Xinv)
        if unc_prev[0] is not None and not tm.validate(unc_prev[0]):
            raise RuntimeError('deferred check failed')
        unc_prev[0] = c

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, n):
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        sync()
        return time.perf_counter() - t0

    # (a) the contract's W + K steps straight from the cold start (no pre-warm): cold_ms_per_step
    step()                                           # (builds the tables kept with `coef`; first-launch costs)
    for _ in range(args.warmup):
        step()
    cold = timed(step, args.steps)
    # (b) After any idle gap of a millisecond or more (and from a cold start) the chip runs this workload 10-30 % slower
    # for the next ~50 launches (20 ms) before it holds its clock again (tools/ramp_probe.py).  For the headline the
    # card is first kept busy with the same steps, untimed, for --prewarm-seconds, the W warm-up steps follow
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
            # the ranks leave the time-based loop up to one batch apart; align them once, then give every rank the same
            # number of steps, so that all arrive at the barrier in front of the timed region together
            sync()
            for _ in range(150):
                step()
            prewarm_steps += 150
    for _ in range(args.warmup):
        step()
    timed_first_step = n_steps_launched[0]
    elapsed = timed(step, args.steps)
    # per-kernel timing with HIP events on the launch stream, directly behind the timed steps (same clock state)
    lib = tm._lib
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
    # The event times are used AS MEASURED (round 4 took the time of an empty event pair off each: with a kernel between the
    # records that latency overlaps the dispatch - the raw times of the two launches already sum to the timed step)
    fwd_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    inv_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_inv]))
    tm.forward_device(Xs, N, coef=coef, Z=Z)
    fwd_kernel = lib.ttm_last_kernel().decode()
    tm.inverse_device(Z, N, coef=coef, X=Xinv)
    inv_kernel = lib.ttm_last_kernel().decode()
    # ... and each launch ALONE, replayed back to back from a captured HIP graph (no event records, no Python between the
    # launches; the other kernel's buffers are not touched in between - informational beside the alternating event times)
    fwd_graph_ms = inv_graph_ms = None
    if separable and world == 1:
        fwd_graph_ms = graph_ms(torch, lambda: tm.forward_device(Xs, N, coef=coef, Z=Z), launches=10, reps=10)
        inv_graph_ms = graph_ms(torch, lambda: tm.inverse_device(Z, N, coef=coef, X=Xinv), launches=10, reps=10)
        for _ in range(30):
            step()
    # (c) the un-cached step, same clock state
    n_unc = max(5, min(args.steps, 50))
    for _ in range(3):
        step_uncached()
    uncached = timed(step_uncached, n_unc)
    tm.deferred_checks = True
    for _ in range(3):
        step_uncached_deferred()
    uncached_deferred = timed(step_uncached_deferred, n_unc)
    tm.validate(unc_prev[0])
    tm.deferred_checks = False
    per_rank_ms = None
    if dist is not None:
        # every rank's own clock over the timed steps (a straggler shows here), then the contract's maximum over the ranks
        mine = torch.tensor([1e3 * elapsed / args.steps], dtype=torch.float64, device='cuda')
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [float(v.item()) for v in every]
        t = torch.tensor([elapsed, cold, uncached], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, cold, uncached = [float(v) for v in t.tolist()]
    # what a plain elementwise pass achieves on the same buffers in the same alternating pattern (X -> Z, Z -> Xinv: the
    # bytes of a step with no arithmetic), directly behind the timed steps: the card's own streaming floor for this step
    floor_ms = None
    if separable:
        def floor_step():
            torch.abs(Xs[:D], out=Z)
            torch.abs(Z, out=Xinv[:D])
        for _ in range(20):
            floor_step()
        evf = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(args.steps, 50))]
        for a, b in evf:
            a.record(); floor_step(); b.record()
        torch.cuda.synchronize()
        floor_ms = float(np.mean([a.elapsed_time(b) for a, b in evf]))
        for _ in range(20):                              # (the buffers hold the map's own results again for the checks below)
            step()
        torch.cuda.synchronize()
    # round trip sanity inside the bench: S^{-1}(S(x)) == x up to the inverse's own accuracy
    err = float((Xinv[:, :N] - Xs[:, :N]).abs().max().item())
    extra = {}
    if not separable:
        # the same inversion with the safeguarded Newton root search (an extension, SURVEY section 8a' K5) and the counted
        # fp64 rates of the three launches (profiles/fp64_counts.json)
        tm.root_finder = 'newton'
        for _ in range(2):
            tm.inverse_device(Z, N, coef=coef, X=Xinv)
        evn = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
        for a, b in evn:
            a.record(); tm.inverse_device(Z, N, coef=coef, X=Xinv); b.record()
        torch.cuda.synchronize()
        tm.root_finder = 'reference'
        extra['inverse_newton_ms'] = float(np.mean([a.elapsed_time(b) for a, b in evn]))
        extra['fp64_counted'] = counted_fp64(args.workload, N, fwd_ms, inv_ms, extra['inverse_newton_ms'])
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
        # the log-determinant-only pass (no map values written: what evaluate_pullback_density needs, 8 N (d + 1) bytes)
        for _ in range(20):
            tm.density_device(Xs, N, coef=coef, logdet=ld, sigma=sigma)
        for a, b in evp:
            a.record()
            tm.density_device(Xs, N, coef=coef, logdet=ld, sigma=sigma)
            b.record()
        torch.cuda.synchronize()
        extra['pullback_logdet_only_ms'] = float(np.mean([a.elapsed_time(b) for a, b in evp]))
        extra['pullback_logdet_only_frac'] = 8.0 * N * (du + 1) / (extra['pullback_logdet_only_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS
        # the same step replayed from a captured HIP graph (rank-local, informational - `value` is the stream-launched loop above): what
        # the gaps between the launches of the Python loop cost
        if args.workload == 'C5' and world == 1:
            try:
                g = graph_ms(torch, step, launches=10, reps=10)
                if g is not None:
                    extra['graph_replay_ms_per_step'] = g
            except Exception as exc:                   # noqa: BLE001
                extra['graph_replay_error'] = repr(exc)
    if world == 1:
        try:
            extra['objective_roofline'] = objective_roofline(torch, tm, args.workload)
        except Exception as exc:                       # noqa: BLE001
            extra['objective_roofline_error'] = repr(exc)
    if world == 1 and not args.no_optimize:
        # secondary metric of BASELINE.json: optimize() wall-clock on the resident ensemble (from coeffs_init)
        saved = ([c.copy() for c in tm.coeffs_mon], [c.copy() for c in tm.coeffs_nonmon])
        for k in range(D):
            tm.coeffs_mon[k] = tm.coeffs_mon[k] * 0 + tm.coeffs_init
            tm.coeffs_nonmon[k] = tm.coeffs_nonmon[k] * 0 + tm.coeffs_init
        def timed_optimize():
            for k in range(D):
                tm.coeffs_mon[k] = tm.coeffs_mon[k] * 0 + tm.coeffs_init
                tm.coeffs_nonmon[k] = tm.coeffs_nonmon[k] * 0 + tm.coeffs_init
            torch.cuda.synchronize()
            t0o = time.perf_counter()
            tm.optimize()
            torch.cuda.synchronize()
            return time.perf_counter() - t0o
        # the first call of a process also loads the code objects of the reduction kernels (~0.15 s, once)
        extra['optimize_first_call_s'] = timed_optimize()
        extra['optimize_s'] = timed_optimize()
        if separable:
            # what was just timed is checked, not only timed: per component the reduced separable objective (TM:2978-3018)
            # at the optimum found is at or below the objective of the reference-optimised coefficients of the fixture
            # evaluated on the SAME ensemble, with the engine's own A matrices
            worst = -np.inf
            for k in range(D):
                A, _ = tm.separable_setup(k)
                J_got = tm.separable_objective(tm.coeffs_mon[k], A, k)[0]
                J_ref = tm.separable_objective(saved[0][k], A, k)[0]
                worst = max(worst, (J_got - J_ref) / (1.0 + abs(J_ref)))
            extra['optimize_J_minus_J_reference_coefficients_max_rel'] = float(worst)
            if not worst <= 1e-8:
                raise RuntimeError('optimize(): objective above that of the reference coefficients by %.3e' % worst)
        tm.coeffs_mon, tm.coeffs_nonmon = saved
    if world == 1 and not args.no_api:
        # What the drop-in caller pays (the class takes and returns NumPy arrays, TM:12-368, 2391-2437, 3639-3796): the
        # constructor (H2D of X, moments, layout change, special-term quantiles), optimize() end to end from a host array
        # ("incl. one H2D of X"), and map() / inverse_map() NumPy in -> NumPy out, next to the one-way
        # PCIe time of the same bytes measured in this run.
        from triangular_transport_toolbox_amd.transport_map import transport_map
        api = {}
        pin = torch.empty((N, d), dtype=torch.float64, pin_memory=True)
        devbuf = torch.empty((N, d), dtype=torch.float64, device='cuda')
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize(); t0p = time.perf_counter()
            devbuf.copy_(pin, non_blocking=True)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0p)
        api['pcie_oneway_ms'] = 1e3 * best
        api['pcie_oneway_GBps'] = 8.0 * N * d / best / 1e9
        # ... and of both directions at once (what a call that takes N x d and returns N x D has to move)
        pin2 = torch.empty((N, d), dtype=torch.float64, pin_memory=True)
        dev2 = torch.empty((N, d), dtype=torch.float64, device='cuda')
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        best2 = 1e9
        for _ in range(4):
            torch.cuda.synchronize(); t0p = time.perf_counter()
            with torch.cuda.stream(sa):
                devbuf.copy_(pin, non_blocking=True)
            with torch.cuda.stream(sb):
                pin2.copy_(dev2, non_blocking=True)
            torch.cuda.synchronize(); best2 = min(best2, time.perf_counter() - t0p)
        api['pcie_duplex_ms'] = 1e3 * best2
        del pin, devbuf, pin2, dev2

        def best_of(fn, n=3):
            b, r = 1e9, None
            for _ in range(n):
                torch.cuda.synchronize(); t0a = time.perf_counter()
                r = fn()
                torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0a)
            return 1e3 * b, r
        tm.map(X[:4096])
        api['api_map_ms'], Zh = best_of(lambda: tm.map(X))
        api['api_inverse_map_ms'], Xh = best_of(lambda: tm.inverse_map(Zh))
        api['api_map_over_pcie_oneway'] = api['api_map_ms'] / api['pcie_oneway_ms']
        api['api_map_over_pcie_duplex'] = api['api_map_ms'] / api['pcie_duplex_ms']
        api['api_roundtrip_max_abs_err'] = float(np.max(np.abs(Xh - X[:, tm.skip_dimensions:])))
        tm.host_pipeline = False
        api['api_map_unpipelined_ms'], _ = best_of(lambda: tm.map(X), 2)
        api['api_inverse_map_unpipelined_ms'], _ = best_of(lambda: tm.inverse_map(Zh), 2)
        tm.host_pipeline = True
        del Zh, Xh

        def construct():
            return transport_map(X=X, monotone=cfg['monotone'], nonmonotone=cfg['nonmonotone'], verbose=False, **cfg['kwargs'])
        ctor_ms, tm_c = best_of(construct, 2)
        api['ctor_s'] = 1e-3 * ctor_ms
        del tm_c
        if not args.no_optimize:
            def end_to_end():
                t = construct()
                t.optimize()
                return t
            e2e_ms, tm_e = best_of(end_to_end, 2)
            api['optimize_end_to_end_s'] = 1e-3 * e2e_ms
            api['optimize_end_to_end'] = 'transport_map(X) from the host array (H2D, moments, layout change, special-term ' \
                                         'quantiles) + optimize() of all components, N = %d' % N
            del tm_e
        extra.update(api)
    if world == 1 and not args.no_other_configs and args.workload == 'C5':
        try:
            extra['other_configs'] = other_configs(torch, ['C2b', 'C2a', 'C3', 'C3int', 'C5int', 'EX01', 'EX03'])
        except Exception as exc:                       # noqa: BLE001  (never fatal for the headline line)
            extra['other_configs_error'] = repr(exc)
        try:
            extra['other_configs']['C4'] = entf_config(torch)
        except Exception as exc:                       # noqa: BLE001
            extra['other_configs_C4_error'] = repr(exc)
        try:
            extra['other_configs']['C4_block_map'] = ents_block_config(torch)
        except Exception as exc:                       # noqa: BLE001
            extra['other_configs_C4_block_map_error'] = repr(exc)
    fl_pre = flops_per_eval(tm) if separable else None
    out = None
    if rank == 0:
        fwd_bytes = 8.0 * N * (du + D)
        inv_bytes = 8.0 * N * (2 * D)
        ms_step = 1e3 * elapsed / args.steps
        gbps = lambda nbytes, ms: nbytes / (ms * 1e-3) / 1e9          # noqa: E731
        traffic = {}
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(args.workload, {})
        # The event-timed launches of the alternating steps sum to the step they are part of; should they ever sum to LESS than
        # the wall-clock step (launch gaps, a slower clock in the timed loop), every kernel time is stretched by the same factor
        # so that forward + inverse = the timed step: roofline fractions never exceed what the driver's own clock supports
        kernel_sum_over_step = (fwd_ms + inv_ms) / ms_step
        stretch = max(1.0, 1.0 / kernel_sum_over_step)
        fwd_ev_ms, inv_ev_ms = fwd_ms, inv_ms
        fwd_ms, inv_ms = fwd_ms * stretch, inv_ms * stretch
        dominant = 'inverse' if inv_ms >= fwd_ms else 'forward'
        dom_bytes, dom_ms, dom_kernel = (inv_bytes, inv_ms, inv_kernel) if dominant == 'inverse' else (fwd_bytes, fwd_ms, fwd_kernel)
        achieved = gbps(dom_bytes, dom_ms)
        # rocprofv3 --kernel-trace averages of the TIMED launches of this command (tools/rocprof_timed.py picks the dispatches
        # [first_timed_step, first_timed_step + K) of each kernel out of the trace; committed under profiles/)
        rocprof = None
        rpath = os.path.join(ROOT, 'profiles', 'r05_timed_launches.json')
        if os.path.exists(rpath):
            rp = json.load(open(rpath)).get(args.workload)
            if rp:
                rocprof = {'source': 'profiles/r05_timed_launches.json', 'forward_avg_ms': rp.get('forward_avg_ms'),
                           'inverse_avg_ms': rp.get('inverse_avg_ms'), 'steps': rp.get('steps'),
                           'forward_frac': (gbps(fwd_bytes, rp['forward_avg_ms']) / HBM_PEAK_GBS if rp.get('forward_avg_ms') else None),
                           'inverse_frac': (gbps(inv_bytes, rp['inverse_avg_ms']) / HBM_PEAK_GBS if rp.get('inverse_avg_ms') else None),
                           'note': 'another run (another box) of the same command under the profiler; profiled dispatches run a few % slower'}
        out = {
            'metric': 'map-evals/sec (forward+inverse, N samples x D comps)',
            'value': world * N * D * args.steps / elapsed,
            'unit': 'map-evals/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': ms_step,
            'per_rank_ms_per_step': per_rank_ms,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': args.workload + ': ' + WORKLOADS[args.workload][3],
                       'N_per_gpu': N, 'D': D, 'columns': d, 'inverse': 'table' if separable else 'bisection',
                       'layout': 'column-major resident in HBM', 'coefficients': 'reference-optimised (tests/golden)',
                       'world_size': (dist.get_world_size() if dist is not None else 1),
                       'backend': (('rccl' if args.backend == 'nccl' else args.backend) if dist is not None else None)},
            # the dominant kernel = the one with the longer measured launch
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic.get("%s_hbm_bytes_per_launch" % dom_kernel.split("<")[0]),
                         'traffic_other_kernel': traffic.get("%s_hbm_bytes_per_launch" % (fwd_kernel if dominant == 'inverse' else inv_kernel).split("<")[0]),
                         'kernel': dom_kernel, 'which': dominant,
                         'algorithmic_bytes_per_launch': dom_bytes, 'avg_launch_ms': dom_ms,
                         'rocprof_timed_avg_launch_ms': ((rocprof or {}).get('%s_avg_ms' % dominant)),
                         'forward_frac': gbps(fwd_bytes, fwd_ms) / HBM_PEAK_GBS, 'forward_kernel': fwd_kernel,
                         'inverse_frac': gbps(inv_bytes, inv_ms) / HBM_PEAK_GBS, 'inverse_kernel': inv_kernel,
                         # forward + inverse as one unit of work against the same roof: by kernel time and by the
                         # wall-clock step (launch gaps included)
                         'pair_frac': gbps(fwd_bytes + inv_bytes, fwd_ms + inv_ms) / HBM_PEAK_GBS,
                         'pair_frac_wallclock': gbps(fwd_bytes + inv_bytes, ms_step) / HBM_PEAK_GBS,
                         # the same against what a copy achieves: the guide's measured float4 copy (6.29 TB/s) and an
                         # elementwise pass over this step's own buffers, measured in this run (bench.py: floor_step)
                         'copy_peak': {'guide_GBps': COPY_PEAK_GBS, 'frac': achieved / COPY_PEAK_GBS,
                                       'pair_frac_wallclock': gbps(fwd_bytes + inv_bytes, ms_step) / COPY_PEAK_GBS,
                                       'measured_elementwise_pair_ms': floor_ms,
                                       'measured_GBps': (gbps(fwd_bytes + inv_bytes, floor_ms) if floor_ms else None),
                                       'pair_frac_of_measured': (floor_ms / ms_step if floor_ms else None)}},
            'forward_ms': fwd_ms, 'inverse_ms': inv_ms,
            'kernel_timing': {'method': 'HIP events on the launch stream around every launch of K alternating steps, directly behind '
                                        'the timed steps, used as measured (no overhead taken off); stretched so that forward + '
                                        'inverse >= the timed step',
                              'forward_event_ms': fwd_ev_ms, 'inverse_event_ms': inv_ev_ms,
                              'kernel_sum_over_timed_step': kernel_sum_over_step, 'stretch': stretch,
                              'forward_alone_graph_replay_ms': fwd_graph_ms, 'inverse_alone_graph_replay_ms': inv_graph_ms,
                              'timed_launches': {'first_step': timed_first_step, 'steps': args.steps,
                                                 'what': 'index of the first timed step among the steps of this process: dispatch '
                                                         'i of the forward / inverse kernel in a kernel trace of this command'},
                              'rocprof_timed': rocprof},
            'forward_GBps_algorithmic': gbps(fwd_bytes, fwd_ms),
            'inverse_GBps_algorithmic': gbps(inv_bytes, inv_ms),
            'roundtrip_max_abs_err': err,
            # the same K steps behind W warm-up steps only, from the cold start (no pre-warm)
            'cold_ms_per_step': 1e3 * cold / args.steps,
            'cold_value': world * N * D * args.steps / cold,
            # coefficient-dependent precompute (fold, U-form splines, 1001-point tables + index) is cached per
            # coefficient vector and is NOT inside `value`; a step that redoes it every time:
            'uncached_ms_per_step': 1e3 * uncached / n_unc,
            'setup_us': 1e3 * (1e3 * uncached / n_unc - ms_step),
            # ... with the per-vector checks read one step later (transport_map.deferred_checks / validate())
            'uncached_deferred_checks_ms_per_step': 1e3 * uncached_deferred / n_unc,
            'setup_deferred_checks_us': 1e3 * (1e3 * uncached_deferred / n_unc - ms_step),
            'cached': 'folded coefficients, U-form section and inverse tables are functions of the coefficient vector '
                      'and are built once before the timed region; uncached_ms_per_step rebuilds them every step '
                      '(the reference rebuilds its table in every inverse_map, TM:4047-4058)',
            'prewarm': {'seconds': args.prewarm_seconds, 'steps': prewarm_steps,
                        'why': 'untimed steps before the warm-up so that the timed steps run at the clock the chip holds under '
                               'sustained load (after an idle gap >= 1 ms the next ~50 launches run 10-30 % slower); '
                               'cold_ms_per_step is the same measurement without it'},
        }
        fl = fl_pre
        if fl is not None:
            tf = lambda f, ms: f * N * D / (ms * 1e-3) / 1e12          # noqa: E731
            out['fp64'] = {'peak_TFLOPs': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                           'forward': {'flop_per_eval': fl[0], 'achieved': tf(fl[0], fwd_ms), 'frac': tf(fl[0], fwd_ms) / FP64_PEAK_TFLOPS},
                           'inverse': {'flop_per_eval': fl[1], 'achieved': tf(fl[1], inv_ms), 'frac': tf(fl[1], inv_ms) / FP64_PEAK_TFLOPS},
                           'note': 'arithmetic the kernels execute per component evaluation (FMA = 2 flop, bench.py:flops_per_eval), '
                                   'vector fp64 peak of MI355X_MICROARCH.md; the measured VALU instruction counts are in '
                                   'profiles/*_pmc_summary.json'}
        out.update(extra)
        if cpu is not None:
            out['cpu_baseline'] = cpu
    if world > 1:
        # The node-level extras (strong scaling, the sample-sharded filter, the component-partitioned optimize()) are the
        # first code of this repository that runs RCCL collectives of its own over more than one rank: they run behind the
        # headline, under a watchdog - if they have not returned within --extras-timeout seconds rank 0 prints the line
        # it has (with `multi_gpu_extras_error`) and every rank leaves, instead of a hung collective taking the headline
        # with it.  (A thread can do that: the blocked calls have released the GIL.)
        import threading
        # ... and should a rank DIE inside them (an abort inside a collective cannot be caught), the guardian started at the
        # top of main() already holds the headline line: it prints the last line it was sent when rank 0's pipe closes
        if rank == 0 and out is not None:
            emit(dict(out, multi_gpu_extras_error='the process ended inside the multi-GPU extras'))

        def give_up():
            if rank == 0 and out is not None:
                out['multi_gpu_extras_error'] = 'not finished within %g s' % args.extras_timeout
                emit(out, last=True)
            else:
                time.sleep(2.0)
            os._exit(3)                                  # (a hung collective is a failed run: the line says why)
        dog = threading.Timer(args.extras_timeout, give_up)
        dog.daemon = True
        dog.start()
        try:
            extra2 = multi_gpu_extras(torch, dist, args, rank, world, args.backend)
            if not args.no_optimize:
                extra2.update(component_sharded_optimize(torch, dist, args))
        except Exception as exc:                       # noqa: BLE001
            extra2 = {'multi_gpu_extras_error': repr(exc)}
        dog.cancel()
        if out is not None:
            out.update(extra2)
    if out is not None:
        emit(out, last=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
