"""
Ensemble Transport Filter / Smoother harnesses (the callers of the hot path in Examples C: the filter of
example_06.py:252-328 and the backward smoother of example_07.py:424-465).

One assimilation cycle = three one-observation-at-a-time composite-map updates
(`reset -> optimize -> map -> inverse_map` with the observed value as X_star) followed by an RK4
Lorenz-63 forecast.  `assimilate` / `smooth` take and return host NumPy ensembles like the reference's loops (the
transport-map work runs on the GPU through the drop-in class).  `Filter` is the same filter with the ensemble resident
on the device: forecast, observation noise, assembly of the map input, reset, optimisation, pushforward and conditional
inverse all run there; per update only a few dozen scalars (column moments, order statistics, coefficients) visit the
host.
"""
import copy
import ctypes

import numpy as np

from . import specs

PERMUTATIONS = [[0, 1, 2], [1, 0, 2], [2, 1, 0]]      # example_06.py:272


def lorenz_dynamics(Z, beta=8 / 3, rho=28, sigma=10):
    """example_06.py:27-44."""
    return np.column_stack((-sigma * Z[..., 0] + sigma * Z[..., 1],
                            -Z[..., 0] * Z[..., 2] + rho * Z[..., 0] - Z[..., 1],
                            Z[..., 0] * Z[..., 1] - beta * Z[..., 2]))


def rk4(Z, dt, nt):
    """example_06.py:47-76 for an N x 3 ensemble."""
    Z = np.array(Z, dtype=float, copy=True)
    for _ in range(nt):
        k1 = lorenz_dynamics(Z)
        k2 = lorenz_dynamics(Z + dt / 2 * k1)
        k3 = lorenz_dynamics(Z + dt / 2 * k2)
        k4 = lorenz_dynamics(Z + dt * k3)
        Z += dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
    return Z


def rk4_point(z, dt, nt, beta=8 / 3, rho=28, sigma=10):
    """rk4() for ONE state on plain floats - the truth of a benchmark run (the same IEEE operations in the same order as the
    array expressions above, without NumPy's per-call cost: 8 against 80 us per cycle of the filter benchmark)."""
    def f(x, y, w):
        return -sigma * x + sigma * y, -x * w + rho * x - y, x * y - beta * w
    x, y, w = float(z[0]), float(z[1]), float(z[2])
    for _ in range(nt):
        a1, b1, c1 = f(x, y, w)
        a2, b2, c2 = f(x + dt / 2 * a1, y + dt / 2 * b1, w + dt / 2 * c1)
        a3, b3, c3 = f(x + dt / 2 * a2, y + dt / 2 * b2, w + dt / 2 * c2)
        a4, b4, c4 = f(x + dt * a3, y + dt * b3, w + dt * c3)
        x += dt / 6 * (a1 + 2 * a2 + 2 * a3 + a4)
        y += dt / 6 * (b1 + 2 * b2 + 2 * b3 + b4)
        w += dt / 6 * (c1 + 2 * c2 + 2 * c3 + c4)
    return np.array([x, y, w])


def make_filter_map(N, maxorder=3, lmbda=0.05, rng=None, **kwargs):
    """The 4-column filtering map of example_06.py:186-231 (dummy samples until the first reset)."""
    from .transport_map import transport_map
    mon, non = specs.entf_filter_spec(maxorder)
    rng = np.random.default_rng(0) if rng is None else rng
    return transport_map(monotone=mon, nonmonotone=non, X=rng.uniform(size=(N, 4)),
                         polynomial_type='hermite function', monotonicity='separable monotonicity',
                         regularization='l2', regularization_lambda=lmbda, verbose=False, **kwargs)


def assimilate(tm, ensemble, observation, noises):
    """The three updates of one cycle (example_06.py:272-317).
    ensemble: N x 3 forecast; observation: 3 observed values; noises[idx]: N observation-noise draws."""
    Xa = np.array(ensemble, dtype=float, copy=True)
    N = Xa.shape[0]
    for idx, perm in enumerate(PERMUTATIONS):
        Yt = Xa[:, idx] + noises[idx]
        map_input = np.column_stack((Yt[:, np.newaxis], Xa[:, perm]))
        tm.reset(copy.copy(map_input))
        tm.optimize()
        Z_pushforward = tm.map(map_input)
        Y_star = np.repeat(np.asarray(observation[idx]).reshape((1, 1)), N, axis=0)
        ret = tm.inverse_map(X_star=Y_star, Z=Z_pushforward)
        Xa = ret[:, perm]
    return Xa


def make_smoother_map(N, maxorder=3, lmbda=0.05, D=3, rng=None, **kwargs):
    """The 2D-column backward-smoothing map of example_07.py:368-408 (dummy samples until the first reset)."""
    from .transport_map import transport_map
    mon, non = specs.ents_smoother_spec(maxorder, D)
    rng = np.random.default_rng(0) if rng is None else rng
    return transport_map(monotone=mon, nonmonotone=non, X=rng.uniform(size=(N, 2 * D)),
                         polynomial_type="probabilist's hermite", monotonicity='separable monotonicity',
                         regularization='l2', regularization_lambda=lmbda, verbose=False, **kwargs)


def smooth(tm, forecasts, analyses):
    """Backward pass of the Ensemble Transport Smoother (example_07.py:424-465).
    forecasts[t], analyses[t]: N x D filtering forecast / analysis ensembles of time t (t = 0..T-1).  Returns the
    smoothing ensembles, T x N x D: the last one is the last filtering analysis; going backward, the joint ensemble
    (forecast at t+1, analysis at t) is mapped and conditioned on the smoothing samples of t+1
    (reset -> optimize -> map -> inverse_map with X_star)."""
    analyses = np.asarray(analyses, dtype=float)
    Xs = np.array(analyses, copy=True)
    for t in range(len(analyses) - 2, -1, -1):
        map_input = copy.copy(np.column_stack((forecasts[t + 1], analyses[t])))
        tm.reset(copy.copy(map_input))
        tm.optimize()
        Z_pushforward = tm.map(map_input)
        Xs[t] = tm.inverse_map(X_star=copy.copy(Xs[t + 1]), Z=Z_pushforward)
    return Xs


class Filter:
    """Device-resident Ensemble Transport Filter for Lorenz-63 (example_06.py:252-328).

    The ensemble is a column-major 3 x ld device matrix for its whole life.  One cycle = `forecast` (RK4,
    ttm_lorenz63_rk4) + `assimilate` (three one-observation-at-a-time updates).  An update: y_t = x_idx + noise
    (ttm_perturb) -> map input [y_t | x_perm] (ttm_map_columns) -> reset_device -> optimize (native L-BFGS-B) ->
    pushforward of the resident samples (ttm_forward) -> conditional inverse with the observed value in the first
    column (ttm_inverse_table) -> back to physical units and ensemble order (ttm_map_columns)."""

    def __init__(self, N, maxorder=3, lmbda=0.05, seed=0, obs_sd=2.0, row0=0, **kwargs):
        # A sample-sharded filter (one rank per GPU, `shard_samples=True` in kwargs) holds N rows of the ensemble starting
        # at global row `row0`: the observation noise is a function of (seed, draw, GLOBAL row), so the sharded filter
        # draws what the single-rank filter draws; everything else that couples the shards is the class's reductions
        # (column moments, order statistics, objective / gradient sums - example_06.py:252-328 has no counterpart).
        self.N = int(N)
        self.row0 = int(row0)
        self.tm = make_filter_map(self.N, maxorder, lmbda, **kwargs)
        self.seed = int(seed)
        self.obs_sd = float(obs_sd)
        self.ens = self.tm._cols(3, self.N, zero=True)           # the ensemble (device)
        self._inp = self.tm._cols(4, self.N, zero=True)          # map input [y | x_perm]
        self._draws = 0                                          # stream counter of the noise generator

    # -- host <-> device, only at the boundaries of a run
    def set_ensemble(self, ensemble):
        import torch
        E = np.ascontiguousarray(np.asarray(ensemble, dtype=float).T)
        self.ens[:, :self.N].copy_(torch.from_numpy(E))

    def ensemble(self):
        return self.ens[:, :self.N].cpu().numpy().T.copy()

    def forecast(self, dt=0.05, nt=2):
        tm = self.tm
        _check(tm._lib.ttm_lorenz63_rk4(tm._ptr(self.ens), self.ens.shape[1], self.N, float(dt), int(nt), tm._stream()))

    def assimilate(self, observation, noises=None):
        """The three updates of one cycle.  noises: optional 3 x N device tensor (or host array) of observation-noise
        draws to ADD to the observed component (the parity tests pass the reference's own draws); default: fresh
        N(0, obs_sd^2) deviates of the counter-based generator."""
        import torch
        tm, N, ens, inp = self.tm, self.N, self.ens, self._inp
        ld = ens.shape[1]
        if noises is not None and not isinstance(noises, torch.Tensor):
            noises = tm._to_dev(np.ascontiguousarray(np.asarray(noises, dtype=float)))
        for idx, perm in enumerate(PERMUTATIONS):
            # y_t = x_idx + noise into column 0 of the map input; columns 1..3 = the ensemble in `perm` order
            tm.map_columns([-1] + list(perm), 4, N, source=ens, out=inp)
            col = ctypes.c_void_p(ens.data_ptr() + 8 * idx * ld)
            if noises is not None:
                _check(tm._lib.ttm_perturb(col, tm._ptr(noises, idx * noises.shape[1]), 1.0, 0, 0, 0, N, tm._ptr(inp), tm._stream()))
            else:
                self._draws += 1
                _check(tm._lib.ttm_perturb(col, None, self.obs_sd, self.seed, self._draws, self.row0, N, tm._ptr(inp), tm._stream()))
            tm.reset_device(inp, N)
            tm.optimize()
            Z = tm.forward_device(tm._Xs, N)
            # conditioning column: the observed value, standardised like the first column of the training samples
            ystar = (float(observation[idx]) - tm.X_mean[0]) / tm.X_std[0] if tm.standardize_samples else float(observation[idx])
            Xc = tm.map_columns([-1, -1, -1, -1], 4, N, shift=[ystar, 0.0, 0.0, 0.0])
            tm.inverse_device(Z, N, X=Xc)
            # back to physical units; ensemble column j is the map's column 1 + perm[j]  (Xa = ret[:, perm])
            src = [1 + p for p in perm]
            scale = [tm.X_std[c] for c in src] if tm.standardize_samples else None
            shift = [tm.X_mean[c] for c in src] if tm.standardize_samples else None
            tm.map_columns(src, 3, N, source=Xc, scale=scale, shift=shift, out=ens)

    def benchmark(self, ensemble, truth, cycles, warmup=3):
        """`cycles` full cycles (truth forecast and observation on the host: 3 numbers; everything of size N on the
        device), timed end to end."""
        import time
        import torch
        rng = np.random.default_rng(self.seed)
        self.set_ensemble(ensemble)
        truth = np.array(truth, dtype=float)

        def cycle(truth):
            truth = rk4_point(truth, 0.05, 2)
            obs = truth + self.obs_sd * rng.standard_normal(3)
            self.forecast(0.05, 2)
            self.assimilate(obs)
            return truth
        dist = self.tm._dist()

        def sync():
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
        for _ in range(warmup):
            truth = cycle(truth)
        sync()
        t0 = time.perf_counter()
        for _ in range(cycles):
            truth = cycle(truth)
        sync()
        el = time.perf_counter() - t0
        if dist is not None:                                      # (sharded: the ensemble mean over all ranks, the slowest rank's clock)
            acc = torch.cat((self.ens[:, :self.N].sum(dim=1), torch.tensor([float(self.N), el], dtype=torch.float64, device=self.ens.device)))
            tmax = torch.tensor([el], dtype=torch.float64, device=self.ens.device)
            dist.all_reduce(acc)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            mean = (acc[:3] / acc[3]).cpu().numpy()
            el = float(tmax.item())
            return dict(workload='C4: Lorenz-63 EnTF, Example-06 map, ensemble SAMPLE-SHARDED over the ranks (device resident)',
                        N_total=int(acc[3].item()), N_per_rank=self.N, cycles=cycles, ms_per_cycle=1e3 * el / cycles, updates_per_cycle=3,
                        rmse_last=float(np.sqrt(np.mean((mean - truth) ** 2))))
        mean = self.ens[:, :self.N].mean(dim=1).cpu().numpy()
        return dict(workload='C4: Lorenz-63 EnTF, Example-06 map (4 columns, D = 3, order 3, L2 0.05), ensemble resident on the device',
                    N=self.N, cycles=cycles, ms_per_cycle=1e3 * el / cycles, updates_per_cycle=3,
                    rmse_last=float(np.sqrt(np.mean((mean - truth) ** 2))))


def _check(rc):
    from . import _capi
    _capi.check(rc)
