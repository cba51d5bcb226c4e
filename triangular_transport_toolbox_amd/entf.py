"""
Ensemble Transport Filter / Smoother harnesses (the callers of the hot path in Examples C: the filter of
example_06.py:252-328 and the backward smoother of example_07.py:424-465).

One assimilation cycle = three one-observation-at-a-time composite-map updates
(`reset -> optimize -> map -> inverse_map` with the observed value as X_star) followed by an RK4
Lorenz-63 forecast.  The transport-map work runs on the GPU through the drop-in class; the forecast and the
observation-noise draws are O(N) host NumPy as in the reference (a device-resident forecast is listed as next
in DESIGN.md).
"""
import copy

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
