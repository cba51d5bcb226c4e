"""
Map specifications and synthetic ensembles for the benchmark configurations
C1..C5 of BASELINE.json / SURVEY.md section 8.

The reference has no configuration system: every example script assembles its
``monotone`` / ``nonmonotone`` lists by hand.  The helpers below emit the same
lists in the same order (coefficient vectors are positional, so the order is
part of the contract):

* spiral (C1/C2a)      - Examples A/Example 01/example_01.py:121-170
* temperature (C2b/C3) - Examples B/Example 03/example_03.py:108-159
* EnTF filter map (C4) - Examples C/Example 06/example_06.py:186-214
* EnTS smoother map    - Examples C/Example 07/example_07.py:368-392
* banded d=40 (C5)     - BASELINE.md section 3 (synthetic, no reference script)

Only NumPy/SciPy host code lives here; nothing in this file touches the GPU.
"""

import itertools

import numpy as np


# ---------------------------------------------------------------------------
# map specifications
# ---------------------------------------------------------------------------

def spiral_spec(maxorder, D=2):
    """Full map with Hermite-function cross terms (example_01.py:121-170).

    nonmonotone[k] = [[], [k-1]*1+['HF'], ..., [k-1]*maxorder+['HF']] (k>0)
    monotone[k]    = all multi-indices over range(k+1) of total order
                     1..maxorder that contain k, each with 'HF' appended, in
                     itertools.combinations_with_replacement order.
    """
    monotone, nonmonotone = [], []
    for k in range(D):
        monotone.append([])
        nonmonotone.append([[]])
        for order in range(maxorder):
            if k > 0:
                nonmonotone[-1].append([k - 1] * (order + 1) + ['HF'])
            for entry in itertools.combinations_with_replacement(range(k + 1), order + 1):
                if k in entry:
                    monotone[-1].append([int(e) for e in entry] + ['HF'])
    return monotone, nonmonotone


def banded_integrated_spec(D, maxorder, band):
    """Integrated-rectifier map whose component k sees columns k-band..k.

    Monotone terms: all HF multi-indices of total order <= maxorder over the
    band that contain k; nonmonotone: constant, linear and HF terms of each
    lower band column.
    """
    monotone, nonmonotone = [], []
    for k in range(D):
        lo = max(0, k - band)
        monotone.append([])
        nonmonotone.append([[]])
        for j in range(lo, k):
            nonmonotone[-1].append([j])
            for order in range(2, maxorder + 1):
                nonmonotone[-1].append([j] * order + ['HF'])
        for order in range(1, maxorder + 1):
            for entry in itertools.combinations_with_replacement(range(lo, k + 1), order):
                if k in entry:
                    monotone[-1].append([int(e) for e in entry] + ['HF'])
    return monotone, nonmonotone


def temperature_spec(maxorder, D=2):
    """Separable map of example_03.py:108-159: LET + (maxorder-1) iRBF + RET."""
    monotone, nonmonotone = [], []
    for k in range(D):
        monotone.append([])
        nonmonotone.append([[]])
        for order in range(maxorder):
            if k > 0:
                nonmonotone[-1].append([k - 1] * (order + 1) + ['HF'])
        if maxorder == 1:
            monotone[-1].append([k])
        else:
            monotone[-1].append('LET ' + str(k))
            for _ in range(maxorder - 1):
                monotone[-1].append('iRBF ' + str(k))
            monotone[-1].append('RET ' + str(k))
    return monotone, nonmonotone


def dense_separable_spec(D, maxorder):
    """C3: separable, monotone LET + (maxorder-2) iRBF + RET (m_mon = maxorder+1
    for maxorder=4 -> 5), dense HF nonmonotone part over all lower columns
    (m_nm = 1 + maxorder*k)."""
    monotone, nonmonotone = [], []
    for k in range(D):
        nonmonotone.append([[]])
        for j in range(k):
            for order in range(1, maxorder + 1):
                nonmonotone[-1].append([j] * order + ['HF'])
        monotone.append(['LET ' + str(k)] + ['iRBF ' + str(k)] * (maxorder - 1) + ['RET ' + str(k)])
    return monotone, nonmonotone


def banded_separable_spec(D, band=2):
    """C5: per component k, nonmonotone
    [], [j], [j,j,'HF'], [j,j,j,'HF'] for j in {k-band..k-1}; monotone
    LET k, iRBF k, iRBF k, RET k."""
    monotone, nonmonotone = [], []
    for k in range(D):
        nonmonotone.append([[]])
        for j in range(max(0, k - band), k):
            nonmonotone[-1].append([j])
            nonmonotone[-1].append([j, j, 'HF'])
            nonmonotone[-1].append([j, j, j, 'HF'])
        monotone.append(['LET ' + str(k), 'iRBF ' + str(k), 'iRBF ' + str(k), 'RET ' + str(k)])
    return monotone, nonmonotone


def density_example_spec(maxorder=3, D=2):
    """The separable map of example_05.py:78-110: monotone [k] + (maxorder-1) iRBF k; nonmonotone [], [k-1] and
    Hermite-function powers of x_{k-1} up to `maxorder`."""
    monotone, nonmonotone = [], []
    for k in range(D):
        monotone.append([[k]] + ['iRBF ' + str(k)] * (maxorder - 1))
        nonmonotone.append([[]])
        if k > 0:
            nonmonotone[-1].append([k - 1])
            for o in range(1, maxorder):
                nonmonotone[-1].append([k - 1] * (o + 1) + ['HF'])
    return monotone, nonmonotone


def entf_filter_spec(maxorder):
    """The 4-column filtering map of example_06.py:186-214 (X is N x 4, D = 3,
    skip_dimensions = 1)."""
    if maxorder == 1:
        nonmonotone = [[[], [0]], [[], [1]], [[], [1], [2]]]
        monotone = [[[1]], [[2]], [[3]]]
    else:
        orders = range(1, maxorder + 1)
        nonmonotone = [
            [[], [0]] + [[0] * od + ['HF'] for od in orders],
            [[], [1]] + [[1] * od + ['HF'] for od in orders],
            [[], [1]] + [[1] * od + ['HF'] for od in orders] + [[2]] + [[2] * od + ['HF'] for od in orders]]
        monotone = [
            ['LET 1'] + ['iRBF 1'] * (maxorder - 1) + ['RET 1'],
            [[2]],
            [[3]]]
    return monotone, nonmonotone


def ents_smoother_spec(maxorder, D=3):
    """The 2D-column backward-smoothing map of example_07.py:368-392 (X is N x 2D: forecast at s+1, analysis at s;
    skip_dimensions = D): dense nonmonotone blocks over all earlier columns, linear monotone terms."""
    def block(cols):
        out = [[]]
        for j in cols:
            out.append([j])
            if maxorder > 1:
                out += [[j] * od + ['HF'] for od in range(1, maxorder + 1)]
        return out
    nonmonotone = [block(range(D + k)) for k in range(D)]
    monotone = [[[D + k]] for k in range(D)]
    return monotone, nonmonotone


# ---------------------------------------------------------------------------
# synthetic ensembles (seeds fixed)
# ---------------------------------------------------------------------------

def sample_spiral(N, seed=0):
    """The spiral target of example_01.py:31-57, drawn after np.random.seed(seed)
    with the same sequence of RNG calls (beta(4,3) then standard normal)."""
    import scipy.stats
    np.random.seed(seed)
    seeds = scipy.stats.beta.rvs(a=4, b=3, size=N) * 3 * np.pi - np.pi
    vals = (seeds + np.pi) / (3 * np.pi) * 6 - 3
    X = np.column_stack((np.cos(seeds), np.sin(seeds))) * \
        ((1 + seeds + np.pi) / (3 * np.pi) * 5)[:, np.newaxis]
    X += np.column_stack([np.cos(seeds), np.sin(seeds)]) * \
        (scipy.stats.norm.rvs(size=N) * scipy.stats.norm.pdf(vals))[:, np.newaxis]
    return X / 2


def sample_banana(N, d=4, seed=0):
    """C3 banana chain: x_k <- 0.6 e_k + 0.8 (x_{k-1}^2 - 1)/sqrt(2) (k odd),
    0.7 e_k + sin(1.3 x_{k-1}) (k even)."""
    np.random.seed(seed)
    X = np.random.randn(N, d)
    for k in range(1, d):
        if k % 2 == 1:
            X[:, k] = 0.6 * X[:, k] + 0.8 * (X[:, k - 1] ** 2 - 1) / np.sqrt(2)
        else:
            X[:, k] = 0.7 * X[:, k] + np.sin(1.3 * X[:, k - 1])
    return X


def sample_mixture(N, d=40, seed=12345):
    """C5: 3-component AR(1) Gaussian mixture (generator in BASELINE.md section 3)."""
    rng = np.random.default_rng(seed)
    w = np.array([0.5, 0.3, 0.2])
    rho = np.array([0.5, -0.3, 0.7])
    s = np.array([1.0, 0.6, 0.8])
    amp = np.array([0.0, 2.0, -1.5])
    lab = rng.choice(3, size=N, p=w)
    eps = rng.standard_normal((N, d))
    M = amp[lab][:, None] * np.cos(0.7 * np.arange(d))[None, :]
    X = np.empty((N, d))
    X[:, 0] = s[lab] * eps[:, 0]
    for i in range(1, d):
        X[:, i] = rho[lab] * X[:, i - 1] + np.sqrt(1 - rho[lab] ** 2) * s[lab] * eps[:, i]
    X += M
    return X


def sample_wavy(N, seed=0):
    """The "wavy" target of example_05.py:22-38 (a sine ridge with beta(2,2) marginal along x_0), drawn after
    np.random.seed(seed) with the same sequence of RNG calls (beta then normal)."""
    import scipy.stats
    np.random.seed(seed)
    x0 = (scipy.stats.beta.rvs(a=2, b=2, size=N) * 2 - 1) * 3
    x1 = scipy.stats.norm.rvs(scale=1 / 6, size=N) + np.sin(x0 * 1.2)
    return np.column_stack((x0 / 1.5, x1 * 1.5))


def logpdf_wavy(X):
    """Log-density of `sample_wavy` as example_05.py:41-68 evaluates it (beta argument clipped to [1e-6, 1-1e-6])."""
    import scipy.stats
    x0 = X[:, 0] * 1.5
    x1 = X[:, 1] / 1.5 - np.sin(x0 * 1.2)
    loc = np.clip((x0 / 3 + 1) / 2, 0.000001, 0.999999)
    return np.log(1 / 6) + scipy.stats.beta.logpdf(loc, a=2, b=2) + scipy.stats.norm.logpdf(x1, scale=1 / 6)


def reference_samples(N, D, seed=1):
    """Z drawn from the standard Gaussian reference (inverse-map input)."""
    return np.random.default_rng(seed).standard_normal((N, D))


# ---------------------------------------------------------------------------
# named configurations
# ---------------------------------------------------------------------------

def config(name):
    """Return a dict(monotone, nonmonotone, kwargs, sampler) for a named config."""
    if name == 'C1':
        mon, non = spiral_spec(3)
        return dict(monotone=mon, nonmonotone=non, sampler=sample_spiral,
                    kwargs=dict(monotonicity='integrated rectifier',
                                quadrature_input={'order': 25}))
    if name == 'EX01':
        # example_01.py:121-170 at its shipped maxorder = 10 (the reference's only known-answer fixture: dict_coeffs_order=10.p)
        mon, non = spiral_spec(10)
        return dict(monotone=mon, nonmonotone=non, sampler=sample_spiral,
                    kwargs=dict(monotonicity='integrated rectifier',
                                quadrature_input={'order': 25}))
    if name == 'C2a':
        mon, non = spiral_spec(5)
        return dict(monotone=mon, nonmonotone=non, sampler=sample_spiral,
                    kwargs=dict(monotonicity='integrated rectifier',
                                quadrature_input={'order': 25}))
    if name == 'C2b':
        mon, non = temperature_spec(5)
        return dict(monotone=mon, nonmonotone=non, sampler=sample_spiral,
                    kwargs=dict(monotonicity='separable monotonicity'))
    if name == 'EX03':
        # example_03.py:103-159 at its shipped maxorder = 10 (LET + 9 iRBF + RET, Hermite-function orders 1..10 of x_{k-1}); the
        # script's temperature records are data files: the spiral target stands in for them
        mon, non = temperature_spec(10)
        return dict(monotone=mon, nonmonotone=non, sampler=sample_spiral,
                    kwargs=dict(monotonicity='separable monotonicity'))
    if name == 'C3':
        mon, non = dense_separable_spec(4, 4)
        return dict(monotone=mon, nonmonotone=non, sampler=sample_banana,
                    kwargs=dict(monotonicity='separable monotonicity'))
    if name == 'C3int':
        mon, non = banded_integrated_spec(4, 4, 1)
        return dict(monotone=mon, nonmonotone=non, sampler=sample_banana,
                    kwargs=dict(monotonicity='integrated rectifier',
                                quadrature_input={'order': 25}))
    if name == 'C5':
        mon, non = banded_separable_spec(40, 2)
        return dict(monotone=mon, nonmonotone=non, sampler=sample_mixture,
                    kwargs=dict(monotonicity='separable monotonicity'))
    if name == 'C5int':
        mon, non = banded_integrated_spec(40, 3, 2)
        return dict(monotone=mon, nonmonotone=non, sampler=sample_mixture,
                    kwargs=dict(monotonicity='integrated rectifier',
                                quadrature_input={'order': 25}))
    if name == 'EX05':
        mon, non = density_example_spec(3)
        return dict(monotone=mon, nonmonotone=non, sampler=sample_wavy,
                    kwargs=dict(monotonicity='separable monotonicity', quadrature_input={'order': 25}))
    raise KeyError(name)
