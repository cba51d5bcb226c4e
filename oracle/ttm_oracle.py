"""
CPU ORACLE - TEST INFRASTRUCTURE ONLY.

A NumPy/SciPy restatement of the hot path of the reference
``transport_map.py`` (MaxRamgraber/Triangular-Transport-Toolbox @ 2025-08-01),
written from the reference's *behaviour* (formulas, ordering rules, quirks) and
citing the reference line ranges each function follows (``TM:a-b`` =
/root/reference/transport_map.py lines a-b).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module, and only as the checker / the timed CPU baseline.
The product (``triangular_transport_toolbox_amd``) never imports it and has no
CPU fallback.

Parity status: PINNED.  The reference ships exactly one known-answer fixture
(Example 01's ``dict_coeffs_order=10.p``); beyond that it has no tests, so the
oracle is pinned against outputs of the reference itself, run in the
development container by ``tests/golden/make_golden.py`` and committed as
``tests/golden/*.npz|json`` (see ``tests/test_oracle_golden.py``).

Third-party algorithms the reference delegates to (not vendored, versions are
those of the image: NumPy 2.2.6, SciPy 1.15.3) and how they are restated here:
  np.polynomial.<family>(c)(x)   Clenshaw evaluation         -> same NumPy call
  np.polynomial.legendre.legroots / legder (Gauss nodes)     -> same NumPy call
  np.quantile (method='linear')                              -> same NumPy call
  scipy.special.erf                                          -> same SciPy call
  scipy.interpolate.interp1d (linear, extrapolate)           -> restated in
      ``interp1d_linear`` (stable argsort, searchsorted-left, slope form) and
      checked against SciPy in the tests
  scipy.optimize.minimize (BFGS / L-BFGS-B)                  -> same SciPy call
  scipy.stats.multivariate_normal.logpdf (identity cov)      -> closed form
"""

import copy
import itertools

import numpy as np
import scipy.special

_FAMILIES = {
    # TM:274-304
    'standard': ('Polynomial', np.polynomial.polynomial.Polynomial, np.polynomial.polynomial.polyder),
    'polynomial': ('Polynomial', np.polynomial.polynomial.Polynomial, np.polynomial.polynomial.polyder),
    'power series': ('Polynomial', np.polynomial.polynomial.Polynomial, np.polynomial.polynomial.polyder),
    'hermite': ('Hermite', np.polynomial.hermite.Hermite, np.polynomial.hermite.hermder),
    "phycisist's hermite": ('Hermite', np.polynomial.hermite.Hermite, np.polynomial.hermite.hermder),
    'phycisists hermite': ('Hermite', np.polynomial.hermite.Hermite, np.polynomial.hermite.hermder),
    'hermite_e': ('HermiteE', np.polynomial.hermite_e.HermiteE, np.polynomial.hermite_e.hermeder),
    "probabilist's hermite": ('HermiteE', np.polynomial.hermite_e.HermiteE, np.polynomial.hermite_e.hermeder),
    'probabilists hermite': ('HermiteE', np.polynomial.hermite_e.HermiteE, np.polynomial.hermite_e.hermeder),
    'chebyshev': ('Chebyshev', np.polynomial.chebyshev.Chebyshev, np.polynomial.chebyshev.chebder),
    'laguerre': ('Laguerre', np.polynomial.laguerre.Laguerre, np.polynomial.laguerre.lagder),
    'legendre': ('Legendre', np.polynomial.legendre.Legendre, np.polynomial.legendre.legder),
    'hermite function': ('HermiteE', np.polynomial.hermite_e.HermiteE, np.polynomial.hermite_e.hermeder),
    'hermite_function': ('HermiteE', np.polynomial.hermite_e.HermiteE, np.polynomial.hermite_e.hermeder),
    'hermite functions': ('HermiteE', np.polynomial.hermite_e.HermiteE, np.polynomial.hermite_e.hermeder),
}


def gauss_legendre(order):
    """Nodes/weights exactly as TM:199-225 (legroots + 2/((1-x^2) P'(x)^2))."""
    coefs = [0] * order + [1]
    coefs_der = np.polynomial.legendre.legder(coefs)
    LegendreDer = np.polynomial.legendre.Legendre(coefs_der)
    xis = np.polynomial.legendre.legroots(coefs)
    Ws = 2.0 / ((1.0 - xis ** 2) * (LegendreDer(xis) ** 2))
    return xis, Ws


def hf_constant(polyfunc, n):
    """a_n = 1/max|P_n(x) exp(-x^2/4)| on linspace(-100,100,100001), TM:1102-1109."""
    hf_x = np.linspace(-100, 100, 100001)
    hfeval = polyfunc([0.] * n + [1.])(hf_x) * np.exp(-hf_x ** 2 / 4)
    return 1 / np.max(np.abs(hfeval))


def interp1d_linear(x, y, x_new):
    """scipy.interpolate.interp1d(x, y, fill_value='extrapolate')(x_new), linear:
    stable sort of x, searchsorted (left) clipped to [1, n-1], slope form."""
    ind = np.argsort(x, kind='mergesort')
    x = x[ind]
    y = y[ind]
    idx = np.searchsorted(x, x_new)
    idx = idx.clip(1, len(x) - 1).astype(int)
    lo = idx - 1
    hi = idx
    slope = (y[hi] - y[lo]) / (x[hi] - x[lo])
    return slope * (x_new - x[lo]) + y[lo]


class Rectifier:
    """TM:4956-5213 (evaluate, evaluate_dfdc, logevaluate)."""

    def __init__(self, mode='softplus', delta=1e-8):
        self.mode = mode
        self.delta = delta

    def evaluate(self, X):
        if self.mode == 'squared':
            return X ** 2
        if self.mode == 'exponential':
            return np.exp(X)
        if self.mode == 'expneg':
            return np.exp(-X)
        if self.mode == 'softplus':
            a = np.log(2)
            aX = a * X
            aX[aX < 0] = 0
            return np.log(1 + np.exp(-np.abs(a * X))) + aX
        if self.mode == 'explinearunit':
            res = np.zeros(X.shape)
            res[X < 0] = np.exp(X[X < 0])
            res[X >= 0] = X[X >= 0] + 1
            return res
        raise ValueError(self.mode)

    def evaluate_dfdc(self, f, dfdc):
        if self.mode == 'exponential':
            res = np.exp(f)
        elif self.mode == 'expneg':
            res = -np.exp(-f)
        elif self.mode == 'softplus':
            res = 1 / (1 + np.exp(-np.log(2) * f))
        else:
            raise Exception('Not implemented yet.')
        return res[:, None] * dfdc

    def logevaluate(self, X):
        if self.mode == 'squared':
            return np.log(X ** 2)
        if self.mode == 'exponential':
            return X if self.delta == 0 else np.log(np.exp(X) + self.delta)
        if self.mode == 'expneg':
            return -X
        if self.mode == 'softplus':
            return np.log(self.evaluate(X) + self.delta)
        if self.mode == 'explinearunit':
            return np.log(self.evaluate(X))
        raise ValueError(self.mode)


def gauss_quadrature(f, b, xis, Ws):
    """Vectorised fixed-order Gauss-Legendre of f over [0, b] (a = 0), nodes
    accumulated sequentially i = 0..Q-1 as TM:4214-4278.  f returns (N,) or (N,C)."""
    a = np.ones(b.shape) * 0
    lim_dif = b - a
    lim_sum = b + a
    funcres = f(lim_dif * 0.5 * xis[0] + lim_sum * 0.5)
    if funcres.ndim == lim_dif.ndim:
        result = lim_dif * 0.5 * (Ws[0] * funcres)
        for i in range(1, len(Ws)):
            funcres = f(lim_dif * 0.5 * xis[i] + lim_sum * 0.5)
            result += lim_dif * 0.5 * (Ws[i] * funcres)
    else:
        result = (lim_dif * 0.5 * Ws[0])[:, None] * funcres
        for i in range(1, len(Ws)):
            funcres = f(lim_dif * 0.5 * xis[i] + lim_sum * 0.5)
            result += (lim_dif * 0.5 * Ws[i])[:, None] * funcres
    return result


class OracleMap:
    """CPU restatement of the reference class for the hot path (see module doc)."""

    def __init__(self, X, monotone, nonmonotone, polynomial_type='hermite function',
                 monotonicity='integrated rectifier', standardize_samples=True,
                 standardization='standard', ST_scale_factor=1.0, ST_scale_mode='dynamic',
                 coeffs_init=0., alternate_root_finding=True, root_search_truncation=True,
                 regularization=None, regularization_lambda=0.1, quadrature_input=None,
                 rectifier_type='exponential', delta=1e-8, verbose=False, linearization=None,
                 linearization_specified_as_quantiles=True, linearization_increment=1e-6, **ignored):
        self.monotone = copy.deepcopy(monotone)
        self.nonmonotone = copy.deepcopy(nonmonotone)
        self.rect = Rectifier(rectifier_type, delta)
        self.delta = delta
        q = dict(quadrature_input or {})
        if 'xis' not in q and 'Ws' not in q:
            q['xis'], q['Ws'] = gauss_legendre(q.get('order', 100))     # TM:199-225
        self.xis, self.Ws = np.asarray(q['xis']), np.asarray(q['Ws'])
        if ST_scale_mode not in ('dynamic', 'static'):
            raise ValueError("'ST_scale_mode' must be either 'dynamic' or 'static'.")
        self.ST_scale_factor, self.ST_scale_mode = ST_scale_factor, ST_scale_mode
        self.standardization = standardization
        self.coeffs_init = coeffs_init
        self.alternate_root_finding = alternate_root_finding
        self.root_search_truncation = root_search_truncation
        self.regularization, self.regularization_lambda = regularization, regularization_lambda
        self.monotonicity = monotonicity
        if monotonicity.lower() not in ('integrated rectifier', 'separable monotonicity'):
            raise ValueError('monotonicity')
        self.linearization = linearization                                       # TM:254-257
        self.linearization_specified_as_quantiles = linearization_specified_as_quantiles
        self.linearization_increment = linearization_increment
        if linearization is not None and monotonicity.lower() == 'separable monotonicity':
            # TM:2063-2080: the derivative functions of a separable map overwrite their input with min(x, lower
            # threshold) in every column (inverted masks) - not restated
            raise NotImplementedError('linearization with separable monotonicity (reference defect TM:2063-2080)')
        if polynomial_type.lower() not in _FAMILIES:
            raise Exception('Polynomial type not understood.')
        self.hf_allowed = polynomial_type.lower().startswith('hermite f') or polynomial_type.lower() == 'hermite_function'
        _, self.polyfunc, self.polyfunc_der = _FAMILIES[polynomial_type.lower()]
        self.X = np.array(X, dtype=float, copy=True)
        self.standardize_samples = standardize_samples
        if standardize_samples:
            self.standardize()
        self.D = len(monotone)
        self.skip_dimensions = X.shape[-1] - self.D
        self._hf = {}
        self.check_for_special_terms()
        self.determine_special_term_locations()
        self._plan()
        self.coeffs_mon = [np.ones(self.n_mon[k]) * coeffs_init for k in range(self.D)]
        self.coeffs_nonmon = [np.ones(len(self.nonmonotone[k])) * coeffs_init for k in range(self.D)]

    # ------------------------------------------------------------------ a1
    def standardize(self):
        """TM:750-787."""
        if self.standardization.lower() == 'standard':
            self.X_mean = np.mean(self.X, axis=0)
            self.X_std = np.std(self.X, axis=0)
        elif self.standardization.lower() in ('quantile', 'quantiles'):
            self.X_mean = np.quantile(self.X, q=0.5, axis=0)
            self.X_std = (np.quantile(self.X - self.X_mean, q=0.8413447460685429, axis=0) -
                          np.quantile(self.X - self.X_mean, q=0.15865525393145707, axis=0)) / 2
        else:
            raise ValueError("'standardization' must be either 'standard' or 'quantiles'.")
        self.X -= self.X_mean
        self.X /= self.X_std

    def reset(self, X):
        """TM:710-748."""
        if len(X.shape) != 2:
            raise Exception('X should be a two-dimensional array')
        self.X = np.array(X, dtype=float, copy=True)
        if self.standardize_samples:
            self.standardize()
        for k in range(self.D):
            self.coeffs_mon[k] = self.coeffs_mon[k] * 0 + self.coeffs_init
            self.coeffs_nonmon[k] = self.coeffs_nonmon[k] * 0 + self.coeffs_init
        self.determine_special_term_locations()

    # ------------------------------------------------------------------ a14
    def check_for_special_terms(self):
        """TM:2136-2217: count special terms per (component, variable)."""
        self.special_terms = {}
        for k in range(self.D):
            kc = k + self.skip_dimensions
            st = self.special_terms[kc] = {}
            for entry in self.nonmonotone[k]:
                if type(entry) == str:
                    index = int(entry.split(' ')[1])
                    st.setdefault(index, {'counter': 0, 'centers': np.asarray([]), 'scales': np.asarray([])})
                    st[index]['counter'] += 1
            for entry in self.monotone[k]:
                if type(entry) == str:
                    index = int(entry.split(' ')[1])
                    if index == kc:
                        st.setdefault(index, {'counter': 0, 'centers': np.asarray([]), 'scales': np.asarray([])})
                        st[index]['counter'] += 1
                    else:
                        ct = st.setdefault('cross-terms', {})
                        ct.setdefault(index, {'counter': 0, 'centers': np.asarray([]), 'scales': np.asarray([])})
                        ct[index]['counter'] += 1

    def _place(self, dictionary):
        """TM:2241-2330."""
        for d in [key for key in dictionary.keys() if key != 'cross-terms']:
            n = dictionary[d]['counter']
            if n == 1:
                dictionary[d]['centers'] = np.asarray([np.quantile(self.X[:, d], q=0.5)])
                dictionary[d]['scales'] = np.asarray(
                    [self.ST_scale_factor / 2 if self.ST_scale_mode == 'dynamic' else self.ST_scale_factor])
            elif n > 1:
                quantiles = np.arange(1, n + 1, 1) / (n + 1)
                scales = np.zeros(n)
                c = dictionary[d]['centers'] = np.quantile(a=self.X[:, d], q=quantiles)
                if self.ST_scale_mode == 'dynamic':
                    for i in range(n):
                        if i == 0:
                            scales[i] = (c[1] - c[0]) * self.ST_scale_factor
                        elif i == n - 1:
                            scales[i] = (c[i] - c[i - 1]) * self.ST_scale_factor
                        else:
                            scales[i] = (c[i + 1] - c[i - 1]) / 2 * self.ST_scale_factor
                    dictionary[d]['scales'] = scales
                else:
                    dictionary[d]['scales'] = scales + self.ST_scale_factor
        return dictionary

    def determine_special_term_locations(self):
        """TM:2219-2389, linearisation thresholds TM:2364-2389."""
        for kc in np.arange(self.D) + self.skip_dimensions:
            if 'cross-terms' in self.special_terms[kc]:
                self.special_terms[kc]['cross-terms'] = self._place(self.special_terms[kc]['cross-terms'])
            self.special_terms[kc] = self._place(self.special_terms[kc])
        if self.linearization is not None:
            self.linearization_threshold = np.zeros((self.X.shape[-1], 2))
            for k in range(self.X.shape[-1]):
                if self.linearization_specified_as_quantiles:
                    self.linearization_threshold[k, 0] = np.quantile(self.X[:, k], q=self.linearization)
                    self.linearization_threshold[k, 1] = np.quantile(self.X[:, k], q=1 - self.linearization)
                else:
                    self.linearization_threshold[k, 0] = -self.linearization
                    self.linearization_threshold[k, 1] = +self.linearization

    # ------------------------------------------------------------------ a2
    def _plan(self):
        """Resolve every list entry to a term plan; apply the special-term
        cross-grid re-ordering of TM:1446-1483.  A plan is a list of factors:
          ('const',) | ('poly', var, order, hf, lin) | ('st', kind, var, cross, index)."""
        self.plan_mon, self.plan_nonmon, self.n_mon = [], [], []
        for k in range(self.D):
            kc = k + self.skip_dimensions
            for which in ('mon', 'nonmon'):
                entries = self.monotone[k] if which == 'mon' else self.nonmonotone[k]
                counter = {}
                terms, st_idx = [], []
                for i, entry in enumerate(entries):
                    if type(entry) == str:
                        kind, var = entry.split(' ')
                        var = int(var)
                        if kind.lower() not in ('let', 'ret', 'rbf', 'irbf'):
                            raise ValueError("Special term '" + kind + "' not understood.")
                        cross = (which == 'mon') and (var != kc)
                        c = counter.get(var, 0)
                        counter[var] = c + 1
                        terms.append([('st', kind.lower(), var, cross, c)])
                        st_idx.append(i)
                    elif len(entry) == 0:
                        terms.append([('const',)])
                    else:
                        hf = any(e == 'HF' for e in entry)
                        lin = any(e == 'LIN' for e in entry)
                        if lin and self.linearization is None:                    # TM:1053-1054
                            raise Exception("'LIN' modifier specified in variable monotone, but the variable "
                                            "linearization is defined as None. Please specify a scalar linearization "
                                            "or remove the 'LIN' modifier.")
                        ints = [e for e in entry if type(e) != str]
                        ui, ct = np.unique(ints, return_counts=True)
                        terms.append([('poly', int(u), int(c), hf, lin) for u, c in zip(ui, ct)])
                if which == 'mon' and 'cross-terms' in self.special_terms[kc]:
                    rbf = [terms[i] for i in st_idx]
                    dims = sorted(set(t[0][2] for t in rbf))
                    by_dim = {d: [t for t in rbf if t[0][2] == d] for d in dims}
                    grid = list(by_dim[dims[0]])
                    for d in dims[1:]:
                        grid = [a + b for a, b in itertools.product(grid, by_dim[d])]
                    terms = [t for i, t in enumerate(terms) if i not in st_idx] + grid
                if which == 'mon':
                    self.plan_mon.append(terms)
                    self.n_mon.append(len(terms))
                else:
                    self.plan_nonmon.append(terms)

    def _st_params(self, kc, var, cross, index):
        d = self.special_terms[kc]['cross-terms'][var] if cross else self.special_terms[kc][var]
        return d['centers'][index], d['scales'][index]

    def _hfconst(self, n):
        if n not in self._hf:
            self._hf[n] = hf_constant(self.polyfunc, n)
        return self._hf[n]

    def _factor(self, f, x, kc, derivative_wrt=None):
        """Value (or d/dx_{derivative_wrt}) of one factor on the sample matrix x."""
        if f[0] == 'const':
            return np.ones(x.shape[:-1]) if derivative_wrt is None else np.zeros(x.shape[:-1])
        if f[0] == 'poly':
            _, var, order, hf, lin = f
            c = [0.] * order + [1.]
            if hf:
                c[-1] = self._hfconst(order)
            xv = x[..., var]
            if derivative_wrt is None or var != derivative_wrt:
                val = self.polyfunc(c)(xv)
                if hf:
                    val = val * np.exp(-xv ** 2 / 4)
                if lin:
                    # TM:1375-1385 as it executes: "__x__" has already been replaced by "x" when the clipped /
                    # extended variables are substituted, so BOTH sides of the blend evaluate the factor at the
                    # unclipped x: P(x) (1 - vec/inc) + P(x) vec/inc = P(x) up to rounding (noise ~ |vec|/inc eps)
                    thr = self.linearization_threshold[var]
                    vec = np.where(xv - thr[0] >= 0, 0.0, xv - thr[0]) + np.where(xv - thr[1] <= 0, 0.0, xv - thr[1])
                    val = val * (1 - vec / self.linearization_increment) + val * vec / self.linearization_increment
                return val
            cder = self.polyfunc_der(c)
            if not hf:
                return self.polyfunc(cder)(xv)                                    # TM:1166-1206
            return -1 / 2 * np.exp(-xv ** 2 / 4) * (xv * self.polyfunc(c)(xv) - 2 * self.polyfunc(cder)(xv))  # TM:1245
        _, kind, var, cross, index = f
        mu, sc = self._st_params(kc, var, cross, index)
        xv = x[..., var]
        erf = scipy.special.erf
        if derivative_wrt is None:                                               # TM:917-1002
            if kind == 'let':
                return ((xv - mu) * (1 - erf((xv - mu) / (np.sqrt(2) * sc))) -
                        sc * np.sqrt(2 / np.pi) * np.exp(-((xv - mu) / (np.sqrt(2) * sc)) ** 2)) / 2
            if kind == 'ret':
                return ((xv - mu) * (1 + erf((xv - mu) / (np.sqrt(2) * sc))) +
                        sc * np.sqrt(2 / np.pi) * np.exp(-((xv - mu) / (np.sqrt(2) * sc)) ** 2)) / 2
            if kind == 'rbf':
                return np.exp(-((xv - mu) / sc) ** 2 / 2) / (sc * np.sqrt(2 * np.pi))
            return (1 + erf((xv - mu) / (np.sqrt(2) * sc))) / 2
        if var != derivative_wrt:
            return np.zeros(x.shape[:-1])
        if kind == 'let':                                                        # TM:926-1016
            return (1 - erf((xv - mu) / (np.sqrt(2) * sc))) / 2
        if kind == 'ret':
            return (1 + erf((xv - mu) / (np.sqrt(2) * sc))) / 2
        if kind == 'rbf':
            return -(xv - mu) / (np.sqrt(2 * np.pi) * sc ** 3) * np.exp(-((xv - mu) / sc) ** 2 / 2)
        return 1 / (np.sqrt(2 * np.pi) * sc) * np.exp(-(xv - mu) ** 2 / (2 * sc ** 2))

    def _basis(self, plans, x, kc, derivative=False):
        cols = []
        for t in plans:
            if not derivative:
                col = self._factor(t[0], x, kc)
                for f in t[1:]:
                    col = col * self._factor(f, x, kc)
            else:
                vars_in = [f[1] if f[0] == 'poly' else (f[2] if f[0] == 'st' else None) for f in t]
                if kc not in vars_in:
                    col = np.zeros(x.shape[:-1])                                  # TM:1255-1258
                else:
                    hf_multi = any(f[0] == 'poly' and f[3] for f in t) and len(t) > 1
                    if hf_multi:
                        raise NotImplementedError('HF cross-term derivative (reference quirk 7, TM:1245)')
                    col = None
                    for f in t:
                        v = self._factor(f, x, kc, derivative_wrt=kc)
                        col = v if col is None else col * v
            cols.append(col)
        return np.stack(cols, axis=-1)

    def fun_mon(self, k, x):
        return self._basis(self.plan_mon[k], x, k + self.skip_dimensions)

    def fun_nonmon(self, k, x):
        if len(self.plan_nonmon[k]) == 0:
            return None                                                           # TM:1817-1821
        return self._basis(self.plan_nonmon[k], x, k + self.skip_dimensions)

    def der_fun_mon(self, k, x):
        return self._basis(self.plan_mon[k], x, k + self.skip_dimensions, derivative=True)

    # ------------------------------------------------------------------ a3/a4
    def s(self, x, k, coeffs_nonmon=None, coeffs_mon=None):
        """TM:2439-2567."""
        if x is None:
            x = self.X
        cm = self.coeffs_mon[k] if coeffs_mon is None else coeffs_mon
        cn = self.coeffs_nonmon[k] if coeffs_nonmon is None else coeffs_nonmon
        Psi_nonmon = self.fun_nonmon(k, x)
        nonmonotone_part = 0 if Psi_nonmon is None else np.dot(Psi_nonmon, cn[:, None])[..., 0]
        kc = self.skip_dimensions + k
        if self.monotonicity == 'integrated rectifier':
            def integral_argument(t):
                X_loc = copy.copy(x)
                X_loc[:, kc] = t
                arg = self.rect.evaluate(np.dot(self.fun_mon(k, X_loc), cm[:, None])[..., 0])
                arg += self.delta
                return arg
            monotone_part = gauss_quadrature(integral_argument, x[..., kc], self.xis, self.Ws)
        else:
            monotone_part = np.dot(self.fun_mon(k, x), cm[:, None])[:, 0]
        return nonmonotone_part + monotone_part

    def _standardized(self, X):
        if X is not None and self.standardize_samples:                            # TM:2410-2422
            X = np.array(X, dtype=float, copy=True)
            X -= self.X_mean
            X /= self.X_std
            return X
        return copy.copy(self.X)

    def map(self, X=None):
        """TM:2391-2437."""
        X = self._standardized(X)
        Z = np.zeros((X.shape[0], self.D))
        for k in range(self.D):
            Z[:, k] = self.s(X, k)
        return Z

    # ------------------------------------------------------------------ a7
    def _reg(self, k, div, cn, cm, grad):
        lam = self.regularization_lambda
        if self.regularization is None:
            return 0
        r = self.regularization.lower()
        if r not in ('l1', 'l2'):
            raise ValueError("regularization_type must be either 'l1' or 'l2'.")
        if np.isscalar(lam):
            ln, lm = lam, lam
        else:
            ln, lm = np.asarray(lam[k][:div]), np.asarray(lam[k][div:])
        if not grad:
            if r == 'l1':
                return np.sum(lm * np.abs(cm)) + np.sum(ln * np.abs(cn))
            return np.sum(lm * cm ** 2) + np.sum(ln * cn ** 2)
        if r == 'l1':
            return np.concatenate((ln * np.sign(cn), lm * np.sign(cm)))
        return np.concatenate((ln * 2 * cn, lm * 2 * cm))

    def objective_function(self, coeffs, k, div=0):
        """TM:3300-3433."""
        cn, cm = (coeffs[:div], coeffs[div:]) if coeffs is not None else (self.coeffs_nonmon[k], self.coeffs_mon[k])
        map_result = self.s(None, k, cn, cm)
        objective = 1 / 2 * map_result ** 2
        g = np.dot(self.fun_mon(k, self.X), cm[:, None])[..., 0]
        objective -= self.rect.logevaluate(g)
        objective = np.mean(objective)
        if self.regularization is not None:
            lam = self.regularization_lambda
            r = self.regularization.lower()
            if np.isscalar(lam):
                if r == 'l1':
                    objective += lam * np.sum(np.abs(cm))
                    objective += lam * np.sum(np.abs(cn))
                elif r == 'l2':
                    objective += lam * np.sum(cm ** 2)
                    objective += lam * np.sum(cn ** 2)
            else:
                objective += self._reg(k, div, cn, cm, False)
        return objective

    def objective_function_jacobian(self, coeffs, k, div=0):
        """TM:3435-3635."""
        cn, cm = (coeffs[:div], coeffs[div:]) if coeffs is not None else (self.coeffs_nonmon[k], self.coeffs_mon[k])
        kc = self.skip_dimensions + k
        term_1_scalar = self.s(None, k, cn, cm)

        def integral_argument_term1_jac(t):
            X_loc = copy.copy(self.X)
            X_loc[:, kc] = t
            Psi = self.fun_mon(k, X_loc)
            return self.rect.evaluate_dfdc(np.dot(Psi, cm[:, None])[..., 0], Psi)
        term_1_vector = gauss_quadrature(integral_argument_term1_jac, self.X[:, kc], self.xis, self.Ws)
        Psi_nonmon = self.fun_nonmon(k, self.X)
        if Psi_nonmon is not None:
            term_1_vector = np.column_stack((Psi_nonmon, term_1_vector))
        term_1 = term_1_scalar[:, None] * term_1_vector
        Psi_mon = self.fun_mon(k, self.X)
        rec_arg = np.dot(Psi_mon, cm[:, None])[..., 0]
        numer = self.rect.evaluate_dfdc(rec_arg, Psi_mon)
        denom = 1 / (self.rect.evaluate(rec_arg) + self.delta)
        term_2 = numer * denom[:, None]
        if div > 0:
            term_2 = np.column_stack((np.zeros((term_2.shape[0], div)), term_2))
        objective = np.mean(term_1 - term_2, axis=0)
        if self.regularization is not None:
            objective = objective + self._reg(k, div, cn, cm, True)
        return objective

    # ------------------------------------------------------------------ a8
    def separable_setup(self, k):
        """A matrix of the reduced separable objective, TM:2959-2975 (QR) and
        TM:3021-3050 (L2).  Returns (A, aux) with aux what the closed-form
        nonmonotone solve needs."""
        Pn, Pm = self.fun_nonmon(k, self.X), self.fun_mon(k, self.X)
        N = self.X.shape[0]
        if self.regularization is None:
            Q, R = np.linalg.qr(Pn, mode='reduced')
            A_sqrt = Pm - np.linalg.multi_dot((Q, Q.T, Pm))
            return np.dot(A_sqrt.T, A_sqrt) / N, (Q, R, Pm)
        if self.regularization.lower() == 'l2':
            lam = self.regularization_lambda
            G = np.linalg.multi_dot((np.linalg.inv(np.dot(Pn.T, Pn) + lam * np.identity(Pn.shape[-1])), Pn.T, Pm))
            Dl = Pm - np.dot(Pn, G)
            A = np.dot(Dl.T, Dl) / 2 + lam * (np.dot(G.T, G) + np.identity(G.shape[-1]))
            return A, (Pn, Pm)
        raise ValueError(self.regularization)

    def separable_objective(self, coeffs_mon, A, k):
        """TM:2978-3018 / 3053-3094: (objective, gradient)."""
        dPsi = self.der_fun_mon(k, self.X)
        N = self.X.shape[0]
        b = self.delta * np.sum(A, axis=-1)
        Ax = np.dot(A, coeffs_mon[:, None])
        dS = np.dot(dPsi, coeffs_mon[:, None]) + np.sum(dPsi, axis=-1)[:, None] * self.delta
        objective = np.dot(coeffs_mon[None, :], Ax)[0, 0] / 2 - np.sum(np.log(dS)) / N + np.inner(coeffs_mon, b)
        grad = Ax[:, 0] - np.sum(dPsi / dS, axis=0) / N + b
        return objective, grad

    def separable_nonmonotone(self, coeffs_mon, aux):
        """Closed-form nonmonotone coefficients, TM:3148-3169."""
        if self.regularization is None:
            Q, R, Pm = aux
            return -np.linalg.multi_dot((np.linalg.inv(R), Q.T, Pm, coeffs_mon[:, None]))[:, 0]
        Pn, Pm = aux
        lam = self.regularization_lambda
        return -np.linalg.multi_dot((np.linalg.inv(np.dot(Pn.T, Pn) + 2 * lam * np.identity(Pn.shape[-1])),
                                     np.dot(Pn.T, Pm), coeffs_mon[:, None]))[:, 0]

    def bounds(self, k):
        """L-BFGS-B bounds, TM:1891-1892,1925-1929: c >= 0 except constant terms."""
        out = []
        for entry in self.monotone[k]:
            out.append([-np.inf, np.inf] if (type(entry) != str and len(entry) == 0) else [0., np.inf])
        return out

    def optimize(self, K=None):
        """TM:2714-2901 with workers == 1 (same SciPy calls as TM:3108-3114, 3252-3257)."""
        from scipy.optimize import minimize
        for k in (range(self.D) if K is None else K):
            if self.monotonicity == 'integrated rectifier':
                div = len(self.coeffs_nonmon[k])
                x0 = np.concatenate((self.coeffs_nonmon[k], self.coeffs_mon[k]))
                opt = minimize(method='BFGS', fun=self.objective_function, jac=self.objective_function_jacobian,
                               x0=x0, args=(k, div))
                self.coeffs_nonmon[k], self.coeffs_mon[k] = opt.x[:div].copy(), opt.x[div:].copy()
            else:
                A, aux = self.separable_setup(k)
                opt = minimize(fun=lambda c: self.separable_objective(c, A, k), method='L-BFGS-B',
                               x0=self.coeffs_mon[k], jac=True, bounds=self.bounds(k))
                self.coeffs_mon[k] = opt.x
                self.coeffs_nonmon[k] = self.separable_nonmonotone(opt.x, aux)

    # ------------------------------------------------------------------ a10-a12
    def inverse_map(self, Z, X_star=None):
        """TM:3639-3796."""
        Z = np.array(Z, dtype=float, copy=True)
        N = Z.shape[0]
        search = (self.vectorized_root_search_alternate
                  if self.alternate_root_finding and self.monotonicity.lower() == 'separable monotonicity'
                  else self.vectorized_root_search_bisection)
        if X_star is None:
            X = np.zeros((N, self.skip_dimensions + self.D))
            for k in range(self.D):
                X = search(X, Z[:, k], k)
            if self.standardize_samples:
                X *= self.X_std
                X += self.X_mean
        elif X_star.shape[-1] == self.skip_dimensions:
            X = np.zeros((N, self.skip_dimensions + self.D))
            X[:, :self.skip_dimensions] = X_star
            if self.standardize_samples:
                X[:, :self.skip_dimensions] -= self.X_mean[:self.skip_dimensions]
                X[:, :self.skip_dimensions] /= self.X_std[:self.skip_dimensions]
            for k in range(self.D):
                X = search(X, Z[:, k], k)
            if self.standardize_samples:
                X *= self.X_std
                X += self.X_mean
        elif self.skip_dimensions == 0:
            skip = X_star.shape[-1]
            D = skip + Z.shape[-1]
            X = np.zeros((N, D))
            X[:, :skip] = X_star
            if self.standardize_samples:
                X[:, :skip] -= self.X_mean[:skip]
                X[:, :skip] /= self.X_std[:skip]
            for i, k in enumerate(range(skip, D)):
                X = search(X, Z[:, i], k)
            if self.standardize_samples:
                X *= self.X_std
                X += self.X_mean
        return X[:, self.skip_dimensions:]

    def vectorized_root_search_bisection(self, X, Zk, k, max_iterations=100, threshold=1e-9, start_distance=2):
        """TM:3798-3985, same index-set bookkeeping (incl. the ``np.sum(indices) > 0``
        loop guard that drops sample 0)."""
        kc = self.skip_dimensions + k
        N = X.shape[0]
        indices = np.arange(N)
        indices = indices[~np.isnan(X[:, kc])]
        pts = np.zeros((N, 2))
        pts[:, 0] = -start_distance
        pts[:, 1] = +start_distance
        out = np.zeros((N, 2))
        X[indices, kc] = pts[indices, 0]
        out[indices, 0] = self.s(X[indices, :], k) - Zk[indices]
        X[indices, kc] = pts[indices, 1]
        out[indices, 1] = self.s(X[indices, :], k) - Zk[indices]

        def resort(idx):
            sw = idx[out[idx, 0] > out[idx, 1]]
            out[sw] = out[sw][:, ::-1]
            pts[sw] = pts[sw][:, ::-1]
        resort(indices)
        shift = indices[np.where(np.prod(out[indices, :], axis=1) > 0)[0]]
        while len(shift) > 0:
            resort(shift)
            sign_failure = np.sign(out[shift, 0])
            difference = np.diff(pts[shift, :], axis=1)[:, 0]
            pos = shift[np.where(sign_failure > 0)[0]]
            dpos = difference[np.where(sign_failure > 0)[0]]
            pts[pos, 1] = pts[pos, 0]
            pts[pos, 0] -= dpos * 2
            out[pos, 1] = out[pos, 0]
            X[pos, kc] = pts[pos, 0]
            if len(pos):
                out[pos, 0] = self.s(X[pos, :], k) - Zk[pos]
            neg = shift[np.where(sign_failure < 0)[0]]
            dneg = difference[np.where(sign_failure < 0)[0]]
            pts[neg, 0] = pts[neg, 1]
            pts[neg, 1] += dneg * 2
            out[neg, 0] = out[neg, 1]
            X[neg, kc] = pts[neg, 1]
            if len(neg):
                out[neg, 1] = self.s(X[neg, :], k) - Zk[neg]
            shift = shift[np.where(np.prod(out[shift, :], axis=1) > 0)[0]]
        itr = 0
        while np.sum(indices) > 0 and itr < max_iterations:
            itr += 1
            mid_pt = np.mean(pts[indices, :], axis=1)
            X[indices, kc] = mid_pt
            mid_out = self.s(X[indices, :], k) - Zk[indices]
            below = np.where(mid_out < 0)[0]
            above = np.where(mid_out > 0)[0]
            pts[indices[below], 0] = mid_pt[below]
            pts[indices[above], 1] = mid_pt[above]
            indices = indices[np.where(np.abs(mid_out) > threshold)]
        return X

    def vectorized_root_search_alternate(self, X, Zk, k, start_distance=10, resolution=1001):
        """TM:3987-4084."""
        X = copy.copy(X)
        kc = self.skip_dimensions + k
        offset = np.dot(self.fun_nonmon(k, copy.copy(X)), self.coeffs_nonmon[k][:, None])[:, 0]
        pts = np.linspace(-start_distance, start_distance, resolution)
        fakeX = np.zeros((resolution, X.shape[-1]))
        fakeX[:, kc] = pts
        out = np.dot(self.fun_mon(k, fakeX), self.coeffs_mon[k][:, None])[:, 0]
        target = -offset + Zk
        if self.root_search_truncation:
            target[target < np.min(out)] = np.min(out)
            target[target > np.max(out)] = np.max(out)
        X[:, kc] = interp1d_linear(out, pts, target)
        return X

    # ------------------------------------------------------------------ a13
    def _log_determinant(self, X, skip_in_std):
        log_determinant = 0
        for k in range(self.D):
            dS = np.dot(self.der_fun_mon(k, copy.copy(X)), self.coeffs_mon[k][:, None])[:, 0]
            dS /= self.X_std[k + (self.skip_dimensions if skip_in_std else 0)]
            log_determinant += np.log(dS)
        return log_determinant

    def evaluate_pullback_density(self, X, X_star=None):
        """TM:2646-2712, bug-compatible (der_fun_mon on the raw X,
        X_std[k] without the skip offset)."""
        assert self.monotonicity == 'separable monotonicity'
        if X_star is not None:
            X = np.column_stack((X_star, X))
        Z = self.map(X)
        log_ref = -0.5 * self.D * np.log(2 * np.pi) - 0.5 * np.sum(Z ** 2, axis=-1)
        return np.exp(log_ref + self._log_determinant(X, skip_in_std=False))

    def evaluate_pushforward_density(self, Z, log_target_pdf, X_star=None):
        """TM:2569-2644."""
        assert self.monotonicity == 'separable monotonicity'
        X = self.inverse_map(Z, X_star)
        log_target = log_target_pdf(X)
        if X_star is not None:
            X = np.column_stack((X_star, X))
        return np.exp(log_target - self._log_determinant(X, skip_in_std=True))
