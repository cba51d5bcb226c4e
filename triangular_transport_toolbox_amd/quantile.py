"""
np.quantile(method='linear') from exact order statistics.

The reference places special terms and (optionally) standardises with ``np.quantile`` (TM:775-778,
2266-2296).  NumPy's 'linear' method interpolates between two order statistics of the column; the device
returns those two values exactly (``ttm_order_statistics``, radix select), and the O(#quantiles) arithmetic
below restates NumPy 2.2's ``_get_indexes`` / ``_get_gamma`` / ``_lerp`` so that the result is bit-identical
to ``np.quantile`` on the same data (tests/test_quantile.py).
"""
import numpy as np


def plan(n, q):
    """ranks (previous, next) and interpolation weight for quantiles q of n values."""
    q = np.atleast_1d(np.asarray(q, dtype=float))
    virtual = (n - 1) * q
    previous = np.asanyarray(np.floor(virtual))
    nxt = np.asanyarray(previous + 1)
    above = virtual >= n - 1
    previous[above] = -1
    nxt[above] = -1
    below = virtual < 0
    previous[below] = 0
    nxt[below] = 0
    previous = previous.astype(np.intp)
    nxt = nxt.astype(np.intp)
    gamma = np.asanyarray(virtual - previous, dtype=virtual.dtype)
    return previous % n, nxt % n, gamma


def lerp(a, b, t):
    """numpy.lib._function_base_impl._lerp"""
    a, b, t = np.asarray(a, dtype=float), np.asarray(b, dtype=float), np.asarray(t, dtype=float)
    diff = b - a
    out = a + diff * t
    hi = t >= 0.5
    out[hi] = (b - diff * (1 - t))[hi]
    return out


_PLANS = {}


def _plan_lists(n, q):
    """plan(n, q) as plain lists + the sorted unique ranks, kept per (n, q): the placement of a filter asks for the same
    quantiles of the same number of samples in every update."""
    key = (int(n), tuple(np.atleast_1d(np.asarray(q, dtype=float)).tolist()))
    p = _PLANS.get(key)
    if p is None:
        if len(_PLANS) > 256:
            _PLANS.clear()
        prev, nxt, gamma = plan(n, q)
        p = _PLANS[key] = (prev.tolist(), nxt.tolist(), gamma.tolist(), np.unique(np.concatenate((prev, nxt))))
    return p


def quantile_from_order_statistics(n, q, fetch, shift=None):
    """fetch(ranks: sorted unique int array) -> values of those order statistics (same order).
    shift: quantiles of (x - shift) instead of x (a monotone map, so the order statistics shift along)."""
    prev, nxt, gamma, ranks = _plan_lists(n, q)
    vals = np.asarray(fetch(ranks), dtype=float)
    if shift is not None:
        vals = vals - shift
    lut = dict(zip(ranks.tolist(), vals.tolist()))
    # lerp() element by element on plain floats (the same IEEE operations as the array expression)
    out = np.empty(len(prev))
    for i, (rp, rn, t) in enumerate(zip(prev, nxt, gamma)):
        a, b = lut[rp], lut[rn]
        diff = b - a
        out[i] = b - diff * (1 - t) if t >= 0.5 else a + diff * t
    return out
