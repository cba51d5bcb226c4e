"""
Accuracy of the branch-free fp64 elementary functions of csrc/ttm_math.h (host build of the same header,
tests/hostemu) against mpmath / NumPy.  These bounds are what DESIGN.md quotes.
"""
import ctypes

import mpmath as mp
import numpy as np

from tests.hostemu import emu


def run(which, a, b=None):
    a = np.ascontiguousarray(a, dtype=float)
    out = np.empty_like(a)
    bb = None if b is None else np.ascontiguousarray(b, dtype=float)
    emu.lib().emu_math(which, emu.ptr(a), emu.ptr(bb), ctypes.c_int64(a.size), emu.ptr(out))
    return out


def ulps(x, ref):
    return np.abs(x - ref) / np.spacing(np.abs(ref))


def test_exp():
    rng = np.random.default_rng(0)
    a = np.concatenate([rng.uniform(-700, 700, 200000), rng.uniform(-2, 2, 200000), [0.0, -0.0, 1e-300, -745.0, 709.0]])
    got = run(0, a)
    mp.mp.dps = 30
    sub = np.r_[0:2000, 200000:202000]
    ref = np.array([float(mp.exp(mp.mpf(float(x)))) for x in a[sub]])
    assert ulps(got[sub], ref).max() <= 1.0
    assert np.abs(got / np.exp(a) - 1).max() < 4e-16
    assert np.isnan(run(0, [np.nan]))[0]
    assert run(0, [-1e9])[0] == 0.0 or run(0, [-1e9])[0] < 1e-300
    assert run(0, [np.inf])[0] > 1e300


def test_erf_and_gauss_table():
    rng = np.random.default_rng(1)
    t = np.concatenate([rng.uniform(-6.5, 6.5, 30000), np.linspace(-6, 6, 2049), np.arange(33) * 0.1875,
                        np.nextafter(np.arange(33) * 0.1875, -1), [0.0, -0.0, 7.0, -7.0, 1e300, -1e300]])
    mp.mp.dps = 30
    e_ref = np.array([float(mp.erf(mp.mpf(float(x)))) for x in t])
    g_ref = np.array([float(mp.exp(-mp.mpf(float(x)) ** 2)) for x in t])
    e, g = run(1, t), run(2, t)
    assert np.abs(e - e_ref).max() < 2.5e-16
    assert np.abs(g - g_ref).max() < 6e-16
    inner = (np.abs(t) < 4)
    with np.errstate(all='ignore'):
        rel = np.abs(g - g_ref) / g_ref
    assert rel[inner].max() < 1e-12
    assert np.array_equal(run(1, -t), -e) and np.array_equal(run(2, -t), g)      # odd / even exactly
    assert np.isnan(run(1, [np.nan]))[0] and np.isnan(run(2, [np.nan]))[0]
    assert abs(run(1, [np.inf])[0] - 1.0) < 2.5e-16 and abs(run(1, [-np.inf])[0] + 1.0) < 2.5e-16


def test_log_rcp_div():
    rng = np.random.default_rng(2)
    x = np.concatenate([np.exp(rng.uniform(-700, 700, 100000)), rng.uniform(0.5, 2.0, 100000), [1.0, 1e-310, 1e308]])
    got = run(3, x)
    mp.mp.dps = 30
    sub = np.r_[0:1500, 100000:101500, len(x) - 3:len(x)]
    ref = np.array([float(mp.log(mp.mpf(float(v)))) for v in x[sub]])
    assert ulps(got[sub], ref)[ref != 0].max() <= 2.0
    assert got[len(x) - 3] == 0.0
    assert np.abs(got - np.log(x)).max() <= 2 * np.spacing(np.abs(np.log(x))).max()
    a = rng.standard_normal(100000) * np.exp(rng.uniform(-50, 50, 100000))
    b = rng.standard_normal(100000) * np.exp(rng.uniform(-50, 50, 100000))
    assert ulps(run(4, b), 1.0 / b).max() <= 1.0
    assert ulps(run(5, a, b), a / b).max() <= 1.0


def test_exp_q_table():
    """exp(-x^2/4) from the 32-entry 2^(j/32) table (hot kernels): <= 2 ulp, NaN in -> NaN out, +-inf -> 0."""
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-12, 12, 100000), rng.standard_normal(100000), np.linspace(-60, 60, 4001),
                        [0.0, -0.0, 1e-200, 54.0, -54.0]])
    got = run(6, x)
    mp.mp.dps = 30
    sub = np.r_[0:3000, 100000:103000, 200000:204006]
    # (relative to exp of the ROUNDED argument -x*x/4, as NumPy's np.exp(-x**2/4) in the reference is)
    ref = np.array([float(mp.exp(mp.mpf(float(-0.25 * (v * v))))) for v in x[sub]])
    ok = ref > 1e-300
    assert ulps(got[sub][ok], ref[ok]).max() <= 2.0
    assert np.abs(got[sub][~ok]).max(initial=0.0) < 1e-299
    assert np.isnan(run(6, [np.nan]))[0]
    assert run(6, [np.inf])[0] == 0.0 and run(6, [-np.inf])[0] == 0.0 and run(6, [1e200])[0] == 0.0


def test_exp_q_fast():
    """exp(-x^2/4) of the forward hot kernels (degree-12 series): <= 2 ulp, NaN in -> NaN out, underflow to 0."""
    rng = np.random.default_rng(4)
    x = np.concatenate([rng.uniform(-12, 12, 100000), rng.standard_normal(100000), np.linspace(-60, 60, 4001),
                        [0.0, -0.0, 1e-200, 54.0, -54.0]])
    got = run(7, x)
    mp.mp.dps = 30
    sub = np.r_[0:3000, 100000:103000, 200000:204006]
    ref = np.array([float(mp.exp(mp.mpf(float(-0.25 * (v * v))))) for v in x[sub]])
    ok = ref > 1e-300
    assert ulps(got[sub][ok], ref[ok]).max() <= 2.0
    assert np.abs(got[sub][~ok]).max(initial=0.0) < 1e-299
    assert np.isnan(run(7, [np.nan]))[0]
    assert run(7, [1e100])[0] == 0.0 and run(7, [100.0])[0] == 0.0 and run(7, [-1e8])[0] == 0.0
