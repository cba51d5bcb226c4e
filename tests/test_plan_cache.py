"""
The statically planned column cache of the fast kernels (termtable._plan_column_cache, PlanCache in
csrc/ttm_eval.h): the plan is replayed in Python to check its invariants, and the planned evaluator is
compared with the run-time tagged one (TTM_NO_PLAN) and on sweeps that start in the middle of the map
(conditional inverse / component shards), where the recorded entry state has to be preloaded.
"""
import os

import numpy as np
import pytest

from tests.hostemu.emu import EmuMap
from tests.test_hostemu_vs_oracle import build
from triangular_transport_toolbox_amd import specs, termtable

FD = termtable.FD_LEN
W = termtable.PLAN_WAYS


def replay(cm, k0=0):
    """Run the plan like the kernel does; returns the number of column loads.  Asserts that a planned hit
    finds the right column (and exp flag) in the slot."""
    fd, fi = cm.fdesc, cm.fints
    slots = [None] * W                     # [column, e_valid]
    po = fd[k0 * FD + 14]
    for w in range(W):
        v = int(fi[po + w])
        if v >= 0:
            slots[w] = [v & ~termtable.PLAN_E, bool(v & termtable.PLAN_E)]
    loads = 0
    for k in range(k0, cm.D):
        d = fd[k * FD:(k + 1) * FD]
        if d[5]:
            continue
        # entry state recorded for this component must be what the replay holds
        po = d[14]
        want = [(-1 if s is None else (s[0] | (termtable.PLAN_E if s[1] else 0))) for s in slots]
        assert list(fi[po:po + W]) == want
        for g in range(d[1]):
            var, P, aoff, fl = fi[d[6] + 4 * g:d[6] + 4 * g + 4]
            slot = (fl >> 8) & 255
            hf = bool(fl & termtable.PLAN_HF)
            if fl & termtable.PLAN_XHIT:
                assert slots[slot] is not None and slots[slot][0] == var
            else:
                loads += 1
                if slot != 255:
                    slots[slot] = [var, False]
            if hf:
                if fl & termtable.PLAN_EHIT:
                    assert fl & termtable.PLAN_XHIT and slots[slot][1]
                elif slot != 255:
                    slots[slot][1] = True
            assert var < d[0]
        if d[13] >= 0:
            slots[d[13]] = [int(d[0]), False]
    return loads


@pytest.mark.parametrize('spec', ['c5', 'dense', 'entf', 'c3'])
def test_plan_replay_invariants(spec):
    if spec == 'c5':
        mon, non = specs.banded_separable_spec(40, band=2)
        d = 40
    elif spec == 'dense':
        mon, non = specs.dense_separable_spec(12, 3)
        d = 12
    elif spec == 'entf':
        mon, non = specs.entf_filter_spec(3)
        d = 4
    else:
        mon, non = specs.banded_separable_spec(10, band=6)
        d = 10
    cm = termtable.compile_map(mon, non, d, 'hermite function', 'separable monotonicity')
    loads0 = replay(cm)
    for k0 in range(1, cm.D):
        replay(cm, k0)
    if spec == 'c5':
        # band 2: x_{k-1} and x_{k-2} are always served from the cache (kept when they were the own column)
        assert loads0 == 0
        assert all(cm.fdesc[k * FD + 13] >= 0 for k in range(cm.D - 1))


def _run(em, coef, X, k0=0, k1=None):
    return em.forward(coef, X, k0, k1)


@pytest.mark.parametrize('name', ['c5_sep', 'c3_sep', 'c2b_sep', 'c5_int', 'misc_family_hermite'])
def test_planned_equals_tagged_and_partial_sweeps(name):
    npz, desc, om, cm, em, special = build(name)
    if (cm.complex & 1).any():
        pytest.skip('map has components that need the generic interpreter')
    from tests.util import coeff_lists
    mon, non = coeff_lists(npz, cm.D)
    coef = em.pack(non, mon)
    rng = np.random.default_rng(5)
    X = rng.standard_normal((64, cm.d_cols))
    # direct kernels (U-form off): statically planned column cache == run-time tagged cache, bit for bit
    os.environ['TTM_NO_UFORM'] = '1'
    try:
        Zd, ldd = _run(em, coef, X)
        os.environ['TTM_NO_PLAN'] = '1'
        try:
            Zt, ldt = _run(em, coef, X)
        finally:
            del os.environ['TTM_NO_PLAN']
    finally:
        del os.environ['TTM_NO_UFORM']
    assert np.array_equal(Zd, Zt) and np.array_equal(ldd, ldt)
    # default dispatch (U-form when the map has one): same values to rounding, same planned cache
    Zp, ldp = _run(em, coef, X)
    assert np.max(np.abs(Zp - Zd) / (1 + np.abs(Zd))) < 1e-12
    ok = np.isfinite(ldd)
    assert np.array_equal(np.isfinite(ldp), ok) and np.max(np.abs(ldp[ok] - ldd[ok]) / (1 + np.abs(ldd[ok])), initial=0) < 1e-10
    # sweeps starting at every k0: same columns as the full sweep (the entry state is preloaded)
    for k0 in range(1, cm.D, max(1, cm.D // 7)):
        Zs, _ = _run(em, coef, X, k0, cm.D)
        assert np.array_equal(Zs, Zp[:, k0:])
        Zs, _ = _run(em, coef, X, k0, min(cm.D, k0 + 2))
        assert np.array_equal(Zs, Zp[:, k0:k0 + 2])
