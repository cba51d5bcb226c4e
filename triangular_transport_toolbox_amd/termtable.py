"""
Term-table compiler: turns the reference's ``monotone`` / ``nonmonotone``
specification lists into the flat int32 / fp64 tables the HIP kernels
interpret (layout: include/ttm.h, "int32 layout of one component block").

The reference generates NumPy source strings from these lists and ``exec``s
them (transport_map.py:823-1261 ``write_basis_function``, 1263-1856
``function_constructor_alternative``, 1860-2134 derivative constructor).  No
code is generated here; the same *semantics* are encoded as data:

* an entry ``[]`` is the constant 1 (TM:885-898);
* an int list (optional ``'HF'``) is a product over its unique variables, in
  ascending order, of the top-order polynomial of the counted order
  (TM:1064, 1096-1160); with ``'HF'`` every factor is the normalised Hermite
  function a_n P_n(x) exp(-x^2/4) with a_n from TM:1102-1109;
* a string ``'LET j' | 'RET j' | 'RBF j' | 'iRBF j'`` is a special term whose
  (centre, scale) index is the running count of special terms on variable j in
  list order (TM:1337, 1417-1427; separate counter for the nonmonotone list
  TM:1623, 1686-1694); monotone special terms on a variable other than the
  component's own live under 'cross-terms' (TM:1407-1412);
* a component whose monotone list has cross-term special terms gets all its
  monotone special terms removed and appended as the full tensor grid, lowest
  variable outermost (TM:1446-1483);
* term order = coefficient order.

Every monotone term is split into an "A" part (factors on columns other than
the component's own column kc) and at most one "B" function of x_kc alone;
distinct B functions are listed once per component so that the kernels can
evaluate g(t) = sum_b w_b B_b(t) at many t (quadrature nodes, bisection
trials, table points) from per-sample weights w_b.
"""

import itertools

import math

import numpy as np

# constants mirrored from include/ttm.h
KIND_POLY, KIND_HF, KIND_LET, KIND_RET, KIND_RBF, KIND_IRBF = 1, 2, 3, 4, 5, 6
ST_KINDS = {'let': KIND_LET, 'ret': KIND_RET, 'rbf': KIND_RBF, 'irbf': KIND_IRBF}
FAM_HERMITE_E, FAM_POWER, FAM_HERMITE, FAM_CHEBYSHEV, FAM_LAGUERRE, FAM_LEGENDRE = range(6)
HDR_LEN = 32
(HDR_KC, HDR_N_NM, HDR_OFF_NM, HDR_N_MON, HDR_OFF_MON, HDR_OFF_FAC, HDR_NB, HDR_OFF_B, HDR_NB_HF, HDR_NB_POLY,
 HDR_NB_ST, HDR_MAXP_HF, HDR_MAXP_POLY, HDR_FLAGS, HDR_N_DPAR, HDR_LEN_BLK, HDR_N_GRP, HDR_OFF_GRP, HDR_N_GEN,
 HDR_OFF_GEN, HDR_N_MNT, HDR_OFF_MNT, HDR_N_FOLD, HDR_OFF_FSLOT, HDR_OFF_FSRC, HDR_OFF_WB, HDR_N_XGRP, HDR_OFF_XGRP,
 HDR_OFF_XPROG, HDR_X_NROW, HDR_X_NSUM, HDR_X_RESERVED) = range(32)
X_MAXF, X_NU_MAX, X_SUM_MAX, XR_ONE, XR_S, XR_J, XR_U = 3, 40, 192, 0, 1, 2, 3      # "X program" (include/ttm.h)
ST_NPAR = 5   # centre, scale, 1/(sqrt2 scale), scale sqrt(2/pi), 1/(sqrt(2 pi) scale)

# polynomial_type -> (family id, numpy class, unified name)   (TM:274-304)
_P = np.polynomial
FAMILIES = {
    'standard': (FAM_POWER, _P.polynomial.Polynomial),
    'polynomial': (FAM_POWER, _P.polynomial.Polynomial),
    'power series': (FAM_POWER, _P.polynomial.Polynomial),
    'hermite': (FAM_HERMITE, _P.hermite.Hermite),
    "phycisist's hermite": (FAM_HERMITE, _P.hermite.Hermite),
    'phycisists hermite': (FAM_HERMITE, _P.hermite.Hermite),
    'hermite_e': (FAM_HERMITE_E, _P.hermite_e.HermiteE),
    "probabilist's hermite": (FAM_HERMITE_E, _P.hermite_e.HermiteE),
    'probabilists hermite': (FAM_HERMITE_E, _P.hermite_e.HermiteE),
    'chebyshev': (FAM_CHEBYSHEV, _P.chebyshev.Chebyshev),
    'laguerre': (FAM_LAGUERRE, _P.laguerre.Laguerre),
    'legendre': (FAM_LEGENDRE, _P.legendre.Legendre),
    'hermite function': (FAM_HERMITE_E, _P.hermite_e.HermiteE),
    'hermite_function': (FAM_HERMITE_E, _P.hermite_e.HermiteE),
    'hermite functions': (FAM_HERMITE_E, _P.hermite_e.HermiteE),
}

_HF_CACHE = {}


def hf_constant(polyclass, n):
    """a_n = 1 / max |P_n(x) exp(-x^2/4)| over linspace(-100, 100, 100001), the
    same NumPy evaluation as TM:1102-1109 (hence bit-identical constants)."""
    key = (polyclass.__name__, int(n))
    if key not in _HF_CACHE:
        hf_x = np.linspace(-100, 100, 100001)
        hfeval = polyclass([0.] * int(n) + [1.])(hf_x) * np.exp(-hf_x ** 2 / 4)
        _HF_CACHE[key] = float(1 / np.max(np.abs(hfeval)))
    return _HF_CACHE[key]


def gauss_legendre(order):
    """Quadrature rule exactly as TM:199-225 (legroots + derivative formula)."""
    coefs = [0] * int(order) + [1]
    coefs_der = np.polynomial.legendre.legder(coefs)
    LegendreDer = np.polynomial.legendre.Legendre(coefs_der)
    xis = np.polynomial.legendre.legroots(coefs)
    Ws = 2.0 / ((1.0 - xis ** 2) * (LegendreDer(xis) ** 2))
    return xis, Ws


def count_special_terms(monotone, nonmonotone, skip):
    """TM:2136-2217: special-term counters per (component column, variable);
    monotone special terms on other variables are counted under 'cross-terms'."""
    special = {}
    for k in range(len(monotone)):
        kc = k + skip
        st = special[kc] = {}
        for entry in nonmonotone[k]:
            if isinstance(entry, str):
                index = int(entry.split(' ')[1])
                st.setdefault(index, {'counter': 0, 'centers': np.asarray([]), 'scales': np.asarray([])})['counter'] += 1
        for entry in monotone[k]:
            if isinstance(entry, str):
                index = int(entry.split(' ')[1])
                tgt = st if index == kc else st.setdefault('cross-terms', {})
                tgt.setdefault(index, {'counter': 0, 'centers': np.asarray([]), 'scales': np.asarray([])})['counter'] += 1
    return special


def place_special_terms(special, column_quantiles, scale_factor, scale_mode):
    """TM:2219-2330.  ``column_quantiles(var, q_array)`` returns np.quantile of the
    standardised training column ``var`` (method 'linear')."""
    def place(dictionary):
        for d in [key for key in dictionary if key != 'cross-terms']:
            n = dictionary[d]['counter']
            if n == 1:
                dictionary[d]['centers'] = np.asarray([column_quantiles(d, np.asarray([0.5]))[0]])
                dictionary[d]['scales'] = np.asarray([scale_factor / 2 if scale_mode == 'dynamic' else scale_factor])
            elif n > 1:
                q = np.arange(1, n + 1, 1) / (n + 1)
                c = dictionary[d]['centers'] = np.array(column_quantiles(d, q), dtype=float)
                scales = np.zeros(n)
                if scale_mode == 'dynamic':
                    for i in range(n):
                        if i == 0:
                            scales[i] = (c[1] - c[0]) * scale_factor
                        elif i == n - 1:
                            scales[i] = (c[i] - c[i - 1]) * scale_factor
                        else:
                            scales[i] = (c[i + 1] - c[i - 1]) / 2 * scale_factor
                    dictionary[d]['scales'] = scales
                else:
                    dictionary[d]['scales'] = scales + scale_factor
        return dictionary
    for kc in special:
        if 'cross-terms' in special[kc]:
            special[kc]['cross-terms'] = place(special[kc]['cross-terms'])
        special[kc] = place(special[kc])
    return special


def quantile_requests(special):
    """All (variable, quantile) pairs place_special_terms will ask for."""
    req = {}
    for kc, d in special.items():
        groups = [d] + ([d['cross-terms']] if 'cross-terms' in d else [])
        for g in groups:
            for var, v in g.items():
                if var == 'cross-terms':
                    continue
                n = v['counter']
                qs = [0.5] if n == 1 else list(np.arange(1, n + 1, 1) / (n + 1))
                req.setdefault(var, set()).update(float(q) for q in qs)
    return {var: sorted(q) for var, q in req.items()}


class CompiledMap:
    """Result of compile_map: flat tables + bookkeeping."""

    def __init__(self):
        self.itab = None
        self.dpar = None
        self.comp_off = None
        self.dpar_off = None
        self.coef_off = None
        self.nslots = None
        self.nb1 = None
        self.fold_off = None
        self.ftab = None
        self.ftab_off = None
        self.fdesc = None
        self.fints = None
        self.n_nm = None
        self.n_mon = None
        self.dpar_sources = []     # per dpar entry: ('hf', value) | ('st', kc, cross, var, index, which)
        self.descriptors_mon = []  # canonical term descriptors (tests)
        self.descriptors_nonmon = []
        self.sep_direct = []       # per component: (column of x_k, [(special-term kind, dpar offset) per coefficient]) or None
        self.bounds = []           # L-BFGS-B bounds per component (TM:1891-1892, 1925-1929)
        self.family = 0
        self.D = 0
        self.d_cols = 0

    def fill_special_terms(self, special):
        """Refresh the special-term constants after a new placement: per special
        term {centre, scale, 1/(sqrt2 scale), scale sqrt(2/pi), 1/(sqrt(2 pi) scale)}."""
        # (the special-term records are groups of five consecutive entries of one term - which = 0..4 -: walked once per term,
        # the list of groups kept; the filter refreshes them three times per cycle)
        groups = getattr(self, '_st_groups', None)
        if groups is None:
            groups = self._st_groups = [(i, src[1:5]) for i, src in enumerate(self.dpar_sources) if src[0] == 'st' and src[5] == 0]
            whole = all(i + 4 < len(self.dpar_sources) and
                        all(self.dpar_sources[i + w][0] == 'st' and tuple(self.dpar_sources[i + w][1:]) == tuple(key) + (w,) for w in range(5))
                        for i, key in groups)
            count = sum(1 for src in self.dpar_sources if src[0] == 'st')
            if not whole or count != 5 * len(groups):
                groups = self._st_groups = False
        if groups is False:
            for i, src in enumerate(self.dpar_sources):
                if src[0] == 'st':
                    _, kc, cross, var, index, which = src
                    d = special[kc]['cross-terms'][var] if cross else special[kc][var]
                    mu, sc = float(d['centers'][index]), float(d['scales'][index])
                    self.dpar[i] = (mu, sc, 1.0 / (np.sqrt(2) * sc), sc * np.sqrt(2 / np.pi),
                                    1.0 / (np.sqrt(2 * np.pi) * sc))[which]
            return self.dpar
        r2, r2pi, rpi = np.sqrt(2), np.sqrt(2 * np.pi), np.sqrt(2 / np.pi)
        for i, (kc, cross, var, index) in groups:
            d = special[kc]['cross-terms'][var] if cross else special[kc][var]
            mu, sc = float(d['centers'][index]), float(d['scales'][index])
            self.dpar[i:i + 5] = (mu, sc, 1.0 / (r2 * sc), sc * rpi, 1.0 / (r2pi * sc))
        return self.dpar


def _parse_entry(entry, kc, which, counter, polynomial_type, linearization=None):
    """One list entry -> list of factors.
    factor = ('poly', var, order, hf) | ('st', kind, var, cross, index)."""
    if isinstance(entry, str):
        parts = entry.split(' ')
        kind = parts[0].lower()
        if kind not in ST_KINDS:
            raise ValueError("Special term '" + str(parts[0]) + "' not understood. Currently, only LET, RET, iRBF, "
                             "and RBF are implemented.")
        var = int(parts[1])
        cross = (which == 'mon') and (var != kc)
        idx = counter.get(var, 0)
        counter[var] = idx + 1
        return [('st', kind, var, cross, idx)]
    if len(entry) == 0:
        return []
    hf = any(e == 'HF' for e in entry)
    if any(e == 'LIN' for e in entry) and linearization is None:            # TM:1053-1054
        raise Exception("'LIN' modifier specified in variable monotone, but the variable linearization is defined as "
                        "None. Please specify a scalar linearization or remove the 'LIN' modifier.")
    # (a 'LIN' factor evaluates to the plain factor: transport_map._linearization_thresholds)
    ints = [e for e in entry if not isinstance(e, str)]
    ui, ct = np.unique(ints, return_counts=True)
    return [('poly', int(u), int(c), bool(hf)) for u, c in zip(ui, ct)]


def _descriptor(term):
    if len(term) == 0:
        return [['const']]
    out = []
    for f in term:
        if f[0] == 'poly':
            out.append(['poly', f[1], f[2], f[3]])
        else:
            out.append(['st', f[1], f[2], f[3], f[4]])
    return out


I_PMAX = 10                   # TTM_I_PMAX (csrc/ttm_dense_table.h)


def dense_class(max_ph, max_pp):
    """Order class (PH, PP) of the monomial-form kernels for a component whose x_k-functions reach these orders
    (csrc/ttm_dense.h: dense_class_of); None when there is none."""
    if max_ph > I_PMAX or max_pp > I_PMAX or (max_ph == 0 and max_pp == 0):
        return None
    if max_pp == 0:
        return (3 if max_ph <= 3 else (5 if max_ph <= 5 else (8 if max_ph <= 8 else 10)), 0)
    if max_ph == 0:
        return (0, 10)
    return (10, 10)


def _x_program(terms, facs, bfuns, n_nm, kc, polyclass, dp_const, dp_hf_value, fold_slots):
    """X program of one component (include/ttm.h, csrc/ttm_xprog.h); None when the component cannot have one (a factor on
    another column that is a special term, more than X_MAXF such factors in a term, a factor of higher order than the
    component's order class, too many distinct values or sums).
    terms: records [f0, nf, b, ci] (nonmonotone, then monotone, each in coefficient order); facs: [var, kind, order, p0];
    bfuns: [kind, order, p0, 0].  dp_const(value) -> index of a constant in the component's dpar slice; fold_slots: the fold
    recipe under construction - the X section (CN | HC | HP) is appended to it.
    Returns dict(ints=[...], n_sum=...)."""
    nB = len(bfuns)
    max_ph = max([b[1] for b in bfuns if b[0] == KIND_HF], default=0)
    max_pp = max([b[1] for b in bfuns if b[0] == KIND_POLY], default=0)
    cls = dense_class(max_ph, max_pp)
    if cls is None:
        return None
    PH, PP = cls
    PM = max(PH, PP)
    used = []
    for f0, nf, _, _ in terms:
        if nf > X_MAXF:
            return None
        for f in facs[f0:f0 + nf]:
            if f[1] not in (KIND_POLY, KIND_HF) or f[0] == kc or f[2] > PM:
                return None
            key = (f[0], 0 if f[1] == KIND_POLY else 1, f[2])
            if key not in used:
                used.append(key)
    if len(used) > X_NU_MAX:
        return None
    used.sort()                                            # by column, plain polynomials before Hermite functions, by order
    cols = sorted(set(u[0] for u in used))
    ucol = {u: XR_U + i for i, u in enumerate(used)}
    # computed products (two or three factors), distinct
    prods = []

    def acol_of(f0, nf):
        cs = sorted(ucol[(f[0], 0 if f[1] == KIND_POLY else 1, f[2])] for f in facs[f0:f0 + nf])
        if len(cs) == 0:
            return XR_ONE
        if len(cs) == 1:
            return cs[0]
        key = tuple(cs)
        if key not in prods:
            prods.append(key)
        return XR_U + len(used) + prods.index(key)
    a_nm, a_mon, trec = [], [], []
    for which, (f0, nf, b, ci) in enumerate(terms):
        col = acol_of(f0, nf)
        lst = a_nm if which < n_nm else a_mon
        if col not in lst:
            lst.append(col)
        trec.append([lst.index(col), -1 if which < n_nm else (b if b >= 0 else nB), 0, 0])
    NH = PH + 1 if PH > 0 else 0
    NQ = NH + PP + 1
    n_sum = 1 + len(a_nm) + len(a_mon) * NQ
    if n_sum > X_SUM_MAX:
        return None
    # constants: Horner rows of the U values (a_n folded in), stride I_PMAX + 1
    LD = I_PMAX + 1

    def mono_row(order):
        c = np.zeros(LD)
        cc = polyclass.basis(order).convert(kind=np.polynomial.Polynomial).coef
        c[:len(cc)] = cc
        return c
    urow_first = None
    for (_, is_hf, order) in used:
        row = mono_row(order) * (dp_hf_value(order) if is_hf else 1.0)
        idx = [dp_const(float(v), force_new=True) for v in row]
        urow_first = idx[0] if urow_first is None else urow_first
        assert idx == list(range(idx[0], idx[0] + LD))
    # fold X section: CN[a] | HC[a][0..LD) | HP[a][0..LD)
    off_x = len(fold_slots)
    fold_slots.extend([] for _ in range(len(a_nm) + 2 * LD * len(a_mon)))
    for which, (f0, nf, b, ci) in enumerate(terms):
        a = trec[which][0]
        if which < n_nm:
            fold_slots[off_x + a].append((ci, -1))
            continue
        gci = n_nm + ci
        hc0 = off_x + len(a_nm) + a * LD
        hp0 = off_x + len(a_nm) + LD * len(a_mon) + a * LD
        if b < 0:
            fold_slots[hp0].append((gci, -1))
        elif bfuns[b][0] == KIND_HF:
            row = mono_row(bfuns[b][1]) * dp_hf_value(bfuns[b][1])
            for j in range(PH + 1):
                if row[j] != 0.0:
                    fold_slots[hc0 + j].append((gci, dp_const(float(row[j]))))
        else:
            row = mono_row(bfuns[b][1])
            for j in range(PP + 1):
                if row[j] != 0.0:
                    fold_slots[hp0 + j].append((gci, dp_const(float(row[j]))))
    ints = [len(cols), len(used), len(prods), len(a_nm), len(a_mon), off_x, urow_first if urow_first is not None else 0, PH | (PP << 8)]
    for c in cols:
        mine = [u for u in used if u[0] == c]
        ints += [c, sum(1 for u in mine if u[1] == 0), sum(1 for u in mine if u[1] == 1), 0]
    for key in prods:
        ints += list(key) + [XR_ONE] * (4 - len(key))
    ints += a_nm + [0] * ((-len(a_nm)) % 4)
    ints += a_mon + [0] * ((-len(a_mon)) % 4)
    for r in trec:
        ints += r
    return dict(ints=ints, n_sum=n_sum, n_u=len(used), n_prod=len(prods), n_var=len(cols))


PLAN_WAYS = 4            # TTM_PLAN_WAYS
PLAN_HF, PLAN_XHIT, PLAN_EHIT, PLAN_E = 1, 2, 4, 1 << 30
FD_LEN, FD_KC_SLOT = 16, 13


def _plan_column_cache(plan_seq, fdesc, fints, ways=None):
    """Static schedule of the per-thread column cache of the fast kernels.

    A sweep over the components touches the columns in an order that is known when the map is compiled:
    per component the columns of its univariate nonmonotone groups, then its own column x_kc (which
    later components read).  So hit / miss / replacement are decided here, with Belady's rule (evict the
    column whose next use is farthest away, bypass when the new column's next use is farther than all
    of them) instead of run-time tag compares: each group record gets a flag word (TTM_PLAN_*: slot,
    column already there, exp(-x^2/4) already there), each component the slot for its own column, and
    the cache contents on entry to every component are recorded so that a sweep can start anywhere
    (conditional inverse, component shards): the kernel preloads that state.  Components that need the
    generic interpreter do not use the planned cache; they leave the state untouched.

    ways = None: the number of ways (<= PLAN_WAYS) is chosen here - the smallest one whose column loads
    stay within 10 % (+2) of what PLAN_WAYS ways need - because the cache lives in LDS and fewer ways mean
    more resident waves.  Returns (ways, column loads of a full sweep)."""
    if ways is None:
        trial = {}
        for w in range(1, PLAN_WAYS + 1):
            trial[w] = _plan_column_cache(plan_seq, list(fdesc), list(fints), w)[1]
        ways = next(w for w in range(1, PLAN_WAYS + 1) if trial[w] <= 1.1 * trial[PLAN_WAYS] + 2)
    # flat access list: (component, column, wants_e, fints index of the flag word or None for the own column)
    acc = []
    for k, (groups, kc, fint_off) in enumerate(plan_seq):
        if groups is None:
            continue
        for gi, (var, hf) in enumerate(groups):
            acc.append((k, var, hf, fint_off + 4 * gi + 3))
        acc.append((k, kc, False, None))
    # next use of the same column after access i (reads only: the own-column put is not a use)
    nxt, last = [None] * len(acc), {}
    for i in range(len(acc) - 1, -1, -1):
        k, var, hf, fi = acc[i]
        nxt[i] = last.get(var, 1 << 60)
        if fi is not None:
            last[var] = i
    slots = [None] * ways               # [column, e_valid, next use]
    loads = 0
    state_at = {}
    cur_k = -1

    def snapshot(k_from, k_to):
        for kk in range(k_from, k_to):
            state_at[kk] = ([(-1 if sl is None else (sl[0] | (PLAN_E if sl[1] else 0))) for sl in slots] +
                            [-1] * (PLAN_WAYS - ways))

    def place(var, nuse):
        """slot for a column that is not cached, or 255 to bypass"""
        for w in range(ways):
            if slots[w] is None:
                return w
        far = max(range(ways), key=lambda w_: slots[w_][2])
        if slots[far][2] <= nuse:
            return 255
        return far

    for i, (k, var, hf, fi) in enumerate(acc):
        if k != cur_k:
            snapshot(cur_k + 1, k + 1)
            cur_k = k
        where = next((w for w in range(ways) if slots[w] is not None and slots[w][0] == var), None)
        if fi is None:                  # own column: keep it if somebody reads it later
            if where is not None:       # (cannot happen: groups only read columns < kc)
                slots[where] = None
            w = place(var, nxt[i]) if nxt[i] < (1 << 60) else 255
            if w != 255:
                slots[w] = [var, False, nxt[i]]
            fdesc[k * FD_LEN + FD_KC_SLOT] = -1 if w == 255 else w
            continue
        flags = PLAN_HF if hf else 0
        if where is not None:
            flags |= PLAN_XHIT | (where << 8)
            if hf and slots[where][1]:
                flags |= PLAN_EHIT
            slots[where][1] = slots[where][1] or hf
            slots[where][2] = nxt[i]
        else:
            loads += 1
            w = place(var, nxt[i]) if nxt[i] < (1 << 60) else 255
            flags |= w << 8
            if w != 255:
                slots[w] = [var, hf, nxt[i]]
        fints[fi] = flags
    snapshot(cur_k + 1, len(plan_seq))
    for k, (groups, kc, fint_off) in enumerate(plan_seq):
        po = fdesc[k * FD_LEN + 14]
        fints[po:po + PLAN_WAYS] = state_at[k]
    return ways, loads


def compile_map(monotone, nonmonotone, d_cols, polynomial_type='hermite function',
                monotonicity='integrated rectifier', linearization=None):
    """Compile the specification lists into device tables (special-term
    constants are filled in later by CompiledMap.fill_special_terms)."""
    if polynomial_type.lower() not in FAMILIES:
        raise Exception("Polynomial type not understood. The variable polynomial_type should be either 'power series', "
                        "'hermite', 'hermite_e', 'chebyshev', 'laguerre', or 'legendre'.")
    family, polyclass = FAMILIES[polynomial_type.lower()]
    separable = monotonicity.lower() == 'separable monotonicity'
    D = len(monotone)
    skip = d_cols - D
    if skip < 0:
        raise ValueError('X has fewer columns than the map has components')
    cm = CompiledMap()
    cm.family, cm.D, cm.d_cols = family, D, d_cols
    itab, dpar = [], []
    comp_off, dpar_off, coef_off, fold_off, ftab_off = [0], [0], [0], [0], [0]
    ftab, fdesc, fints = [], [], []
    plan_seq, complex_all, u_info = [], [], []
    nslots, nb1, n_nm_all, n_mon_all = [], [], [], []

    for k in range(D):
        kc = k + skip
        # ---- parse --------------------------------------------------------
        counter = {}
        nm_terms = [_parse_entry(e, kc, 'nonmon', counter, polynomial_type, linearization) for e in nonmonotone[k]]
        counter = {}
        mon_terms, st_idx = [], []
        for i, e in enumerate(monotone[k]):
            t = _parse_entry(e, kc, 'mon', counter, polynomial_type, linearization)
            mon_terms.append(t)
            if isinstance(e, str):
                st_idx.append(i)
        has_cross = any(t[0][3] for t in (mon_terms[i] for i in st_idx))
        if has_cross:
            if separable:
                raise NotImplementedError('monotone cross-term special terms are not supported with separable '
                                          'monotonicity (their x_k-derivative is identically zero, TM:2000-2037)')
            rbf = [mon_terms[i] for i in st_idx]
            dims = sorted(set(t[0][2] for t in rbf))
            by_dim = {d: [t for t in rbf if t[0][2] == d] for d in dims}
            grid = list(by_dim[dims[0]])
            for d in dims[1:]:
                grid = [a + b for a, b in itertools.product(grid, by_dim[d])]
            mon_terms = [t for i, t in enumerate(mon_terms) if i not in st_idx] + grid
        # ---- validate -------------------------------------------------------
        for t in nm_terms:
            for f in t:
                var = f[1] if f[0] == 'poly' else f[2]
                if var < 0 or var >= kc:
                    raise ValueError('nonmonotone term of component %d depends on column %d; only columns < %d are '
                                     'allowed' % (k, var, kc))
        for t in mon_terms:
            for f in t:
                var = f[1] if f[0] == 'poly' else f[2]
                if var < 0 or var > kc:
                    raise ValueError('monotone term of component %d depends on column %d > %d' % (k, var, kc))
        if separable:
            keys = set()
            for t in mon_terms:
                polys = [f for f in t if f[0] == 'poly']
                if any(f[3] for f in polys) and len(t) > 1:
                    raise NotImplementedError('Hermite-function cross terms in a separable monotone list: the '
                                              'reference derivative is ill-defined (TM:1245 overwrites factors)')
                for f in polys:
                    if f[1] == kc:
                        keys.add((f[2], f[3]))
            if len(set(o for o, _ in keys)) != len(keys):
                raise NotImplementedError('the same polynomial order appears with and without HF in a separable '
                                          'monotone list (reference shares the derivative key, TM:1172/1201)')
        # ---- constants ------------------------------------------------------
        dp_local, dp_index = [], {}

        def dp_hf(order):
            key = ('hf', order)
            if key not in dp_index:
                dp_index[key] = len(dp_local)
                dp_local.append(('hf', hf_constant(polyclass, order)))
            return dp_index[key]

        def dp_const(value, force_new=False):
            key = ('const', value)
            if force_new or key not in dp_index:
                if not force_new:
                    dp_index[key] = len(dp_local)
                dp_local.append(('hf', value))
                return len(dp_local) - 1
            return dp_index[key]

        def dp_st(f):
            key = ('st', f[3], f[2], f[4])
            if key not in dp_index:
                dp_index[key] = len(dp_local)
                for which in range(ST_NPAR):
                    dp_local.append(('st', kc, f[3], f[2], f[4], which))
            return dp_index[key]

        def fac_record(f):
            if f[0] == 'poly':
                return [f[1], KIND_HF if f[3] else KIND_POLY, f[2], dp_hf(f[2]) if f[3] else 0]
            return [f[2], ST_KINDS[f[1]], 0, dp_st(f)]

        # ---- B functions (of x_kc alone) --------------------------------------
        bkeys = []
        for t in mon_terms:
            for f in t:
                var = f[1] if f[0] == 'poly' else f[2]
                if var == kc:
                    key = ('hf' if f[3] else 'poly', f[2]) if f[0] == 'poly' else ('st', f[1], f[4])
                    if key not in bkeys:
                        bkeys.append(key)
        b_hf = sorted([b for b in bkeys if b[0] == 'hf'], key=lambda b: b[1])
        b_poly = sorted([b for b in bkeys if b[0] == 'poly'], key=lambda b: b[1])
        b_st = [b for b in bkeys if b[0] == 'st']
        blist = b_hf + b_poly + b_st
        bfuns = []
        for b in blist:
            if b[0] == 'hf':
                bfuns.append([KIND_HF, b[1], dp_hf(b[1]), 0])
            elif b[0] == 'poly':
                bfuns.append([KIND_POLY, b[1], 0, 0])
            else:
                bfuns.append([ST_KINDS[b[1]], 0, dp_st(('st', b[1], kc, False, b[2])), 0])
        # ---- term + factor records ------------------------------------------
        facs, terms = [], []
        for ci, t in enumerate(nm_terms):
            f0 = len(facs)
            facs.extend(fac_record(f) for f in t)
            terms.append([f0, len(t), -1, ci])
        all_trivial = True
        for ci, t in enumerate(mon_terms):
            f0 = len(facs)
            b = -1
            nf = 0
            for f in t:
                var = f[1] if f[0] == 'poly' else f[2]
                if var == kc:
                    key = ('hf' if f[3] else 'poly', f[2]) if f[0] == 'poly' else ('st', f[1], f[4])
                    b = blist.index(key)
                else:
                    facs.append(fac_record(f))
                    nf += 1
            if nf:
                all_trivial = False
            terms.append([f0, nf, b, ci])
        # ---- folded coefficients: constant, per-variable groups, B weights ---------
        # fold slot list: each slot = list of (coefficient index within [nonmon | mon], dpar multiplier or -1)
        fold_slots = [[]]                                   # slot 0: constant nonmonotone terms
        groups, gen_idx = [], []
        by_var = {}
        for ci, t in enumerate(nm_terms):
            if len(t) == 0:
                fold_slots[0].append((ci, -1))
            elif len(t) == 1 and t[0][0] == 'poly':
                by_var.setdefault(t[0][1], []).append((ci, t[0]))
            else:
                gen_idx.append(ci)
        for var in sorted(by_var):
            P = max(f[2] for _, f in by_var[var])
            has_hf = any(f[3] for _, f in by_var[var])
            off = len(fold_slots)
            fold_slots.extend([] for _ in range(2 * P))
            for ci, f in by_var[var]:
                if f[3]:
                    fold_slots[off + P + f[2] - 1].append((ci, dp_hf(f[2])))
                else:
                    fold_slots[off + f[2] - 1].append((ci, -1))
            groups.append([var, P, off, 1 if has_hf else 0])
        off_wb = len(fold_slots)
        fold_slots.extend([] for _ in range(len(bfuns) + 1))
        mnt_idx = []
        xby = {}                   # cross terms with ONE factor, polynomial or Hermite function: (b, var) -> [(ci, factor)]
        for ci, tr in enumerate(terms[len(nm_terms):]):
            cross = [f for f in mon_terms[ci] if (f[1] if f[0] == 'poly' else f[2]) != kc]
            if tr[1] == 0:
                fold_slots[off_wb + (tr[2] if tr[2] >= 0 else len(bfuns))].append((len(nm_terms) + ci, -1))
            elif tr[1] == 1 and cross[0][0] == 'poly':
                xby.setdefault((tr[2] if tr[2] >= 0 else len(bfuns), cross[0][1]), []).append((ci, cross[0]))
            else:
                mnt_idx.append(ci)
        # cross groups: the weight of B function b gets, per conditioning variable, a polynomial + Hermite-function
        # series in that variable whose coefficients are folded like the nonmonotone groups' (records {var, P, fold
        # offset, has_hf, b, 0, 0, 0}; alpha_n at offset + n - 1, beta_n = a_n c at offset + P + n - 1)
        xgroups = []
        for (b_, var) in sorted(xby):
            P = max(f[2] for _, f in xby[(b_, var)])
            has_hf = any(f[3] for _, f in xby[(b_, var)])
            off = len(fold_slots)
            fold_slots.extend([] for _ in range(2 * P))
            for ci, f in xby[(b_, var)]:
                if f[3]:
                    fold_slots[off + P + f[2] - 1].append((len(nm_terms) + ci, dp_hf(f[2])))
                else:
                    fold_slots[off + f[2] - 1].append((len(nm_terms) + ci, -1))
            xgroups.append([var, P, off, 1 if has_hf else 0, b_, 0, 0, 0])
        # "stream" section for the fast path (components without cross / generic terms): weights indexed
        # densely by polynomial order and one 5-double record per special-term B function
        # {w_b, centre, 1/(sqrt2 scale), scale sqrt(2/pi), 1/(sqrt(2 pi) scale)}; a source (-1, p) copies dpar[p]
        maxP_hf = max([b[1] for b in b_hf], default=0)
        maxP_poly = max([b[1] for b in b_poly], default=0)
        stream_rel = len(fold_slots)
        fold_slots.extend([] for _ in range(maxP_hf + maxP_poly + 5 * len(b_st)))
        for bi, b in enumerate(blist):
            trivial_src = list(fold_slots[off_wb + bi])
            if b[0] == 'hf':
                fold_slots[stream_rel + b[1] - 1] = [(ci_, dp_hf(b[1])) for ci_, _ in trivial_src]
            elif b[0] == 'poly':
                fold_slots[stream_rel + maxP_hf + b[1] - 1] = trivial_src
            else:
                si = bi - len(b_hf) - len(b_poly)
                rec = stream_rel + maxP_hf + maxP_poly + 5 * si
                p0 = bfuns[bi][2]
                fold_slots[rec] = trivial_src
                for j_, which in enumerate((0, 2, 3, 4)):
                    fold_slots[rec + 1 + j_] = [(-1, p0 + which)]
        # ---- X program (include/ttm.h): every factor value on another column once per sample, the monomial form of g and
        # the nonmonotone sum from folded matrices (their recipe joins the fold recipe), every objective / gradient sum a
        # product of two row entries
        poly_b_x = (not separable and len(b_st) == 0 and len(b_hf) + len(b_poly) > 0)
        xprog = _x_program(terms, facs, bfuns, len(nm_terms), kc, polyclass, dp_const,
                           lambda order: hf_constant(polyclass, order), fold_slots) if poly_b_x else None
        fslots, fsrc = [], []
        for sl in fold_slots:
            fslots.append([len(fsrc), len(sl)])
            fsrc.extend([list(e) for e in sl])
        # fast-path descriptor (TTM_FDESC_LEN int32) and int stream {per group: var, P, alpha offset, flags |
        # ST kinds | order of the unified ST records | planned-cache entry state}
        complex_comp = 1 if (len(gen_idx) or len(mnt_idx) or len(xgroups) or maxP_hf > 16 or maxP_poly > 16) else 0
        st_kinds = [bf[0] for bf in bfuns[len(b_hf) + len(b_poly):]]
        # unified special-term records: those whose VALUE needs the Gaussian (LET / RET / RBF) first
        st_order = ([i for i, kd in enumerate(st_kinds) if kd != KIND_IRBF] +
                    [i for i, kd in enumerate(st_kinds) if kd == KIND_IRBF])
        n_stA = sum(1 for kd in st_kinds if kd != KIND_IRBF)
        st8_rel = len(fslots) + ((-len(fslots)) % 8)
        fold_len = st8_rel + 8 + 8 * len(b_st)
        fint_off = len(fints)
        for g_ in groups:
            fints.extend(g_)
        fints.extend(st_kinds)
        fints.extend(st_order)
        fints.extend([0] * ((-len(fints)) % 4))
        plan_off = len(fints)
        fints.extend([-1] * PLAN_WAYS)
        fdesc.extend([kc, len(groups), len(b_st), maxP_hf, maxP_poly, complex_comp, fint_off, fold_off[-1], stream_rel,
                      len(bfuns), off_wb, st8_rel, n_stA, -1, plan_off, 0])
        plan_seq.append(([(g_[0], bool(g_[3])) for g_ in groups] if not complex_comp else None, kc, fint_off))
        u_info.append(dict(kc=kc, complex=complex_comp, fint_off=fint_off, stream_rel=stream_rel, maxP_hf=maxP_hf,
                           maxP_poly=maxP_poly,
                           groups=[(g_[0], g_[1], g_[2],
                                    max([f[2] for _, f in by_var[g_[0]] if f[3]], default=0),
                                    max([f[2] for _, f in by_var[g_[0]] if not f[3]], default=0)) for g_ in groups],
                           st_p0=[bf[2] for bf in bfuns[len(b_hf) + len(b_poly):]]))
        # ---- assemble the block ---------------------------------------------
        hdr = [0] * HDR_LEN
        off_nm = HDR_LEN
        off_mon = off_nm + 4 * len(nm_terms)
        off_fac = off_mon + 4 * len(mon_terms)
        off_b = off_fac + 4 * len(facs)
        off_grp = off_b + 4 * len(bfuns)
        off_gen = off_grp + 4 * len(groups)
        off_mnt = off_gen + len(gen_idx)
        off_xgrp = off_mnt + len(mnt_idx)
        xpad = (-(off_xgrp + 8 * len(xgroups))) % 4         # (records of four int32: 16-byte aligned within the block)
        off_xprog = off_xgrp + 8 * len(xgroups) + xpad
        blk_len = off_xprog + (len(xprog['ints']) if xprog else 0)
        off_fslot = 0                                       # fold recipes live in a separate table (ftab):
        off_fsrc = 2 * len(fslots)                          # they are read once at staging, not kept in LDS
        pad = (-blk_len) % 4                               # keep every block 16-byte aligned
        blk_len += pad
        hdr[HDR_KC] = kc
        hdr[HDR_N_NM], hdr[HDR_OFF_NM] = len(nm_terms), off_nm
        hdr[HDR_N_MON], hdr[HDR_OFF_MON] = len(mon_terms), off_mon
        hdr[HDR_OFF_FAC] = off_fac
        hdr[HDR_NB], hdr[HDR_OFF_B] = len(bfuns), off_b
        hdr[HDR_NB_HF], hdr[HDR_NB_POLY], hdr[HDR_NB_ST] = len(b_hf), len(b_poly), len(b_st)
        hdr[HDR_MAXP_HF] = max([b[1] for b in b_hf], default=0)
        hdr[HDR_MAXP_POLY] = max([b[1] for b in b_poly], default=0)
        hdr[HDR_FLAGS] = 1 if all_trivial else 0
        hdr[HDR_N_DPAR] = len(dp_local)
        hdr[HDR_LEN_BLK] = blk_len
        hdr[HDR_N_GRP], hdr[HDR_OFF_GRP] = len(groups), off_grp
        hdr[HDR_N_GEN], hdr[HDR_OFF_GEN] = len(gen_idx), off_gen
        hdr[HDR_N_MNT], hdr[HDR_OFF_MNT] = len(mnt_idx), off_mnt
        hdr[HDR_N_XGRP], hdr[HDR_OFF_XGRP] = len(xgroups), off_xgrp
        hdr[HDR_N_FOLD], hdr[HDR_OFF_FSLOT], hdr[HDR_OFF_FSRC] = len(fslots), off_fslot, off_fsrc
        hdr[HDR_OFF_WB] = off_wb
        if xprog:
            hdr[HDR_OFF_XPROG] = off_xprog
            hdr[HDR_X_NROW] = XR_U + xprog['n_u'] + xprog['n_prod']            # row columns in front of the q columns
            hdr[HDR_X_NSUM] = xprog['n_sum']
        block = (hdr + [v for t in terms for v in t] + [v for f in facs for v in f] + [v for b in bfuns for v in b] +
                 [v for g in groups for v in g] + gen_idx + mnt_idx + [v for g in xgroups for v in g] + [0] * xpad +
                 (xprog['ints'] if xprog else []) + [0] * pad)
        assert len(block) == blk_len
        fold_off.append(fold_off[-1] + fold_len)
        ftab.extend([v for sl in fslots for v in sl] + [v for e in fsrc for v in e])
        ftab_off.append(len(ftab))
        itab.extend(block)
        for src in dp_local:
            cm.dpar_sources.append(src)
            dpar.append(src[1] if src[0] == 'hf' else np.nan)
        comp_off.append(len(itab))
        dpar_off.append(len(dpar))
        coef_off.append(coef_off[-1] + len(nm_terms) + len(mon_terms))
        # per-sample weight slots: components with monotone cross terms, and integrated components whose B functions are
        # the dense order sets 1..P (ttm_eval.h "dense B set": their weights are copied there with the constants folded in)
        dense_b = (not separable and len(b_st) == 0 and len(b_hf) == hdr[HDR_MAXP_HF] and len(b_poly) == hdr[HDR_MAXP_POLY])
        # integrated components whose x_k-functions are polynomials / Hermite functions only (any set of orders, no special
        # terms): g(t) has the monomial form E(t) H(t) + A(t) that csrc/ttm_dense.h evaluates (flag bit 2); the weights of
        # their B functions go through the per-sample slots as well
        poly_b = (not separable and len(b_st) == 0 and len(b_hf) + len(b_poly) > 0)
        nslots.append(len(bfuns) + 1 if (len(mnt_idx) or len(xgroups) or dense_b or poly_b) else 0)
        # (bits 8-11 / 12-15: the largest Hermite-function / plain polynomial order among the x_k-functions - the dense
        # integrated kernels of csrc/ttm_int.hip are instantiated per order class)
        # (bit 3: the component has special terms somewhere - a kernel without any needs no erf table in LDS)
        has_st = any(f[0] == 'st' for t in list(nm_terms) + list(mon_terms) for f in t)
        # (bit 4: the component has an X program)
        complex_all.append(int(complex_comp) | (2 if dense_b else 0) | (4 if poly_b else 0) | (8 if has_st else 0) | (16 if xprog else 0) |
                           (((XR_U + xprog['n_u'] + xprog['n_prod']) << 16) if xprog else 0) |   # (bits 16-23: row columns in front of the q columns)
                           ((((xprog['n_sum'] + 63) // 64) << 24) if xprog else 0) |             # (bits 24-25: sums of an evaluation / 64, rounded up)
                           (min(int(hdr[HDR_MAXP_HF]), 15) << 8) | (min(int(hdr[HDR_MAXP_POLY]), 15) << 12))
        nb1.append(len(bfuns) + 1)
        n_nm_all.append(len(nm_terms))
        n_mon_all.append(len(mon_terms))
        # separable components whose monotone terms are all plain special terms of x_k: (kind, offset of the term's five
        # constants in the component's dpar slice) per coefficient - the optimiser then recomputes the derivative basis
        # from the x_k column instead of streaming a cached N x m matrix (ttm_objective_sep_direct_marked)
        mon_records = terms[len(nm_terms):]
        if separable and len(mon_records) and all(r[1] == 0 and r[2] >= 0 and bfuns[r[2]][0] in ST_KINDS.values() for r in mon_records):
            cm.sep_direct.append((kc, [(bfuns[r[2]][0], bfuns[r[2]][2]) for r in sorted(mon_records, key=lambda r: r[3])]))
        else:
            cm.sep_direct.append(None)
        cm.descriptors_mon.append([_descriptor(t) for t in mon_terms])
        cm.descriptors_nonmon.append([_descriptor(t) for t in nm_terms] if len(nm_terms) else None)
        cm.bounds.append([[-np.inf, np.inf] if (not isinstance(e, str) and len(e) == 0) else [0., np.inf]
                          for e in monotone[k]])

    cm.itab = np.asarray(itab, dtype=np.int32)
    cm.dpar = np.asarray(dpar if len(dpar) else [0.0], dtype=np.float64)
    cm.comp_off = np.asarray(comp_off, dtype=np.int32)
    cm.dpar_off = np.asarray(dpar_off, dtype=np.int32)
    cm.coef_off = np.asarray(coef_off, dtype=np.int32)
    cm.nslots = np.asarray(nslots, dtype=np.int32)
    cm.nb1 = np.asarray(nb1, dtype=np.int32)
    cm.fold_off = np.asarray(fold_off, dtype=np.int32)
    cm.ftab = np.asarray(ftab if len(ftab) else [0], dtype=np.int32)
    cm.ftab_off = np.asarray(ftab_off, dtype=np.int32)
    cm.plan_ways, cm.plan_loads = _plan_column_cache(plan_seq, fdesc, fints)
    cm.complex = np.asarray(complex_all, dtype=np.int32)
    cm.fdesc = np.asarray(fdesc, dtype=np.int32)
    cm.fints = np.asarray(fints if len(fints) else [0], dtype=np.int32)
    cm.offsets = np.concatenate((cm.comp_off, cm.dpar_off, cm.coef_off, cm.fold_off, cm.ftab_off)).astype(np.int32)
    cm.n_nm = np.asarray(n_nm_all, dtype=np.int32)
    cm.n_mon = np.asarray(n_mon_all, dtype=np.int32)
    _compile_uform(cm, u_info, polyclass, separable)
    return cm


# ---- univariate form (include/ttm.h "U-form", csrc/ttm_uform.h) -----------------------------------------
U_PMAX, U_TSTRIDE, U_NI_MAX, UC_LEN, UG_LEN = 10, 14, 128, 8, 8
U_GSTRIDE = 24               # doubles per group in the U section: B[0..11], A[0..11]
H_HDR, H_NG_MAX, H_GS = 8, 5, (0, 8, 16, 24, 24)
H_DB, H_DA = (0, 3, 5, 7, 10), (0, 1, 5, 7, 10)
P_HDR, P_LAG_MAX, P_FEW_D = 8, 5, 4
UCF_OWN, UGF_POLY = 1, 1 << 20
U_KAPPA = 0.75                # spline interval width / smallest special-term scale (degree 11: fit error < 1e-14;
                              # 0.5: 3e-15, 0.9: 5e-14 - wider intervals = smaller tables to stream per sweep step)
U_SUPPORT = 6.0 * np.sqrt(2.0)   # |x - centre| / scale beyond which erf / the Gaussian are at their limits to 1e-16
U_TOL_VALUE, U_TOL_DERIV = 2e-13, 2e-11     # accepted fit errors (relative to 1 + |exact|) before falling back


def _compile_uform(cm, u_info, polyclass, separable):
    """Static part of the U-form: group records with the planned-cache flags, monomial conversion matrix.
    The spline geometry depends on the special-term constants and is (re)computed by uform_geometry()."""
    cm.u_static = False
    cm.u_enabled = False
    cm.u_info = u_info
    cm.ucomp = np.zeros(max(1, cm.D) * UC_LEN, dtype=np.int32)
    cm.ugrp = np.zeros(UG_LEN, dtype=np.int32)
    cm.umono = np.zeros((U_PMAX + 1) ** 2)
    cm.ugeo = np.zeros(2 * max(1, cm.D))
    cm.u_size, cm.u_err_off = 0, 0
    cm.u_h_off, cm.u_h_cls, cm.u_h_ng = 0, 0, 0
    cm.u_p_off, cm.u_p_lag, cm.u_p_stride = 0, 0, 0
    if not separable or any(u['complex'] for u in u_info):
        return
    if any(g[1] > U_PMAX for u in u_info for g in u['groups']) or \
            any(max(u['maxP_hf'], u['maxP_poly']) > U_PMAX for u in u_info):
        return
    for n in range(U_PMAX + 1):
        c = polyclass.basis(n).convert(kind=np.polynomial.Polynomial).coef
        cm.umono[n * (U_PMAX + 1):n * (U_PMAX + 1) + len(c)] = c
    ucomp, ugrp = [], []
    for k, u in enumerate(u_info):
        grp_off = len(ugrp) // UG_LEN
        for gi, (var, P, off, p_hf, p_plain) in enumerate(u['groups']):
            fl = int(cm.fints[u['fint_off'] + 4 * gi + 3]) | (p_hf << 16) | (p_plain << 24) | (UGF_POLY if p_plain else 0)
            ugrp += [var, fl, off if p_plain else -1, p_plain, off + P if p_hf else -1, p_hf, 0, 0]
        flags = 0
        if u['maxP_hf'] or u['maxP_poly']:
            flags |= UCF_OWN
            fl = (PLAN_HF if u['maxP_hf'] else 0) | (UGF_POLY if u['maxP_poly'] else 0) | \
                (u['maxP_hf'] << 16) | (u['maxP_poly'] << 24)
            ugrp += [u['kc'], fl, u['stream_rel'] + u['maxP_hf'] if u['maxP_poly'] else -1, u['maxP_poly'],
                     u['stream_rel'] if u['maxP_hf'] else -1, u['maxP_hf'], 0, 0]
        ucomp += [u['kc'], int(cm.fdesc[k * FD_LEN + FD_KC_SLOT]), len(u['groups']), grp_off, 0, 0, 0, flags]
    ucomp = np.asarray(ucomp, dtype=np.int32).reshape(-1, UC_LEN)
    ugrp = np.asarray(ugrp if len(ugrp) else [0] * UG_LEN, dtype=np.int32).reshape(-1, UG_LEN)
    state = _plan_eager_e(ucomp, ugrp, cm.fints, [int(cm.fdesc[k * FD_LEN + 14]) for k in range(cm.D)])
    cm.ucomp = np.concatenate((ucomp.ravel(), state.ravel())).astype(np.int32)
    cm.ugrp = ugrp.ravel().copy()
    cm.u_static = True
    # hot records (include/ttm.h "H section"): all-hit sweeps of maps with few groups per component
    need = PLAN_HF | PLAN_XHIT | PLAN_EHIT
    all_hit = True
    mb = ma = 0
    for k in range(cm.D):
        for g in range(int(ucomp[k, 2])):
            fl = int(ugrp[int(ucomp[k, 3]) + g, 1])
            hit = (fl & PLAN_XHIT) and ((fl & need) == need or not (fl & PLAN_HF))
            all_hit = all_hit and bool(hit)
            if fl & PLAN_HF:
                mb = max(mb, (fl >> 16) & 15)
            if fl & UGF_POLY:
                ma = max(ma, (fl >> 24) & 15)
    ng = int(max(ucomp[:, 2], default=0))
    # banded map (include/ttm.h "push records"): consecutive columns, every group reads one of the P_LAG_MAX columns in
    # front of its component, a spline in every component -> csrc/ttm_band.hip.  Records for lag 2 (what the kernels of
    # large maps are instantiated for) unless a group reaches three columns back, which only the kernels of maps with a
    # few components (P_FEW_D) take.
    kc0 = int(ucomp[0, 0]) if cm.D else 0
    lags = [int(ucomp[k, 0]) - int(ugrp[int(ucomp[k, 3]) + g, 0]) for k in range(cm.D) for g in range(int(ucomp[k, 2]))]
    maxlag = max(lags, default=1)
    # the monotone part of a component: its special-term spline, and for maps of a few components possibly ONE linear term
    # of its own variable next to it (or instead of it: the [k] terms of examples 05 / 06 / 07) - the push record carries
    # that coefficient; other polynomial / Hermite-function terms of the own variable stay with the generic kernels
    own = [bool(int(f) & UCF_OWN) for f in ucomp[:, 7]]
    own_linear = [o and u['maxP_hf'] == 0 and u['maxP_poly'] == 1 for o, u in zip(own, u_info)]
    few = cm.D <= P_FEW_D
    banded = cm.D >= 1 and all(int(ucomp[k, 0]) == kc0 + k for k in range(cm.D)) and all(lag >= 1 for lag in lags) and \
        (maxlag <= 2 or (maxlag <= P_LAG_MAX and few)) and \
        all((len(u['st_p0']) > 0 and not o) or (few and ol) for u, o, ol in zip(u_info, own, own_linear)) and \
        (any(len(u['st_p0']) > 0 for u in u_info) or few)            # (no spline at all: the smoother's block map, linear monotone parts)
    # (a banded map of a few components whose groups do not all hit the planned column cache - a group three columns back,
    # conditioning columns in front of the first component - or with linear own terms still gets hot records: as the source of
    # its push records only, u_p_lag = 3 says so)
    few_only = banded and few and (maxlag >= 3 or not all_hit or any(own))
    # (the hot-record kernels are instantiated for 2 and 4 group records; five groups - the third component of the smoother's block
    # map, example_07.py:368-408 - exist for maps of a few components, whose records only feed the push records)
    if ((all_hit and not any(own)) or few_only) and ng <= (H_NG_MAX if few_only else 4):
        # (orders 8..10 - class 4, example_03.py:103 - exist for the kernels of maps with a few components only)
        cm.u_h_cls = 1 if (mb <= 3 and ma <= 1) else (2 if (mb <= 5 and ma <= 5) else (3 if (mb <= 7 and ma <= 7) else (4 if (banded and few) else 0)))
    if cm.u_h_cls:
        cm.u_h_ng = 2 if ng <= 2 else (4 if ng <= 4 else H_NG_MAX)        # the kernels are instantiated for 2 and 4 group records
        if banded:
            cm.u_p_lag = (3 if maxlag <= 3 else P_LAG_MAX) if few_only else 2
            gp = H_DB[cm.u_h_cls] + 1 + H_DA[cm.u_h_cls]
            cm.u_p_stride = -(-(P_HDR + cm.u_p_lag * gp) // 8) * 8        # whole 64-byte lines
    uform_geometry(cm)


UCF_PUT_E = 2


def _plan_eager_e(ucomp, ugrp, fints, plan_offs):
    """The planned column cache of the direct kernels computes exp(-x^2/4) of a cached column lazily, at its
    first Hermite-function reader (that reader's flag word lacks TTM_PLAN_EHIT).  The U-form kernels compute
    it when the column is PUT (the component's own x_k, already in registers) if any reader needs it before the
    slot is reused: every such reader then carries EHIT and takes the two-LDS-read fast path.  Returns the
    entry-state words (TTM_PLAN_WAYS per component, column | TTM_PLAN_E) under these semantics, for sweeps
    that start at k0 > 0."""
    D = ucomp.shape[0]
    # events in sweep order: ('r', group index, slot, hf, hit) reads and ('p', k, slot) puts
    events = []
    for k in range(D):
        for g in range(int(ucomp[k, 2])):
            gi = int(ucomp[k, 3]) + g
            fl = int(ugrp[gi, 1])
            events.append(('r', gi, (fl >> 8) & 255, bool(fl & PLAN_HF), bool(fl & PLAN_XHIT), k))
        events.append(('p', k, int(ucomp[k, 1]), False, False, k))
    for i, ev in enumerate(events):
        if ev[0] != 'p' or ev[2] < 0:
            continue
        readers = []
        for ev2 in events[i + 1:]:
            if ev2[2] != ev[2]:
                continue
            if ev2[0] == 'p' or not ev2[4]:
                break                                   # slot rewritten (put, or filled by a miss)
            if ev2[3]:
                readers.append(ev2[1])
        if readers:
            ucomp[ev[1], 7] |= UCF_PUT_E
            for gi in readers:
                ugrp[gi, 1] |= PLAN_EHIT
    # entry states: replay
    state = np.full((D, PLAN_WAYS), -1, dtype=np.int64)
    slots = {}
    ei = 0
    for k in range(D):
        if k == 0:
            base = fints[plan_offs[0]:plan_offs[0] + PLAN_WAYS]
            for w in range(PLAN_WAYS):
                if base[w] >= 0:
                    slots[w] = [int(base[w]) & ~PLAN_E, bool(int(base[w]) & PLAN_E)]
        for w, (col, e) in slots.items():
            state[k, w] = col | (PLAN_E if e else 0)
        while ei < len(events) and events[ei][5] == k:
            ev = events[ei]
            ei += 1
            if ev[2] in (255, -1):
                continue
            if ev[0] == 'p':
                slots[ev[2]] = [int(ucomp[k, 0]), bool(ucomp[k, 7] & UCF_PUT_E)]
            else:
                gi = ev[1]
                if not ev[4]:
                    slots[ev[2]] = [int(ugrp[gi, 0]), ev[3]]
                elif ev[3]:
                    slots[ev[2]][1] = True
    return state.astype(np.int32)


def uform_geometry(cm, kappa=None):
    """Spline geometry of every component from the current special-term constants (cm.dpar): support
    [t_lo, t_hi] = union of centre -+ U_SUPPORT scale, n intervals of width h <= kappa min(scale); fills the
    NI / TAB_OFF / DBL_OFF words of ucomp, ugeo, the U-section size, and decides u_enabled."""
    if not cm.u_static:
        cm.u_enabled = False
        return False
    if kappa is None:
        import os
        kappa = float(os.environ.get('TTM_U_KAPPA', U_KAPPA))      # (tuning knob)
    uc = cm.ucomp[:cm.D * UC_LEN].reshape(-1, UC_LEN)       # (view: the entry-state words follow)
    off, ok = 0, True
    geo = np.zeros((cm.D, 2))
    for k, u in enumerate(cm.u_info):
        ng = int(uc[k, 2]) + (1 if uc[k, 7] & UCF_OWN else 0)
        uc[k, 6] = off
        off += 4 + U_GSTRIDE * ng
    dpar = cm.dpar.tolist()         # (plain floats: a handful of terms per component - the array machinery of NumPy costs 45 us of
    isfinite = math.isfinite        #  every filter update here; the arithmetic is the same IEEE operations)
    for k, u in enumerate(cm.u_info):
        nI = 0
        if len(u['st_p0']):
            base = int(cm.dpar_off[k])
            mu = [dpar[base + p0] for p0 in u['st_p0']]
            sc = [dpar[base + p0 + 1] for p0 in u['st_p0']]
            if not (all(isfinite(v) for v in mu) and all(isfinite(v) and v > 0 for v in sc)):
                ok = False          # constants not placed yet (or degenerate): no U-form until they are
                mu, sc = [0.0], [1.0]
            t_lo = min(m_ - U_SUPPORT * s_ for m_, s_ in zip(mu, sc))
            t_hi = max(m_ + U_SUPPORT * s_ for m_, s_ in zip(mu, sc))
            n_int = int(math.ceil((t_hi - t_lo) / (kappa * min(sc))))
            n_int = max(n_int, 2)
            n_int += n_int % 2      # even number of columns: 16-byte copies
            nI = n_int + 2
            if nI > U_NI_MAX:
                ok = False
                nI = 2 + 2
                n_int = 2
            geo[k] = (t_lo, (t_hi - t_lo) / n_int)
        uc[k, 4] = nI
        uc[k, 5] = off
        off += U_TSTRIDE * nI
    cm.u_err_off = off
    off += 2 * cm.D
    off += (-off) % 8
    cm.u_h_off = off
    if cm.u_h_cls:
        off += cm.D * (H_HDR + cm.u_h_ng * H_GS[cm.u_h_cls])
    off += (-off) % 8
    cm.u_p_off = off
    if cm.u_p_lag:
        off += (cm.D + cm.u_p_lag) * cm.u_p_stride
    cm.u_size = off + (off % 2)
    cm.ugeo = geo.ravel().copy()
    cm.u_enabled = bool(ok)
    return cm.u_enabled
