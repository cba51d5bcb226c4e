// ttm_eval.h - per-sample evaluators of the triangular transport map.
//
// Everything a single sample needs (basis factors, the x_k-univariate "B"
// functions, Gauss-Legendre integration of the rectified monotone part, the
// objective/gradient contributions, the bisection and table root searches) is
// written once here as inline functions over three abstractions:
//   * a program view  (term tables of one component, staged in LDS on the GPU),
//   * a sample accessor x(var) (one coalesced column-major load per variable),
//   * per-sample scratch "slots" (an LDS column per thread on the GPU).
// The HIP kernels in ttm_kernels.hip wrap these bodies with LDS staging, grid
// loops and deterministic reductions.  The same bodies compile for the host so
// the interpreter logic can be unit-tested without a GPU (tests/hostemu) - that
// build is test infrastructure, never a fallback of the product.
//
// Evaluation strategy:
//   nonmonotone part  = folded constant
//                     + per variable: one polynomial recurrence with summed ("folded")
//                       coefficients for the plain and the Hermite-function terms, one
//                       exp(-x^2/4) per variable (cached across components in VarCache)
//                     + generic products for cross terms / special terms
//   monotone part     = g(t) = w_none + sum_b w_b B_b(t) over the distinct functions of x_k;
//                       w_b = folded coefficients (+ per-sample products for cross terms)
//
// Reference formulas restated (TM = /root/reference/transport_map.py):
//   factors / special terms  TM:905-1026, 1096-1150, 1166-1248
//   s() and the quadrature    TM:2499-2558, 4238-4258
//   rectifiers                TM:4981-5018, 5112-5165, 5167-5213
//   objective / jacobian      TM:3343-3376, 3475-3569 ; separable TM:2984-3006
//   bisection root search     TM:3842-3976 ; table root search TM:4039-4082
#pragma once

#include <math.h>
#include <stdint.h>

#include "../../include/ttm.h"
#include "ttm_math.h"

// Term tables, constants, coefficients and the quadrature rule are identical for every lane.  On the
// GPU they are read through constant-address-space pointers: the compiler then emits scalar loads
// (s_load_dword*) into SGPRs, the interpreter's control flow is scalar (s_cbranch) and uniform fp64
// operands feed v_fma_f64 directly as SGPR pairs - no LDS traffic, no v_readfirstlane, no VGPRs.
#if defined(__HIP_DEVICE_COMPILE__)
#define TTM_C __attribute__((address_space(4)))
#else
#define TTM_C
#endif
#define TTM_UNI(x) (x)

namespace ttm {

typedef const TTM_C int* cint_p;
typedef const TTM_C double* cdbl_p;

static constexpr double kLn2 = 0.6931471805599453;          // np.log(2)

// ---------------------------------------------------------------------------
// program views
// ---------------------------------------------------------------------------

struct Prog {            // what is common to all components
    cdbl_p qx;           // quadrature nodes / weights
    cdbl_p qw;
    const double* erf_tab;   // per-lane gathers: staged in LDS on the GPU
    int Q;
    int family;
    int mono;
    int rect;
    double delta;
};

struct Comp {            // one component block (pointers into staged tables)
    cint_p nm_terms;
    cint_p mon_terms;
    cint_p facs;
    cint_p bfuns;
    cint_p grp;
    cint_p gen;
    cint_p mnt;
    cint_p xgrp;         // cross groups {var, P, fold offset, has_hf, b, 0, 0, 0}
    cint_p fslot;        // fold recipe of the component (kernels that need the members of a folded sum set these):
    cint_p fsrc;         // slot s sums fsrc[2 (fslot[2s] + j)] = coefficient index (< 0: a constant), [.. + 1] = dpar index or -1
    cdbl_p dpar;
    cdbl_p cnm;          // coefficients of the nonmonotone / monotone terms
    cdbl_p cmon;
    cdbl_p fold;         // folded coefficients (see include/ttm.h)
    int kc, n_nm, n_mon, nB, nB_hf, nB_poly, nB_st, maxP_hf, maxP_poly, flags, n_grp, n_gen, n_mnt, n_xgrp, off_wb;
};

TTM_HD Comp make_comp(cint_p cb, cdbl_p dpar, cdbl_p coef, cdbl_p fold) {
    Comp c;
    c.kc = TTM_UNI(cb[TTM_HDR_KC]);
    c.n_nm = TTM_UNI(cb[TTM_HDR_N_NM]);
    c.n_mon = TTM_UNI(cb[TTM_HDR_N_MON]);
    c.nB = TTM_UNI(cb[TTM_HDR_NB]);
    c.nB_hf = TTM_UNI(cb[TTM_HDR_NB_HF]);
    c.nB_poly = TTM_UNI(cb[TTM_HDR_NB_POLY]);
    c.nB_st = TTM_UNI(cb[TTM_HDR_NB_ST]);
    c.maxP_hf = TTM_UNI(cb[TTM_HDR_MAXP_HF]);
    c.maxP_poly = TTM_UNI(cb[TTM_HDR_MAXP_POLY]);
    c.flags = TTM_UNI(cb[TTM_HDR_FLAGS]);
    c.n_grp = TTM_UNI(cb[TTM_HDR_N_GRP]);
    c.n_gen = TTM_UNI(cb[TTM_HDR_N_GEN]);
    c.n_mnt = TTM_UNI(cb[TTM_HDR_N_MNT]);
    c.n_xgrp = TTM_UNI(cb[TTM_HDR_N_XGRP]);
    c.xgrp = cb + TTM_UNI(cb[TTM_HDR_OFF_XGRP]);
    c.fslot = nullptr;
    c.fsrc = nullptr;
    c.off_wb = TTM_UNI(cb[TTM_HDR_OFF_WB]);
    c.nm_terms = cb + TTM_UNI(cb[TTM_HDR_OFF_NM]);
    c.mon_terms = cb + TTM_UNI(cb[TTM_HDR_OFF_MON]);
    c.facs = cb + TTM_UNI(cb[TTM_HDR_OFF_FAC]);
    c.bfuns = cb + TTM_UNI(cb[TTM_HDR_OFF_B]);
    c.grp = cb + TTM_UNI(cb[TTM_HDR_OFF_GRP]);
    c.gen = cb + TTM_UNI(cb[TTM_HDR_OFF_GEN]);
    c.mnt = cb + TTM_UNI(cb[TTM_HDR_OFF_MNT]);
    c.dpar = dpar;
    c.cnm = coef;
    c.cmon = coef + c.n_nm;
    c.fold = fold;
    return c;
}

// folded coefficients of one component: slots [first, last) with the given stride
// (one thread per slot on the GPU; each slot sums its sources in a fixed order);
// cb = the component's itab block, fb = its ftab block
TTM_HD void fold_coeffs(const int* cb, const int* fb, const double* dpar, const double* coef, double* fold, int first, int stride) {
    // (plain pointers: the slot index differs per thread here)
    const int n = cb[TTM_HDR_N_FOLD];
    const int* slots = fb + cb[TTM_HDR_OFF_FSLOT];
    const int* src = fb + cb[TTM_HDR_OFF_FSRC];
    for (int s = first; s < n; s += stride) {
        const int s0 = slots[2 * s], ns = slots[2 * s + 1];
        double acc = 0.0;
        for (int j = 0; j < ns; ++j) {
            const int ci = src[2 * (s0 + j)], p0 = src[2 * (s0 + j) + 1];
            acc += (ci < 0) ? dpar[p0] : ((p0 >= 0) ? coef[ci] * dpar[p0] : coef[ci]);
        }
        fold[s] = acc;
    }
}

// second stage of the fold (after every slot of fold_coeffs is written): the unified special-term section
// of the fast path, TTM_FD_ST8.  fd = the component's fast-path descriptor, fold = its folded array.
TTM_HD void fold_st8(const int* fd, const int* fints, double* fold, int first, int stride) {
    const int n_st = fd[TTM_FD_N_ST];
    const int* kinds = fints + fd[TTM_FD_FINT_OFF] + 4 * fd[TTM_FD_N_GRP];
    const int* order = kinds + n_st;
    const double* rec5 = fold + fd[TTM_FD_STREAM] + fd[TTM_FD_MAXP_HF] + fd[TTM_FD_MAXP_POLY];
    double* out = fold + fd[TTM_FD_ST8];
    for (int s = first; s < n_st; s += stride) {
        const int src = order[s], kind = kinds[src];
        const double w = rec5[5 * src], mu = rec5[5 * src + 1], inv = rec5[5 * src + 2], k1 = rec5[5 * src + 3],
                     k2 = rec5[5 * src + 4];
        double A1 = 0.0, B0 = 0.0, B1 = 0.0, G = 0.0, DG = 0.0, DT = 0.0;
        if (kind == TTM_KIND_LET) { B0 = 0.5 * w; B1 = -0.5 * w; G = -(w * (0.5 * k1)); }
        else if (kind == TTM_KIND_RET) { B0 = 0.5 * w; B1 = 0.5 * w; G = w * (0.5 * k1); }
        else if (kind == TTM_KIND_RBF) { G = w * k2; DT = (-2.0 * inv) * (w * k2); }
        else { A1 = 0.5 * w; DG = w * k2; }
        double* r = out + 8 + 8 * s;
        r[0] = mu; r[1] = inv; r[2] = A1; r[3] = B0; r[4] = B1; r[5] = G; r[6] = DG; r[7] = DT;
    }
    if (first == 0) {
        double g0 = fold[fd[TTM_FD_OFF_WB] + fd[TTM_FD_NB]];
        for (int s = 0; s < n_st; ++s)
            if (kinds[s] == TTM_KIND_IRBF) g0 += 0.5 * rec5[5 * s];
        out[0] = g0;
        for (int j = 1; j < 8; ++j) out[j] = 0.0;
    }
}

// ---------------------------------------------------------------------------
// polynomial families: three-term recurrences with derivative propagation
//   P_{n+1} = (a x + b) P_n - c P_{n-1}
// (R = double or VecD<N>: per-sample values; everything else is uniform)
// ---------------------------------------------------------------------------

template <class R>
TTM_HD void poly_first(int fam, const R& x, R& p, R& dp) {
    switch (fam) {
        case TTM_FAM_HERMITE: p = 2.0 * x; dp = R(2.0); break;
        case TTM_FAM_LAGUERRE: p = 1.0 - x; dp = R(-1.0); break;
        default: p = x; dp = R(1.0); break;
    }
}

template <bool DER, class R>
TTM_HD void poly_next(int fam, int n, const R& x, R& pm, R& p, R& dpm, R& dp) {
    double a, b, c;
    switch (fam) {
        case TTM_FAM_HERMITE_E: a = 1.0; b = 0.0; c = (double)n; break;
        case TTM_FAM_POWER: a = 1.0; b = 0.0; c = 0.0; break;
        case TTM_FAM_HERMITE: a = 2.0; b = 0.0; c = 2.0 * n; break;
        case TTM_FAM_CHEBYSHEV: a = 2.0; b = 0.0; c = 1.0; break;
        case TTM_FAM_LAGUERRE: { double r = 1.0 / (n + 1.0); a = -r; b = (2.0 * n + 1.0) * r; c = n * r; } break;
        default: /* LEGENDRE */ { double r = 1.0 / (n + 1.0); a = (2.0 * n + 1.0) * r; b = 0.0; c = n * r; } break;
    }
    const R lin = vfma(a, x, b);
    const R pn = vfma(lin, p, -c * pm);
    if (DER) {
        const R dpn = vfma(a, p, vfma(lin, dp, -c * dpm));
        dpm = dp; dp = dpn;
    }
    pm = p; p = pn;
}

// P_order(x), order >= 1
template <class R>
TTM_HD R poly_eval(int fam, int order, const R& x) {
    R pm(1.0), dpm(0.0), p, dp;
    poly_first(fam, x, p, dp);
    for (int n = 1; n < order; ++n) poly_next<false>(fam, n, x, pm, p, dpm, dp);
    return p;
}

// ---------------------------------------------------------------------------
// special terms (TM:917-1016), value and d/dx.
// par = {centre, scale, 1/(sqrt2 scale), scale sqrt(2/pi), 1/(sqrt(2 pi) scale)}
// ---------------------------------------------------------------------------

template <bool VAL, bool DER, class R>
TTM_HD void st_eval(const Prog& p, int kind, const R& x, cdbl_p par, R& val, R& der) {
    const R d = x - par[0];
    const R t = d * par[2];
    val = R(0.0); der = R(0.0);
    R e, g;
    if (kind == TTM_KIND_LET || kind == TTM_KIND_RET) {
        if (VAL) erf_gauss_tab<true>(p.erf_tab, t, e, g); else erf_gauss_tab<false>(p.erf_tab, t, e, g);
        const R h = (kind == TTM_KIND_LET) ? vfma(e, -0.5, 0.5) : vfma(e, 0.5, 0.5);   // (1 -/+ erf)/2
        if (VAL) {
            const R gg = (0.5 * par[3]) * g;
            val = (kind == TTM_KIND_LET) ? vfma(d, h, -gg) : vfma(d, h, gg);
        }
        if (DER) der = h;
    } else if (kind == TTM_KIND_RBF) {
        erf_gauss_tab<true>(p.erf_tab, t, e, g);
        const R gg = g * par[4];
        if (VAL) val = gg;
        if (DER) der = (-2.0 * par[2]) * t * gg;                                    // -(x-mu)/scale^2 * value
    } else {  // TTM_KIND_IRBF
        if (DER) erf_gauss_tab<true>(p.erf_tab, t, e, g); else erf_gauss_tab<false>(p.erf_tab, t, e, g);
        if (VAL) val = vfma(e, 0.5, 0.5);
        if (DER) der = par[4] * g;
    }
}

// ---------------------------------------------------------------------------
// rectifiers
// ---------------------------------------------------------------------------

template <class R>
TTM_HD R rect_eval(int mode, const R& g) {
    switch (mode) {
        case TTM_RECT_EXPONENTIAL: return fast_exp(g);
        case TTM_RECT_SOFTPLUS: { const R ag = kLn2 * g; return fast_log(1.0 + fast_exp(-vabs(ag))) + vselect_lt0(ag, R(0.0), ag); }
        case TTM_RECT_SQUARED: return g * g;
        case TTM_RECT_EXPNEG: return fast_exp(-g);
        default: return vselect_lt0(g, fast_exp(g), g + 1.0);   // ELU, TM:5012-5016
    }
}

// r(g), the factor of evaluate_dfdc (TM:5112-5165) and log(r + delta) as logevaluate (TM:5167-5213)
template <class R>
TTM_HD void rect_all(int mode, double delta, const R& g, R& r, R& dr, R& logr) {
    switch (mode) {
        case TTM_RECT_EXPONENTIAL:
            r = fast_exp(g); dr = r; logr = (delta == 0.0) ? g : fast_log(r + delta); break;
        case TTM_RECT_SOFTPLUS:
            // (the exponent capped: 1 / (1 + inf) is 0 in the reference's arithmetic, TM:5140-5147, but NaN through the Newton steps
            // of fast_rcp; 1 / (1 + e^700) = 1e-304 stands for it)
            r = rect_eval(mode, g); dr = fast_rcp(1.0 + fast_exp(vmin(-kLn2 * g, 700.0))); logr = fast_log(r + delta); break;
        case TTM_RECT_EXPNEG:
            r = fast_exp(-g); dr = -r; logr = -g; break;
        case TTM_RECT_SQUARED:
            r = g * g; dr = R(NAN); logr = fast_log(r); break;     // dfdc "not implemented" in the reference
        default:
            r = rect_eval(mode, g); dr = R(NAN); logr = fast_log(r); break;
    }
}

// ---------------------------------------------------------------------------
// per-sample cache of recently used columns: x_j and exp(-x_j^2/4).
// Tags are uniform (SGPRs); banded maps reuse each column in 2-3 consecutive
// components, the inverse kernels read back columns they have just written.
// ---------------------------------------------------------------------------

// PAIRED = false: value i of sample e at base[(i L + e) stride] (base = cache + tid): consecutive lanes read
// consecutive doubles.  PAIRED = true (hot kernels): the two values of a slot (x_j, exp(-x_j^2/4)) are adjacent,
// slot s of sample e at base[(s L + e) 2 stride] (base = cache + 2 tid): get2 / set2 move both with ONE 16-byte
// LDS access (ds_read_b128: 256 B/clk, against 128 B/clk for the ds_read2st64_b64 the compiler makes of two gets).
template <class R, bool PAIRED = false>
struct CacheStore {          // 8 values per sample: x of four columns and their exp(-x^2/4)
    double* base;            // this thread's column
    int stride;              // doubles between consecutive slots of one element
    const double* etab = nullptr;    // 2^(j/32) table for exp_q_tab (hot kernels only)
    static constexpr int L = lanes_of<R>::value;
    TTM_HD double* at(int i, int e) const {
        return PAIRED ? base + ((i >> 1) * L + e) * (2 * stride) + (i & 1) : base + (i * L + e) * stride;
    }
    TTM_HD R get(int i) const {
        R r;
#pragma unroll
        for (int e = 0; e < L; ++e) set_elem(r, e, *at(i, e));
        return r;
    }
    TTM_HD void set(int i, const R& v) const {
#pragma unroll
        for (int e = 0; e < L; ++e) *at(i, e) = elem(v, e);
    }
    // both values of the slot whose first value has the (even) index i2
    TTM_HD void get2(int i2, R& x, R& ev) const {
        if (PAIRED) {
#pragma unroll
            for (int e = 0; e < L; ++e) {
                double a, b;
                load_pair(at(i2, e), a, b);
                set_elem(x, e, a); set_elem(ev, e, b);
            }
        } else {
            x = get(i2); ev = get(i2 + 1);
        }
    }
    TTM_HD void set2(int i2, const R& x, const R& ev) const {
        if (PAIRED) {
#pragma unroll
            for (int e = 0; e < L; ++e) store_pair(at(i2, e), elem(x, e), elem(ev, e));
        } else {
            set(i2, x); set(i2 + 1, ev);
        }
    }
};

template <class XA, class R>
struct VarCache {
    // four ways; tags / valid flags are uniform (SGPRs), the values live in a per-thread LDS column
    // (keeping them in VGPRs costs a register shuffle at every uniform branch merge)
    const XA& xa;
    CacheStore<R> st;
    int t0, t1, t2, t3;          // tags
    int h0, h1, h2, h3;          // exp(-x^2/4) valid
    int rr;
    TTM_HD VarCache(const XA& x, const CacheStore<R>& store)
        : xa(x), st(store), t0(-1), t1(-1), t2(-1), t3(-1), h0(0), h1(0), h2(0), h3(0), rr(0) {}
    TTM_HD int find(int var) const { return t0 == var ? 0 : (t1 == var ? 1 : (t2 == var ? 2 : (t3 == var ? 3 : -1))); }
    TTM_HD void set_h(int s, int v) { if (s == 0) h0 = v; else if (s == 1) h1 = v; else if (s == 2) h2 = v; else h3 = v; }
    TTM_HD int get_h(int s) const { return s == 0 ? h0 : (s == 1 ? h1 : (s == 2 ? h2 : h3)); }
    TTM_HD int put(int var, const R& x) {
        int s = find(var);
        if (s < 0) {
            s = rr;
            rr = (rr + 1) & 3;
            if (s == 0) t0 = var; else if (s == 1) t1 = var; else if (s == 2) t2 = var; else t3 = var;
        }
        set_h(s, 0);
        st.set(s, x);
        return s;
    }
    TTM_HD R get(int var) {
        const int s = find(var);
        if (s >= 0) return st.get(s);
        const R x = xa(var);
        put(var, x);
        return x;
    }
    // x and exp(-x^2/4)
    TTM_HD void get_e(int var, R& x, R& e) {
        int s = find(var);
        if (s >= 0) {
            x = st.get(s);
        } else {
            x = xa(var);
            s = put(var, x);
        }
        if (get_h(s)) {
            e = st.get(4 + s);
        } else {
            e = exp_q_fast(x);
            st.set(4 + s, e);
            set_h(s, 1);
        }
    }
    TTM_HD R operator()(int var) { return get(var); }
};

// ---------------------------------------------------------------------------
// A-part of a term: product of its factors on columns other than kc
// (all 'HF' factors of one term share a single exp(-sum x^2 / 4))
// ---------------------------------------------------------------------------

// exp(-x_var^2 / 4) of ONE column: from the column cache when the accessor has one
template <class XA, class R>
TTM_HD R hf_exp_of(XA&, int, const R& xv) { return fast_exp(-0.25 * (xv * xv)); }
template <class XA2, class R>
TTM_HD R hf_exp_of(VarCache<XA2, R>& x, int var, const R&) {
    R xv, e;
    x.get_e(var, xv, e);
    return e;
}

template <class R, class XA>
TTM_HD R eval_A(cint_p term, const Comp& c, const Prog& p, XA& x) {
    const int f0 = TTM_UNI(term[0]);
    const int nf = TTM_UNI(term[1]);
    R prod(1.0), ssq(0.0);
    bool hf = false;
    if (nf == 1) {                                            // the usual cross / nonmonotone term: one factor
        cint_p F = c.facs + 4 * f0;
        const int var = TTM_UNI(F[0]), kind = TTM_UNI(F[1]);
        if (kind == TTM_KIND_HF) {
            const R xv = x(var);
            return (c.dpar[TTM_UNI(F[3])] * poly_eval(p.family, TTM_UNI(F[2]), xv)) * hf_exp_of(x, var, xv);
        }
    }
    for (int f = 0; f < nf; ++f) {
        cint_p F = c.facs + 4 * (f0 + f);
        const int var = TTM_UNI(F[0]);
        const int kind = TTM_UNI(F[1]);
        const int order = TTM_UNI(F[2]);
        const int p0 = TTM_UNI(F[3]);
        const R xv = x(var);
        if (kind == TTM_KIND_POLY || kind == TTM_KIND_HF) {
            const R P = poly_eval(p.family, order, xv);
            if (kind == TTM_KIND_HF) {
                prod = prod * (c.dpar[p0] * P);
                ssq = vfma(xv, xv, ssq);
                hf = true;
            } else {
                prod = prod * P;
            }
        } else {
            R v, dv;
            st_eval<true, false>(p, kind, xv, c.dpar + p0, v, dv);
            prod = prod * v;
        }
    }
    if (hf) prod = prod * fast_exp(-0.25 * ssq);
    return prod;
}

// sum_i c_i Psi_nonmon,i : folded constant + per-variable recurrences + generic terms
template <class R, class XA>
TTM_HD R nonmon_sum(const Comp& c, const Prog& p, VarCache<XA, R>& x) {
    R s(c.fold[0]);
    for (int g = 0; g < c.n_grp; ++g) {
        cint_p G = c.grp + 4 * g;
        const int var = TTM_UNI(G[0]);
        const int P = TTM_UNI(G[1]);
        cdbl_p al = c.fold + TTM_UNI(G[2]);
        cdbl_p be = al + P;
        const int has_hf = TTM_UNI(G[3]);
        R xv, e(0.0);
        if (has_hf) x.get_e(var, xv, e); else xv = x.get(var);
        R pm(1.0), dpm(0.0), pn, dp, accp(0.0), acch(0.0);
        poly_first(p.family, xv, pn, dp);
        for (int n = 1; n <= P; ++n) {
            accp = vfma(al[n - 1], pn, accp);
            if (has_hf) acch = vfma(be[n - 1], pn, acch);
            if (n < P) poly_next<false>(p.family, n, xv, pm, pn, dpm, dp);
        }
        s = s + accp;
        if (has_hf) s = vfma(e, acch, s);
    }
    for (int i = 0; i < c.n_gen; ++i) {
        cint_p T = c.nm_terms + 4 * TTM_UNI(c.gen[i]);
        s = vfma(c.cnm[TTM_UNI(T[3])], eval_A<R>(T, c, p, x), s);
    }
    return s;
}

// weights of the B functions for this sample: folded part + cross-term products.
// Only needed when the component has monotone cross terms (n_mnt > 0).
template <class R, class XA, class Slots>
TTM_HD void mon_weights(const Comp& c, const Prog& p, XA& x, Slots& w) {
    cdbl_p wb = c.fold + c.off_wb;
    for (int b = 0; b <= c.nB; ++b) w.set(b, R(wb[b]));
    // cross terms with one polynomial / Hermite-function factor: per (B function, variable) a folded series - one
    // recurrence and the variable's cached exp(-x^2/4) instead of a walk through the term and factor records per term
#ifndef INT_X_NOXGRP                                        /* (INT_X_*: timing experiments, results wrong by construction) */
    for (int g = 0; g < c.n_xgrp; ++g) {
        cint_p G = c.xgrp + 8 * g;
        const int var = TTM_UNI(G[0]);
        const int P = TTM_UNI(G[1]);
        cdbl_p al = c.fold + TTM_UNI(G[2]);
        cdbl_p be = al + P;
        const int has_hf = TTM_UNI(G[3]);
        const int b = TTM_UNI(G[4]);
        R xv, e(0.0);
        if (has_hf) x.get_e(var, xv, e); else xv = x.get(var);
        R pm(1.0), dpm(0.0), pn, dp, accp(0.0), acch(0.0);
        poly_first(p.family, xv, pn, dp);
        for (int n = 1; n <= P; ++n) {
            accp = vfma(al[n - 1], pn, accp);
            if (has_hf) acch = vfma(be[n - 1], pn, acch);
            if (n < P) poly_next<false>(p.family, n, xv, pm, pn, dpm, dp);
        }
        R wv = w.get(b) + accp;
        if (has_hf) wv = vfma(e, acch, wv);
        w.set(b, wv);
    }
#endif
#ifndef INT_X_NOMNT
    for (int j = 0; j < c.n_mnt; ++j) {
        cint_p T = c.mon_terms + 4 * TTM_UNI(c.mnt[j]);
        int b = TTM_UNI(T[2]);
        if (b < 0) b = c.nB;
        w.set(b, vfma(c.cmon[TTM_UNI(T[3])], eval_A<R>(T, c, p, x), w.get(b)));
    }
#endif
}

struct UniformW {            // uniform weights straight from the folded array (SGPR operands)
    cdbl_p wb;
    TTM_HD double get(int i) const { return wb[i]; }
};

// visit every distinct x_k-univariate function: f(b, B_b(t), B_b'(t))
template <bool DER, class R, class F>
TTM_HD void for_each_B(const Comp& c, const Prog& p, const R& t, F&& f) {
    int b = 0;
    if (c.nB_hf > 0) {
        const R E = fast_exp(-0.25 * (t * t));
        R pm(1.0), dpm(0.0), pn, dp;
        poly_first(p.family, t, pn, dp);
        for (int n = 1; n <= c.maxP_hf; ++n) {
            cint_p B = c.bfuns + 4 * b;
            if (TTM_UNI(B[1]) == n) {
                const double a = c.dpar[TTM_UNI(B[2])];
                // d/dt [a P e^{-t^2/4}] = -1/2 e^{-t^2/4} (t aP - 2 aP')   (TM:1245)
                f(b, a * pn * E, DER ? -0.5 * E * (t * (a * pn) - 2.0 * (a * dp)) : R(0.0));
                ++b;
            }
            if (n < c.maxP_hf) poly_next<DER>(p.family, n, t, pm, pn, dpm, dp);
        }
    }
    if (c.nB_poly > 0) {
        R pm(1.0), dpm(0.0), pn, dp;
        poly_first(p.family, t, pn, dp);
        for (int n = 1; n <= c.maxP_poly; ++n) {
            cint_p B = c.bfuns + 4 * b;
            if (TTM_UNI(B[1]) == n) {
                f(b, pn, dp);
                ++b;
            }
            if (n < c.maxP_poly) poly_next<DER>(p.family, n, t, pm, pn, dpm, dp);
        }
    }
    for (int s = 0; s < c.nB_st; ++s, ++b) {
        cint_p B = c.bfuns + 4 * b;
        R v, dv;
        st_eval<true, DER>(p, TTM_UNI(B[0]), t, c.dpar + TTM_UNI(B[2]), v, dv);
        f(b, v, dv);
    }
}

// g(t) = w[nB] + sum_b w[b] B_b(t)   (the argument of the rectifier, or the
// monotone part itself for separable maps) and dg/dt
template <bool DER, class R, class W>
TTM_HD void g_eval(const Comp& c, const Prog& p, const R& t, const W& w, R& g, R& dg) {
    R acc(w.get(c.nB)), dacc(0.0);
    for_each_B<DER>(c, p, t, [&](int b, const R& v, const R& dv) {
        const auto wb = w.get(b);
        acc = vfma(wb, v, acc);
        if (DER) dacc = vfma(wb, dv, dacc);
    });
    g = acc; dg = dacc;
}

// int_0^{xk} (r(g(t)) + delta) dt with the reference's node order and grouping (TM:4238-4258)
template <class R, class W>
TTM_HD R integrate_rect(const Comp& c, const Prog& p, const R& xk, const W& w) {
    const R half = xk * 0.5;
    R res(0.0);
    for (int q = 0; q < p.Q; ++q) {
        const R t = half * p.qx[q] + half;
        R g, dg;
        g_eval<false>(c, p, t, w, g, dg);
        const R fr = rect_eval(p.rect, g) + p.delta;
        const R term = half * (p.qw[q] * fr);
        res = (q == 0) ? term : res + term;
    }
    return res;
}

// ---------------------------------------------------------------------------
// Dense B set: the x_k-univariate functions of the component are the Hermite-function orders 1..Ph and the plain
// polynomial orders 1..Pp, no special terms - what an integrated-rectifier component of a polynomial map looks like.
// g(t) = w0 + E(t) sum_n wh_n P_n(t) + sum_n wp_n P_n(t) then needs no term table at all: the quadrature loop of
// for_each_B spends more scalar instructions (order tests, parameter loads, the family switch per order) than vector
// ones - the scalar unit, one per CU, was the busiest pipe of the integrated kernels (2.6e9 SALU against 2.0e9 VALU
// wave-instructions per bisection launch at C2a).  The weights come from per-sample slots with the normalisation
// constants a_n folded in (dense_weights); the family and the rectifier are fixed outside the node loop.
// ---------------------------------------------------------------------------
#ifndef TTM_DENSE_NODES
#define TTM_DENSE_NODES 5
#endif
TTM_HD bool dense_B(const Comp& c) { return c.nB_st == 0 && c.nB_hf == c.maxP_hf && c.nB_poly == c.maxP_poly; }

// weights of the dense B set in the sample's slots: [a_n w_n (Hermite functions) | w_n (polynomials) | w_none]
template <class R, class XA, class Slots>
TTM_HD void dense_weights(const Comp& c, const Prog& p, XA& x, Slots& w) {
    mon_weights<R>(c, p, x, w);
    for (int b = 0; b < c.nB_hf; ++b) w.set(b, c.dpar[TTM_UNI(c.bfuns[4 * b + 2])] * w.get(b));
}

template <int FAM, bool DER, class R, class W, class W0>
TTM_HD void g_eval_dense(int fam, int Ph, int Pp, const R& t, const W& w, const W0& w0, R& g, R& dg) {
    const int F = (FAM >= 0) ? FAM : fam;
    const int P = Ph > Pp ? Ph : Pp;
    R ah(0.0), dah(0.0), ap(w0), dap(0.0);
    R pm(1.0), dpm(0.0), pn, dp;
    poly_first(F, t, pn, dp);
    for (int n = 1; n <= P; ++n) {
        if (n <= Ph) {
            const auto wn = w.get(n - 1);
            ah = vfma(wn, pn, ah);
            if (DER) dah = vfma(wn, dp, dah);
        }
        if (n <= Pp) {
            const auto wn = w.get(Ph + n - 1);
            ap = vfma(wn, pn, ap);
            if (DER) dap = vfma(wn, dp, dap);
        }
        if (n < P) poly_next<DER>(F, n, t, pm, pn, dpm, dp);
    }
    g = ap; dg = dap;
    if (Ph > 0) {
        const R E = fast_exp(-0.25 * (t * t));
        g = vfma(ah, E, ap);
        if (DER) dg = vfma(E, dah - 0.5 * (t * ah), dap);    // d/dt [P e^{-t^2/4}] = e^{-t^2/4} (P' - t P / 2)
    }
}

// int_0^{xk} (r(g(t)) + delta) dt, node order and grouping of TM:4238-4258
template <int FAM, int RECT, class R, class W, class W0>
TTM_HD R integrate_rect_dense(const Prog& p, int Ph, int Pp, const R& xk, const W& w, const W0& w0) {
    const int rect = (RECT >= 0) ? RECT : p.rect;
    const R half = xk * 0.5;
    R res(0.0);
    int q = 0;
    if constexpr (lanes_of<R>::value == 1) {
        // one sample per lane (root searches, objective): TTM_DENSE_NODES quadrature nodes are evaluated together as a
        // short vector - the scalar instructions of a node (loop control, the 64-bit constants of exp / log, which the
        // compiler re-materialises with two s_mov each per use) are then paid once per group; terms are still added in
        // node order
        typedef VecD<TTM_DENSE_NODES> V;
        for (; q + TTM_DENSE_NODES <= p.Q; q += TTM_DENSE_NODES) {
            V t;
#pragma unroll
            for (int e = 0; e < TTM_DENSE_NODES; ++e) t[e] = half * p.qx[q + e] + half;
            V g, dg;
            g_eval_dense<FAM, false>(p.family, Ph, Pp, t, w, w0, g, dg);
            const V r = rect_eval(rect, g);
#pragma unroll
            for (int e = 0; e < TTM_DENSE_NODES; ++e) {
                const R term = half * (p.qw[q + e] * (r[e] + p.delta));
                res = (q + e == 0) ? term : res + term;
            }
        }
    }
    for (; q < p.Q; ++q) {
        const R t = half * p.qx[q] + half;
        R g, dg;
        g_eval_dense<FAM, false>(p.family, Ph, Pp, t, w, w0, g, dg);
        const R fr = rect_eval(rect, g) + p.delta;
        const R term = half * (p.qw[q] * fr);
        res = (q == 0) ? term : res + term;
    }
    return res;
}

// mon_eval for a dense B set (w: dense_weights)
template <int MONO, bool DER, class R, class W>
TTM_HD void mon_eval_dense(const Comp& c, const Prog& p, const R& t, const W& w, R& m, R& dm) {
    const int mono = (MONO >= 0) ? MONO : p.mono;
    const int Ph = c.maxP_hf, Pp = c.maxP_poly;
    const auto w0 = w.get(Ph + Pp);
    if (mono == TTM_MONO_SEPARABLE) {
        g_eval_dense<-1, DER>(p.family, Ph, Pp, t, w, w0, m, dm);
        return;
    }
    const bool common = p.family == TTM_FAM_HERMITE_E && p.rect == TTM_RECT_SOFTPLUS;      // the reference's defaults
    m = common ? integrate_rect_dense<TTM_FAM_HERMITE_E, TTM_RECT_SOFTPLUS>(p, Ph, Pp, t, w, w0)
               : integrate_rect_dense<-1, -1>(p, Ph, Pp, t, w, w0);
    dm = R(0.0);
    if (DER) {
        R g, dg;
        g_eval_dense<-1, false>(p.family, Ph, Pp, t, w, w0, g, dg);
        dm = rect_eval(p.rect, g) + p.delta;
    }
}

// monotone part of S_k at x_k = t given the sample's weights: value and dS/dx_k
// MONO >= 0 fixes the monotonicity mode at compile time (kernels), MONO < 0 reads it from the program
template <int MONO, bool DER, class R, class W>
TTM_HD void mon_eval(const Comp& c, const Prog& p, const R& t, const W& w, R& m, R& dm) {
    const int mono = (MONO >= 0) ? MONO : p.mono;
    if (mono == TTM_MONO_SEPARABLE) {
        g_eval<DER>(c, p, t, w, m, dm);
    } else {
        m = integrate_rect(c, p, t, w);
        dm = R(0.0);
        if (DER) {
            R g, dg;
            g_eval<false>(c, p, t, w, g, dg);
            dm = rect_eval(p.rect, g) + p.delta;
        }
    }
}

// weights of a dense B set as a type of their own: the root searches take any weight source and evaluate through
// mon_eval - with DenseSet<W> that is the table-free evaluator
template <class W>
struct DenseSet {
    const W& w;
};
template <int MONO, bool DER, class R, class W>
TTM_HD void mon_eval(const Comp& c, const Prog& p, const R& t, const DenseSet<W>& d, R& m, R& dm) {
    mon_eval_dense<MONO, DER>(c, p, t, d.w, m, dm);
}

// ---------------------------------------------------------------------------
// per-sample bodies
// ---------------------------------------------------------------------------

// S_k(x) and dS_k/dx_k.  scratch: nB+1 slots (touched only for components with monotone cross terms).
template <int MONO, bool DER, class R, class XA, class Slots>
TTM_HD void sample_forward(const Comp& c, const Prog& p, VarCache<XA, R>& x, Slots& w, bool want_value, R& S, R& dS) {
    R m, dm;
    const R xk = x.get(c.kc);
    const int mono = (MONO >= 0) ? MONO : p.mono;
    if (mono == TTM_MONO_INTEGRATED && dense_B(c)) {
        dense_weights<R>(c, p, x, w);
        const DenseSet<Slots> dw{w};
        mon_eval<MONO, DER>(c, p, xk, dw, m, dm);
    } else if (c.n_mnt == 0 && c.n_xgrp == 0) {
        const UniformW uw{c.fold + c.off_wb};
        mon_eval<MONO, DER>(c, p, xk, uw, m, dm);
    } else {
        mon_weights<R>(c, p, x, w);
        mon_eval<MONO, DER>(c, p, xk, w, m, dm);
    }
    S = want_value ? nonmon_sum<R>(c, p, x) + m : m;
    dS = dm;
}

// basis rows (inspection): which 0 Psi_nonmon, 1 Psi_mon, 2 dPsi_mon/dx_k ; out(i, value)
template <class XA, class Out>
TTM_HD void sample_basis(const Comp& c, const Prog& p, int which, XA& x, Out&& out) {
    if (which == 0) {
        for (int i = 0; i < c.n_nm; ++i) {
            cint_p T = c.nm_terms + 4 * i;
            out(TTM_UNI(T[3]), eval_A<double>(T, c, p, x));
        }
        return;
    }
    const double xk = x(c.kc);
    // every distinct x_k-function once; the terms that carry it are found by a scalar scan of the term table
    for_each_B<true>(c, p, xk, [&](int b, double bv, double bdv) {
        for (int i = 0; i < c.n_mon; ++i) {
            cint_p T = c.mon_terms + 4 * i;
            if (TTM_UNI(T[2]) == b) out(TTM_UNI(T[3]), eval_A<double>(T, c, p, x) * (which == 1 ? bv : bdv));
        }
    });
    for (int i = 0; i < c.n_mon; ++i) {                      // terms without a factor in x_k: value 1, derivative 0
        cint_p T = c.mon_terms + 4 * i;
        if (TTM_UNI(T[2]) < 0) out(TTM_UNI(T[3]), eval_A<double>(T, c, p, x) * (which == 1 ? 1.0 : 0.0));
    }
}

// Objective + gradient contribution of one sample, integrated rectifier
// (TM:3343-3376, 3475-3569).  acc layout: [0] J, [1..n_nm] d/dc_nonmon, then d/dc_mon.
// scratch slots: w (nB+1) | Bv (nB+1) | I (nB+1)
// Members of a folded series {var, P, fold offset, has_hf} (a nonmonotone group or a cross group): f(ci, value) for every
// coefficient ci that the fold recipe sums into the series, with the value of its term - one recurrence and the cached
// exp(-x^2/4) of the variable for the whole group instead of a walk through term and factor records per term.
template <class XA, class F>
TTM_HD void for_each_member(const Comp& c, const Prog& p, VarCache<XA, double>& x, int var, int P, int off, int has_hf, F&& f) {
    double xv, e = 0.0;
    if (has_hf) x.get_e(var, xv, e); else xv = x.get(var);
    double pm = 1.0, dpm = 0.0, pn, dp;
    poly_first(p.family, xv, pn, dp);
    for (int n = 1; n <= P; ++n) {
        int s0 = TTM_UNI(c.fslot[2 * (off + n - 1)]), ns = TTM_UNI(c.fslot[2 * (off + n - 1) + 1]);
        for (int j = 0; j < ns; ++j) f(TTM_UNI(c.fsrc[2 * (s0 + j)]), pn);
        if (has_hf) {
            s0 = TTM_UNI(c.fslot[2 * (off + P + n - 1)]); ns = TTM_UNI(c.fslot[2 * (off + P + n - 1) + 1]);
            for (int j = 0; j < ns; ++j)
                f(TTM_UNI(c.fsrc[2 * (s0 + j)]), (c.dpar[TTM_UNI(c.fsrc[2 * (s0 + j) + 1])] * pn) * e);
        }
        if (n < P) poly_next<false>(p.family, n, xv, pm, pn, dpm, dp);
    }
}

// gradient contributions of one sample once S, the integrals I_b, the values B_b(x_k) and r'/(r + delta) are known
// (TM:3475-3569): d/dc_nonmon,i = S Psi_i ; d/dc_mon,i = A_i (S I_b - rinv B_b)
template <class XA, class Slots, class Acc>
TTM_HD void objective_gradient(const Comp& c, const Prog& p, VarCache<XA, double>& x, double S, double rinv, Slots& Bv, Slots& I, Acc& acc) {
    if (c.fslot) {
        // through the fold recipe: constants, nonmonotone groups, generic nonmonotone terms
        const int s0 = TTM_UNI(c.fslot[0]), ns = TTM_UNI(c.fslot[1]);
        for (int j = 0; j < ns; ++j) acc.add(1 + TTM_UNI(c.fsrc[2 * (s0 + j)]), S);
        for (int g = 0; g < c.n_grp; ++g) {
            cint_p G = c.grp + 4 * g;
            for_each_member(c, p, x, TTM_UNI(G[0]), TTM_UNI(G[1]), TTM_UNI(G[2]), TTM_UNI(G[3]),
                            [&](int ci, double v) { acc.add(1 + ci, S * v); });
        }
        for (int i = 0; i < c.n_gen; ++i) {
            cint_p T = c.nm_terms + 4 * TTM_UNI(c.gen[i]);
            acc.add(1 + TTM_UNI(T[3]), S * eval_A<double>(T, c, p, x));
        }
        // monotone terms of x_k alone, cross groups, generic cross terms
        for (int i = 0; i < c.n_mon; ++i) {
            cint_p T = c.mon_terms + 4 * i;
            if (TTM_UNI(T[1]) == 0) {
                int b = TTM_UNI(T[2]);
                if (b < 0) b = c.nB;
                acc.add(1 + c.n_nm + TTM_UNI(T[3]), S * I.get(b) - rinv * Bv.get(b));
            }
        }
        for (int g = 0; g < c.n_xgrp; ++g) {
            cint_p G = c.xgrp + 8 * g;
            const int b = TTM_UNI(G[4]);
            const double fac = S * I.get(b) - rinv * Bv.get(b);
            for_each_member(c, p, x, TTM_UNI(G[0]), TTM_UNI(G[1]), TTM_UNI(G[2]), TTM_UNI(G[3]),
                            [&](int ci, double v) { acc.add(1 + ci, v * fac); });
        }
        for (int j = 0; j < c.n_mnt; ++j) {
            cint_p T = c.mon_terms + 4 * TTM_UNI(c.mnt[j]);
            int b = TTM_UNI(T[2]);
            if (b < 0) b = c.nB;
            acc.add(1 + c.n_nm + TTM_UNI(T[3]), eval_A<double>(T, c, p, x) * (S * I.get(b) - rinv * Bv.get(b)));
        }
        return;
    }
    for (int i = 0; i < c.n_nm; ++i) {
        cint_p T = c.nm_terms + 4 * i;
        acc.add(1 + TTM_UNI(T[3]), S * eval_A<double>(T, c, p, x));
    }
    for (int i = 0; i < c.n_mon; ++i) {
        cint_p T = c.mon_terms + 4 * i;
        const int nf = TTM_UNI(T[1]);
        int b = TTM_UNI(T[2]);
        if (b < 0) b = c.nB;
        const double a = (nf == 0) ? 1.0 : eval_A<double>(T, c, p, x);
        acc.add(1 + c.n_nm + TTM_UNI(T[3]), a * (S * I.get(b) - rinv * Bv.get(b)));
    }
}

// The same for a component with a dense B set: per group of TTM_DENSE_NODES quadrature nodes one pass of the
// recurrence for g, one for the integrals of cq B_b (the polynomial values are cheaper to recompute than to keep);
// sums over the nodes of a group are taken in node order.
template <int FAM, int RECT, class XA, class Slots, class Acc>
TTM_HD void sample_objective_int_dense(const Comp& c, const Prog& p, VarCache<XA, double>& x, Slots& w, Slots& Bv, Slots& I, Acc& acc) {
    typedef VecD<TTM_DENSE_NODES> V;
    const int F = (FAM >= 0) ? FAM : p.family;
    const int rect = (RECT >= 0) ? RECT : p.rect;
    const int Ph = c.maxP_hf, Pp = c.maxP_poly, P = Ph > Pp ? Ph : Pp;
    dense_weights<double>(c, p, x, w);
    const double xk = x.get(c.kc);
    const double half = xk * 0.5;
    double mono = 0.0;
    for (int b = 0; b <= c.nB; ++b) I.set(b, 0.0);
    auto nodes = [&](auto tag, int q) {
        typedef decltype(tag) T;                                // V or double
        constexpr int L = lanes_of<T>::value;
        T t;
#pragma unroll
        for (int e = 0; e < L; ++e) set_elem(t, e, half * p.qx[q + e] + half);
        T E(1.0);
        if (Ph > 0) E = fast_exp(-0.25 * (t * t));
        T ah(0.0), ap(w.get(Ph + Pp));
        {
            T pm(1.0), dpm(0.0), pn, dp;
            poly_first(F, t, pn, dp);
            for (int n = 1; n <= P; ++n) {
                if (n <= Ph) ah = vfma(w.get(n - 1), pn, ah);
                if (n <= Pp) ap = vfma(w.get(Ph + n - 1), pn, ap);
                if (n < P) poly_next<false>(F, n, t, pm, pn, dpm, dp);
            }
        }
        const T g = vfma(ah, E, ap);
        T r, dr, logr;
        rect_all(rect, 0.0, g, r, dr, logr);                    // logr unused here
        T cq;
        double csum = 0.0;
#pragma unroll
        for (int e = 0; e < L; ++e) {
            const double term = half * (p.qw[q + e] * (elem(r, e) + p.delta));
            mono = (q + e == 0) ? term : mono + term;
            const double ce = (half * p.qw[q + e]) * elem(dr, e);      // lim_dif*0.5*W_q * r'(g_q)   (TM:4264-4278, 5127-5133)
            set_elem(cq, e, ce);
            csum += ce;
        }
        const T cqE = cq * E;
        {
            T pm(1.0), dpm(0.0), pn, dp;
            poly_first(F, t, pn, dp);
            for (int n = 1; n <= P; ++n) {
                if (n <= Ph) {
                    const T v = cqE * pn;
                    double sh = 0.0;
#pragma unroll
                    for (int e = 0; e < L; ++e) sh += elem(v, e);
                    I.set(n - 1, I.get(n - 1) + sh);
                }
                if (n <= Pp) {
                    const T v = cq * pn;
                    double sp = 0.0;
#pragma unroll
                    for (int e = 0; e < L; ++e) sp += elem(v, e);
                    I.set(Ph + n - 1, I.get(Ph + n - 1) + sp);
                }
                if (n < P) poly_next<false>(F, n, t, pm, pn, dpm, dp);
            }
        }
        I.set(c.nB, I.get(c.nB) + csum);
    };
    int q = 0;
    for (; q + TTM_DENSE_NODES <= p.Q; q += TTM_DENSE_NODES) nodes(V(0.0), q);
    for (; q < p.Q; ++q) nodes(0.0, q);
    // the Hermite-function integrals carry their normalisation constants from here on
    for (int b = 0; b < c.nB_hf; ++b) I.set(b, c.dpar[TTM_UNI(c.bfuns[4 * b + 2])] * I.get(b));
    const double S = nonmon_sum<double>(c, p, x) + mono;
    // values at x_k for the log term (the weights are not needed after g: Bv may be the very columns of w - k_objective
    // passes them so and saves a third of the scratch)
    double g, dg;
    g_eval_dense<FAM, false>(p.family, Ph, Pp, xk, w, w.get(Ph + Pp), g, dg);
    for_each_B<false>(c, p, xk, [&](int b, double v, double) { Bv.set(b, v); });
    Bv.set(c.nB, 1.0);
    double r, dr, logr;
    rect_all(rect, p.delta, g, r, dr, logr);
    acc.add(0, 0.5 * S * S - logr);
    const double rinv = dr * fast_rcp(r + p.delta);
    objective_gradient(c, p, x, S, rinv, Bv, I, acc);
}

template <class XA, class Slots, class Acc>
TTM_HD void sample_objective_int(const Comp& c, const Prog& p, VarCache<XA, double>& x, Slots& w, Slots& Bv, Slots& I, Acc& acc) {
    if (dense_B(c)) {
        if (p.family == TTM_FAM_HERMITE_E && p.rect == TTM_RECT_SOFTPLUS)
            sample_objective_int_dense<TTM_FAM_HERMITE_E, TTM_RECT_SOFTPLUS>(c, p, x, w, Bv, I, acc);
        else
            sample_objective_int_dense<-1, -1>(c, p, x, w, Bv, I, acc);
        return;
    }
    mon_weights<double>(c, p, x, w);
    const double xk = x.get(c.kc);
    const double half = xk * 0.5;
    double mono = 0.0;
    for (int b = 0; b <= c.nB; ++b) I.set(b, 0.0);
    for (int q = 0; q < p.Q; ++q) {
        const double t = half * p.qx[q] + half;
        double g = w.get(c.nB);
        for_each_B<false>(c, p, t, [&](int b, double v, double) {
            g = fma(w.get(b), v, g);
            Bv.set(b, v);
        });
        double r, dr, logr;
        rect_all(p.rect, 0.0, g, r, dr, logr);          // logr unused here
        const double term = half * (p.qw[q] * (r + p.delta));
        mono = (q == 0) ? term : mono + term;
        const double cq = (half * p.qw[q]) * dr;       // lim_dif*0.5*W_q * r'(g_q)   (TM:4264-4278, 5127-5133)
        for (int b = 0; b < c.nB; ++b) I.set(b, fma(cq, Bv.get(b), I.get(b)));
        I.set(c.nB, I.get(c.nB) + cq);
    }
    const double S = nonmon_sum<double>(c, p, x) + mono;
    // values at x_k for the log term
    double g = w.get(c.nB);
    for_each_B<false>(c, p, xk, [&](int b, double v, double) {
        g = fma(w.get(b), v, g);
        Bv.set(b, v);
    });
    Bv.set(c.nB, 1.0);
    double r, dr, logr;
    rect_all(p.rect, p.delta, g, r, dr, logr);
    acc.add(0, 0.5 * S * S - logr);
    const double rinv = dr * fast_rcp(r + p.delta);
    objective_gradient(c, p, x, S, rinv, Bv, I, acc);
}

// Separable objective pieces of one sample (TM:2990-3006):
// acc[0] += log dS, acc[1+i] += dPsi_i / dS with dS = dPsi.c + delta * rowsum(dPsi)
// scratch: dB (nB+1)
template <class XA, class Slots, class Acc>
TTM_HD void sample_objective_sep(const Comp& c, const Prog& p, VarCache<XA, double>& x, Slots& dB, Acc& acc) {
    const double xk = x.get(c.kc);
    for_each_B<true>(c, p, xk, [&](int b, double, double dv) { dB.set(b, dv); });
    dB.set(c.nB, 0.0);
    double dS = 0.0, rowsum = 0.0;
    for (int i = 0; i < c.n_mon; ++i) {
        cint_p T = c.mon_terms + 4 * i;
        const int nf = TTM_UNI(T[1]);
        int b = TTM_UNI(T[2]);
        if (b < 0) b = c.nB;
        const double a = (nf == 0) ? 1.0 : eval_A<double>(T, c, p, x);
        const double d = a * dB.get(b);
        dS = fma(c.cmon[TTM_UNI(T[3])], d, dS);
        rowsum += d;
    }
    dS += rowsum * p.delta;
    acc.add(0, fast_log(dS));
    const double inv = fast_rcp(dS);
    for (int i = 0; i < c.n_mon; ++i) {
        cint_p T = c.mon_terms + 4 * i;
        const int nf = TTM_UNI(T[1]);
        int b = TTM_UNI(T[2]);
        if (b < 0) b = c.nB;
        const double a = (nf == 0) ? 1.0 : eval_A<double>(T, c, p, x);
        acc.add(1 + TTM_UNI(T[3]), a * dB.get(b) * inv);
    }
}

// ---------------------------------------------------------------------------
// fast path: components whose terms are all univariate (no cross terms, no special terms in the
// nonmonotone list).  Everything is read from two flat streams at offsets known from one 12-int
// descriptor, so the scalar loads of the next record are issued before the current record's
// arithmetic (the generic interpreter chases header -> record -> parameters pointers instead).
// ---------------------------------------------------------------------------

struct FastComp {
    cint_p gi;        // groups {var, P, alpha offset, TTM_PLAN_* flags}, then the special-term kinds
    cdbl_p fold;      // the component's folded coefficients
    cdbl_p stream;    // wHF | wPoly | special-term records (5 doubles, fold stage 1)
    cdbl_p st8;       // unified special-term section (fold stage 2)
    int kc, n_grp, n_st, n_stA, maxP_hf, maxP_poly;
};

TTM_HD FastComp make_fast(cint_p fd, cint_p fints, cdbl_p fold_all, int fold_base) {
    FastComp f;
    f.kc = fd[TTM_FD_KC];
    f.n_grp = fd[TTM_FD_N_GRP];
    f.n_st = fd[TTM_FD_N_ST];
    f.n_stA = fd[TTM_FD_N_STA];
    f.maxP_hf = fd[TTM_FD_MAXP_HF];
    f.maxP_poly = fd[TTM_FD_MAXP_POLY];
    f.gi = fints + fd[TTM_FD_FINT_OFF];
    f.fold = fold_all + (fd[TTM_FD_FOLD_OFF] - fold_base);
    f.stream = f.fold + fd[TTM_FD_STREAM];
    f.st8 = f.fold + fd[TTM_FD_ST8];
    return f;
}

// run-time tagged cache (VarCache) behind the interface the group evaluator wants
template <class XA, class R>
struct TaggedFetch {
    VarCache<XA, R>& c;
    TTM_HD void fetch(int var, int fl, R& x, R& e) {
        if (fl & TTM_PLAN_HF) c.get_e(var, x, e); else x = c.get(var);
    }
};

// statically planned cache (termtable.py:_plan_column_cache): the flag word of the group record says where
// the column lives, no tag compares.  Way w keeps its column in slot 2w and exp(-x^2/4) in slot 2w+1.
template <class XA, class R, class ST = CacheStore<R>>
struct PlanCache {
    const XA& xa;
    ST st;
    TTM_HD PlanCache(const XA& x, const ST& store) : xa(x), st(store) {}
    // contents on entry to a component (state words: column | TTM_PLAN_E, -1 = empty)
    TTM_HD void warm(cint_p state) {
        for (int w = 0; w < TTM_PLAN_WAYS; ++w) {
            const int v = state[w];
            if (v >= 0) {
                const R x = xa(v & ~TTM_PLAN_E);
                st.set(2 * w, x);
                // (the second half of the slot is defined either way: hot-record evaluators read both halves, see h_component)
                st.set(2 * w + 1, (v & TTM_PLAN_E) ? (st.etab ? exp_q_tab(st.etab, x) : exp_q_fast(x)) : R(0.0));
            }
        }
    }
    TTM_HD void put(int slot, const R& x) { if (slot >= 0) st.set(2 * slot, x); }
    TTM_HD void fetch(int var, int fl, R& x, R& e) {
        const int slot = TTM_PLAN_SLOT(fl);
        if (fl & TTM_PLAN_XHIT) {
            x = st.get(2 * slot);
        } else {
            x = xa(var);
            if (slot != 255) st.set(2 * slot, x);
        }
        if (fl & TTM_PLAN_HF) {
            if (fl & TTM_PLAN_EHIT) {
                e = st.get(2 * slot + 1);
            } else {
                e = exp_q_fast(x);
                if (slot != 255) st.set(2 * slot + 1, e);
            }
        }
    }
};

// sum_n alpha_n P_n(x) + e * sum_n beta_n P_n(x) for a group of PN orders, straight-line
// (FAM >= 0: polynomial family known at compile time)
template <int FAM, int PN, bool HF, class R>
TTM_HD void group_fixed(int fam, cdbl_p al, const R& x, const R& e, R& s) {
    const int F = (FAM >= 0) ? FAM : fam;
    double a[PN], b[PN];
#pragma unroll
    for (int j = 0; j < PN; ++j) { a[j] = al[j]; b[j] = HF ? al[PN + j] : 0.0; }
    R pm(1.0), dpm(0.0), pn, dp, accp(0.0), acch(0.0);
    poly_first(F, x, pn, dp);
#pragma unroll
    for (int n = 1; n <= PN; ++n) {
        accp = vfma(a[n - 1], pn, accp);
        if (HF) acch = vfma(b[n - 1], pn, acch);
        if (n < PN) poly_next<false>(F, n, x, pm, pn, dpm, dp);
    }
    s = s + accp;
    if (HF) s = vfma(e, acch, s);
}

// nonmonotone part of a fast-path component: constant + univariate groups
template <int FAM, class R, class Fetch>
TTM_HD R nonmon_sum_fast(const FastComp& f, const Prog& p, Fetch& x) {
    R s(f.fold[0]);
    if (f.n_grp == 0) return s;
    // software-pipelined over the groups: while group g is evaluated, the record of group g+1 has been
    // loaded and its column (and exp(-x^2/4)) fetched from the per-thread cache - independent work that
    // fills the latency of the current group's dependent FMA chain
    int var = f.gi[0], P = f.gi[1], aoff = f.gi[2], fl = f.gi[3];
    R xv, e(0.0);
    x.fetch(var, fl, xv, e);
    for (int g = 0; g < f.n_grp; ++g) {
        const int cP = P, chf = fl & TTM_PLAN_HF;
        cdbl_p al = f.fold + aoff;
        const R cx = xv, ce = e;
        if (g + 1 < f.n_grp) {
            cint_p G = f.gi + 4 * (g + 1);
            var = G[0]; P = G[1]; aoff = G[2]; fl = G[3];
            x.fetch(var, fl, xv, e);
        }
        switch (cP + (chf ? 8 : 0)) {
            case 1: group_fixed<FAM, 1, false>(p.family, al, cx, ce, s); break;
            case 2: group_fixed<FAM, 2, false>(p.family, al, cx, ce, s); break;
            case 3: group_fixed<FAM, 3, false>(p.family, al, cx, ce, s); break;
            case 4: group_fixed<FAM, 4, false>(p.family, al, cx, ce, s); break;
            case 9: group_fixed<FAM, 1, true>(p.family, al, cx, ce, s); break;
            case 10: group_fixed<FAM, 2, true>(p.family, al, cx, ce, s); break;
            case 11: group_fixed<FAM, 3, true>(p.family, al, cx, ce, s); break;
            case 12: group_fixed<FAM, 4, true>(p.family, al, cx, ce, s); break;
            default: {
                const int F = (FAM >= 0) ? FAM : p.family;
                cdbl_p be = al + cP;
                R pm(1.0), dpm(0.0), pn, dp, accp(0.0), acch(0.0);
                poly_first(F, cx, pn, dp);
                for (int n = 1; n <= cP; ++n) {
                    accp = vfma(al[n - 1], pn, accp);
                    if (chf) acch = vfma(be[n - 1], pn, acch);
                    if (n < cP) poly_next<false>(F, n, cx, pm, pn, dpm, dp);
                }
                s = s + accp;
                if (chf) s = vfma(ce, acch, s);
            }
        }
    }
    return s;
}

// unified special-term records {centre, 1/(sqrt2 scale), A1, B0, B1, G, DG, DT} (TTM_FD_ST8):
// branch-free; the next record is loaded while the current one is evaluated (the folded array is padded
// by one record, so the read-ahead of the last iteration stays inside the allocation).
// EDGE: records of LET / RET / RBF (value needs the Gaussian); otherwise iRBF records (A1, DG only).
template <bool EDGE, bool DER, class R>
TTM_HD void st8_accumulate(const Prog& p, cdbl_p rec, int n, const R& t, R& acc, R& dacc) {
    if (n <= 0) return;
    double mu = rec[0], inv = rec[1], A1 = rec[2], B0 = rec[3], B1 = rec[4], G = rec[5], DG = rec[6], DT = rec[7];
    for (int s = 0; s < n; ++s) {
        const double cmu = mu, cinv = inv, cA1 = A1, cB0 = B0, cB1 = B1, cG = G, cDG = DG, cDT = DT;
        cdbl_p nx = rec + 8 * (s + 1);
        mu = nx[0]; inv = nx[1]; A1 = nx[2]; B0 = nx[3]; B1 = nx[4]; G = nx[5]; DG = nx[6]; DT = nx[7];
        const R d = t - cmu;
        const R tt = d * cinv;
        R e, gs;
        erf_gauss_tab<(EDGE || DER)>(p.erf_tab, tt, e, gs);
        if (EDGE) {
            const R h = vfma(e, cB1, cB0);
            acc = vfma(d, h, acc);
            acc = vfma(cG, gs, acc);
            if (DER) {
                dacc = dacc + h;
                dacc = vfma(vfma(tt, cDT, cDG), gs, dacc);
            }
        } else {
            acc = vfma(cA1, e, acc);
            if (DER) dacc = vfma(cDG, gs, dacc);
        }
    }
}

// g(t) = w_none + sum_n wHF[n] P_n(t) e^{-t^2/4} + sum_n wPoly[n] P_n(t) + sum_s w_s ST_s(t), and dg/dt
template <int FAM, bool DER, class R>
TTM_HD void g_eval_fast(const FastComp& f, const Prog& p, const R& t, R& g, R& dg) {
    const int F = (FAM >= 0) ? FAM : p.family;
    R acc(f.st8[0]), dacc(0.0);
    cdbl_p w = f.stream;
    if (f.maxP_hf > 0) {
        const R E = fast_exp(-0.25 * (t * t));
        R pm(1.0), dpm(0.0), pn, dp, a(0.0), da(0.0);
        poly_first(F, t, pn, dp);
        for (int n = 1; n <= f.maxP_hf; ++n) {
            const double wn = w[n - 1];                      // a_n already folded in
            a = vfma(wn, pn, a);
            if (DER) da = vfma(wn, dp, da);
            if (n < f.maxP_hf) poly_next<DER>(F, n, t, pm, pn, dpm, dp);
        }
        acc = vfma(a, E, acc);
        if (DER) dacc = vfma(E, da - 0.5 * (t * a), dacc);   // d/dt [P e^{-t^2/4}] = e^{-t^2/4} (P' - t P / 2)
        w += f.maxP_hf;
    }
    if (f.maxP_poly > 0) {
        R pm(1.0), dpm(0.0), pn, dp;
        poly_first(F, t, pn, dp);
        for (int n = 1; n <= f.maxP_poly; ++n) {
            const double wn = w[n - 1];
            acc = vfma(wn, pn, acc);
            if (DER) dacc = vfma(wn, dp, dacc);
            if (n < f.maxP_poly) poly_next<DER>(F, n, t, pm, pn, dpm, dp);
        }
    }
    st8_accumulate<true, DER>(p, f.st8 + 8, f.n_stA, t, acc, dacc);
    st8_accumulate<false, DER>(p, f.st8 + 8 + 8 * f.n_stA, f.n_st - f.n_stA, t, acc, dacc);
    g = acc; dg = dacc;
}

struct StreamW {             // uniform weights of a fast-path component: wHF | wPoly with the constants folded in
    cdbl_p s;
    TTM_HD double get(int i) const { return s[i]; }
};

template <int MONO, int FAM, bool DER, class R>
TTM_HD void mon_eval_fast(const FastComp& f, const Prog& p, const R& t, R& m, R& dm) {
    const int mono = (MONO >= 0) ? MONO : p.mono;
    if (mono == TTM_MONO_SEPARABLE) {
        g_eval_fast<FAM, DER>(f, p, t, m, dm);
    } else if (f.n_st == 0 && lanes_of<R>::value == 1) {
        // dense B set with uniform weights, one sample per lane: quadrature nodes in short vectors (integrate_rect_dense)
        const StreamW sw{f.stream};
        const double w0 = f.st8[0];
        const bool common = ((FAM >= 0) ? FAM : p.family) == TTM_FAM_HERMITE_E && p.rect == TTM_RECT_SOFTPLUS;
        m = common ? integrate_rect_dense<TTM_FAM_HERMITE_E, TTM_RECT_SOFTPLUS>(p, f.maxP_hf, f.maxP_poly, t, sw, w0)
                   : integrate_rect_dense<FAM, -1>(p, f.maxP_hf, f.maxP_poly, t, sw, w0);
        dm = R(0.0);
        if (DER) {
            R g, dg;
            g_eval_fast<FAM, false>(f, p, t, g, dg);
            dm = rect_eval(p.rect, g) + p.delta;
        }
    } else {
        const R half = t * 0.5;
        R res(0.0);
        for (int q = 0; q < p.Q; ++q) {
            const R tq = half * p.qx[q] + half;
            R g, dg;
            g_eval_fast<FAM, false>(f, p, tq, g, dg);
            const R fr = rect_eval(p.rect, g) + p.delta;
            const R term = half * (p.qw[q] * fr);
            res = (q == 0) ? term : res + term;
        }
        m = res;
        dm = R(0.0);
        if (DER) {
            R g, dg;
            g_eval_fast<FAM, false>(f, p, t, g, dg);
            dm = rect_eval(p.rect, g) + p.delta;
        }
    }
}

// S and dS/dx_k of a fast-path component; xk = this thread's x_kc, x = column cache (TaggedFetch / PlanCache)
template <int MONO, int FAM, bool DER, class R, class Fetch>
TTM_HD void sample_forward_fast(const FastComp& f, const Prog& p, const R& xk, Fetch& x, bool want_value, R& S, R& dS) {
    R m, dm;
    mon_eval_fast<MONO, FAM, DER>(f, p, xk, m, dm);
    S = want_value ? nonmon_sum_fast<FAM, R>(f, p, x) + m : m;
    dS = dm;
}

// Bisection root search of one sample for one component (TM:3842-3976).
// Returns the last trial point (what the reference leaves in X[:, kc]) and the
// number of midpoint iterations it needed.  `cap` < 0: no cap.
template <int MONO, class W>
TTM_HD double sample_bisect(const Comp& c, const Prog& p, double off, double zk, const W& w, int cap, int& iters) {
    double lo = -2.0, hi = 2.0, m, dm;
    mon_eval<MONO, false>(c, p, lo, w, m, dm);
    double flo = (off + m) - zk;
    mon_eval<MONO, false>(c, p, hi, w, m, dm);
    double fhi = (off + m) - zk;
    double last = hi;
    if (flo > fhi) { double t = flo; flo = fhi; fhi = t; t = lo; lo = hi; hi = t; }
    // window shifts (TM:3894-3941); bounded so that every wave terminates
    for (int guard = 0; guard < 2000 && (flo * fhi > 0.0); ++guard) {
        if (flo > fhi) { double t = flo; flo = fhi; fhi = t; t = lo; lo = hi; hi = t; }
        const double diff = hi - lo;
        if (flo > 0.0) {
            hi = lo; lo = lo - diff * 2.0;
            fhi = flo;
            last = lo;
            mon_eval<MONO, false>(c, p, lo, w, m, dm);
            flo = (off + m) - zk;
        } else if (flo < 0.0) {
            lo = hi; hi = hi + diff * 2.0;
            flo = fhi;
            last = hi;
            mon_eval<MONO, false>(c, p, hi, w, m, dm);
            fhi = (off + m) - zk;
        } else {
            break;
        }
    }
    iters = 0;
    const int maxit = (cap >= 0 && cap < 100) ? cap : 100;
    while (iters < maxit) {
        ++iters;
        const double mid = (lo + hi) * 0.5;     // np.mean over two values
        last = mid;
        mon_eval<MONO, false>(c, p, mid, w, m, dm);
        const double fm = (off + m) - zk;
        if (fm < 0.0) lo = mid;
        if (fm > 0.0) hi = mid;
        if (!(fabs(fm) > 1e-9)) break;
    }
    return last;
}

// Safeguarded Newton root search of one sample for one component - NOT the reference's method (SURVEY section 8a',
// K5 "newton"): same start (+-2), same window doubling and same stopping rule (|S - z| <= 1e-9, at most 100 trial points)
// as sample_bisect, but inside the bracket the next trial point is the Newton step with the analytic dS/dx_k (the
// rectified integrand itself for integrated maps), falling back to the midpoint whenever the step leaves the bracket.
// ~6 evaluations of S instead of ~33; converges to the same root, not to the same last midpoint.
template <int MONO, class W>
TTM_HD double sample_newton(const Comp& c, const Prog& p, double off, double zk, const W& w, int& iters) {
    double lo = -2.0, hi = 2.0, m, dm;
    mon_eval<MONO, false>(c, p, lo, w, m, dm);
    double flo = (off + m) - zk;
    mon_eval<MONO, false>(c, p, hi, w, m, dm);
    double fhi = (off + m) - zk;
    double last = hi;
    if (flo > fhi) { double t = flo; flo = fhi; fhi = t; t = lo; lo = hi; hi = t; }
    for (int guard = 0; guard < 2000 && (flo * fhi > 0.0); ++guard) {
        if (flo > fhi) { double t = flo; flo = fhi; fhi = t; t = lo; lo = hi; hi = t; }
        const double diff = hi - lo;
        if (flo > 0.0) {
            hi = lo; lo = lo - diff * 2.0;
            fhi = flo;
            last = lo;
            mon_eval<MONO, false>(c, p, lo, w, m, dm);
            flo = (off + m) - zk;
        } else if (flo < 0.0) {
            lo = hi; hi = hi + diff * 2.0;
            flo = fhi;
            last = hi;
            mon_eval<MONO, false>(c, p, hi, w, m, dm);
            fhi = (off + m) - zk;
        } else {
            break;
        }
    }
    iters = 0;
    if (!(flo * fhi <= 0.0)) return last;                     // no sign change (NaN, or a map that is not monotone)
    double x = (fhi != flo) ? lo - flo * fast_div(hi - lo, fhi - flo) : (lo + hi) * 0.5;     // secant start
    if (!(x > fmin(lo, hi) && x < fmax(lo, hi))) x = (lo + hi) * 0.5;
    while (iters < 100) {
        ++iters;
        last = x;
        mon_eval<MONO, true>(c, p, x, w, m, dm);
        const double f = (off + m) - zk;
        if (!(fabs(f) > 1e-9)) break;
        if (f < 0.0) lo = x; else hi = x;
        double xn = x - fast_div(f, dm);
        if (!(xn > fmin(lo, hi) && xn < fmax(lo, hi))) xn = (lo + hi) * 0.5;
        x = xn;
    }
    return last;
}

// root search of one sample for one component with the weight source the component calls for (kernels and the host
// test double share this dispatch)
template <int MONO, bool NEWTON, class XA, class Slots>
TTM_HD double sample_root(const Comp& c, const Prog& p, VarCache<XA, double>& x, Slots& w, double off, double zk, int cap, int& it) {
    const int mono = (MONO >= 0) ? MONO : p.mono;
    if (mono == TTM_MONO_INTEGRATED && dense_B(c)) {
        dense_weights<double>(c, p, x, w);
        const DenseSet<Slots> dw{w};
        return NEWTON ? sample_newton<MONO>(c, p, off, zk, dw, it) : sample_bisect<MONO>(c, p, off, zk, dw, cap, it);
    }
    if (c.n_mnt == 0 && c.n_xgrp == 0) {
        const UniformW uw{c.fold + c.off_wb};
        return NEWTON ? sample_newton<MONO>(c, p, off, zk, uw, it) : sample_bisect<MONO>(c, p, off, zk, uw, cap, it);
    }
    mon_weights<double>(c, p, x, w);
    return NEWTON ? sample_newton<MONO>(c, p, off, zk, w, it) : sample_bisect<MONO>(c, p, off, zk, w, cap, it);
}

// interp1d lookup (TM:4062-4082): xs non-decreasing table of map outputs, ys the abscissae
TTM_HD double table_lookup(const double* xs, const double* ys, int T, double target) {
    // np.searchsorted(xs, target) (left): first i with xs[i] >= target
    int lo = 0, hi = T;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (xs[mid] < target) lo = mid + 1; else hi = mid;
    }
    int i = lo < 1 ? 1 : (lo > T - 1 ? T - 1 : lo);
    const double x_lo = xs[i - 1], x_hi = xs[i], y_lo = ys[i - 1], y_hi = ys[i];
    // (x_hi == x_lo can only happen at the clamped start of a table with a flat tail, where target == x_lo: interp1d's
    // inf * 0 = NaN there is an accident of rounding - which of two noise-level table entries is larger - so the tie
    // returns the first abscissa, as the untied neighbour case does)
    const double slope = fast_div(y_hi - y_lo, fmax(x_hi - x_lo, 1e-300));
    return slope * (target - x_lo) + y_lo;
}

}  // namespace ttm
