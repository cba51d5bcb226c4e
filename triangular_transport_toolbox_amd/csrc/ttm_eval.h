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

#if defined(__HIPCC__)
#define TTM_HD __host__ __device__ __forceinline__
#else
#define TTM_HD inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// table entries are identical in every lane: move them to SGPRs so the
// interpreter's control flow is scalar (s_cbranch) instead of exec-masked
#define TTM_UNI(x) __builtin_amdgcn_readfirstlane(x)
#else
#define TTM_UNI(x) (x)
#endif

namespace ttm {

static constexpr double kSqrt2 = 1.4142135623730951;       // np.sqrt(2)
static constexpr double kSqrt2OverPi = 0.7978845608028654;  // np.sqrt(2/np.pi)
static constexpr double kSqrt2Pi = 2.5066282746310002;      // np.sqrt(2*np.pi)
static constexpr double kLn2 = 0.6931471805599453;          // np.log(2)

// ---------------------------------------------------------------------------
// program views
// ---------------------------------------------------------------------------

struct Prog {            // what is common to all components
    const double* qx;    // quadrature nodes / weights
    const double* qw;
    int Q;
    int family;
    int mono;
    int rect;
    double delta;
};

struct Comp {            // one component block (pointers into staged tables)
    const int* nm_terms;
    const int* mon_terms;
    const int* facs;
    const int* bfuns;
    const double* dpar;
    const double* cnm;   // coefficients of the nonmonotone / monotone terms
    const double* cmon;
    int kc, n_nm, n_mon, nB, nB_hf, nB_poly, nB_st, maxP_hf, maxP_poly, flags;
};

TTM_HD Comp make_comp(const int* cb, const double* dpar, const double* coef) {
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
    c.nm_terms = cb + TTM_UNI(cb[TTM_HDR_OFF_NM]);
    c.mon_terms = cb + TTM_UNI(cb[TTM_HDR_OFF_MON]);
    c.facs = cb + TTM_UNI(cb[TTM_HDR_OFF_FAC]);
    c.bfuns = cb + TTM_UNI(cb[TTM_HDR_OFF_B]);
    c.dpar = dpar;
    c.cnm = coef;
    c.cmon = coef + c.n_nm;
    return c;
}

// ---------------------------------------------------------------------------
// polynomial families: three-term recurrences with derivative propagation
//   P_{n+1} = (a x + b) P_n - c P_{n-1}
// ---------------------------------------------------------------------------

TTM_HD void poly_first(int fam, double x, double& p, double& dp) {
    switch (fam) {
        case TTM_FAM_HERMITE: p = 2.0 * x; dp = 2.0; break;
        case TTM_FAM_LAGUERRE: p = 1.0 - x; dp = -1.0; break;
        default: p = x; dp = 1.0; break;
    }
}

TTM_HD void poly_next(int fam, int n, double x, double& pm, double& p, double& dpm, double& dp) {
    double a, b, c;
    switch (fam) {
        case TTM_FAM_HERMITE_E: a = 1.0; b = 0.0; c = (double)n; break;
        case TTM_FAM_POWER: a = 1.0; b = 0.0; c = 0.0; break;
        case TTM_FAM_HERMITE: a = 2.0; b = 0.0; c = 2.0 * n; break;
        case TTM_FAM_CHEBYSHEV: a = 2.0; b = 0.0; c = 1.0; break;
        case TTM_FAM_LAGUERRE: { double r = 1.0 / (n + 1.0); a = -r; b = (2.0 * n + 1.0) * r; c = n * r; } break;
        default: /* LEGENDRE */ { double r = 1.0 / (n + 1.0); a = (2.0 * n + 1.0) * r; b = 0.0; c = n * r; } break;
    }
    const double lin = fma(a, x, b);
    const double pn = fma(lin, p, -c * pm);
    const double dpn = fma(a, p, fma(lin, dp, -c * dpm));
    pm = p; p = pn; dpm = dp; dp = dpn;
}

// P_order(x) and its derivative, order >= 1
TTM_HD void poly_eval(int fam, int order, double x, double& P, double& dP) {
    double pm = 1.0, dpm = 0.0, p, dp;
    poly_first(fam, x, p, dp);
    for (int n = 1; n < order; ++n) poly_next(fam, n, x, pm, p, dpm, dp);
    P = p; dP = dp;
}

// ---------------------------------------------------------------------------
// special terms (TM:917-1016), value and d/dx
// ---------------------------------------------------------------------------

template <bool VAL, bool DER>
TTM_HD void st_eval(int kind, double x, double mu, double sc, double& val, double& der) {
    const double d = x - mu;
    val = 0.0; der = 0.0;
    if (kind == TTM_KIND_LET || kind == TTM_KIND_RET) {
        const double t = d / (kSqrt2 * sc);
        const double e = erf(t);
        const double sgn = (kind == TTM_KIND_LET) ? -1.0 : 1.0;
        if (VAL) {
            const double g = exp(-(t * t));
            val = (d * (1.0 + sgn * e) + sgn * (sc * kSqrt2OverPi * g)) / 2.0;
        }
        if (DER) der = (1.0 + sgn * e) / 2.0;
    } else if (kind == TTM_KIND_RBF) {
        const double u = d / sc;
        const double g = exp(-(u * u) / 2.0);
        if (VAL) val = g / (sc * kSqrt2Pi);
        if (DER) der = -d / (kSqrt2Pi * (sc * sc * sc)) * g;
    } else {  // TTM_KIND_IRBF
        if (VAL) val = (1.0 + erf(d / (kSqrt2 * sc))) / 2.0;
        if (DER) der = 1.0 / (kSqrt2Pi * sc) * exp(-(d * d) / (2.0 * (sc * sc)));
    }
}

// ---------------------------------------------------------------------------
// rectifiers
// ---------------------------------------------------------------------------

TTM_HD double rect_eval(int mode, double g) {
    switch (mode) {
        case TTM_RECT_EXPONENTIAL: return exp(g);
        case TTM_RECT_SOFTPLUS: { const double ag = kLn2 * g; return log(1.0 + exp(-fabs(ag))) + (ag < 0.0 ? 0.0 : ag); }
        case TTM_RECT_SQUARED: return g * g;
        case TTM_RECT_EXPNEG: return exp(-g);
        default: return g < 0.0 ? exp(g) : g + 1.0;   // ELU, TM:5012-5016
    }
}

// r(g), the factor of evaluate_dfdc (TM:5112-5165) and log(r + delta) as logevaluate (TM:5167-5213)
TTM_HD void rect_all(int mode, double delta, double g, double& r, double& dr, double& logr) {
    switch (mode) {
        case TTM_RECT_EXPONENTIAL:
            r = exp(g); dr = r; logr = (delta == 0.0) ? g : log(r + delta); break;
        case TTM_RECT_SOFTPLUS:
            r = rect_eval(mode, g); dr = 1.0 / (1.0 + exp(-kLn2 * g)); logr = log(r + delta); break;
        case TTM_RECT_EXPNEG:
            r = exp(-g); dr = -r; logr = -g; break;
        case TTM_RECT_SQUARED:
            r = g * g; dr = NAN; logr = log(r); break;          // dfdc "not implemented" in the reference
        default:
            r = rect_eval(mode, g); dr = NAN; logr = log(r); break;
    }
}

// ---------------------------------------------------------------------------
// A-part of a term: product of its factors on columns other than kc
// (all 'HF' factors of one term share a single exp(-sum x^2 / 4))
// ---------------------------------------------------------------------------

template <class XA>
TTM_HD double eval_A(const int* term, const Comp& c, int fam, const XA& x) {
    const int f0 = TTM_UNI(term[0]);
    const int nf = TTM_UNI(term[1]);
    double prod = 1.0, ssq = 0.0;
    bool hf = false;
    for (int f = 0; f < nf; ++f) {
        const int* F = c.facs + 4 * (f0 + f);
        const int var = TTM_UNI(F[0]);
        const int kind = TTM_UNI(F[1]);
        const int order = TTM_UNI(F[2]);
        const int p0 = TTM_UNI(F[3]);
        const double xv = x(var);
        if (kind == TTM_KIND_POLY || kind == TTM_KIND_HF) {
            double P, dP;
            poly_eval(fam, order, xv, P, dP);
            if (kind == TTM_KIND_HF) {
                prod *= c.dpar[p0] * P;
                ssq = fma(xv, xv, ssq);
                hf = true;
            } else {
                prod *= P;
            }
        } else {
            double v, dv;
            st_eval<true, false>(kind, xv, c.dpar[p0], c.dpar[p0 + 1], v, dv);
            prod *= v;
        }
    }
    if (hf) prod *= exp(-0.25 * ssq);
    return prod;
}

// sum_i c_i A_i over the nonmonotone terms
template <class XA>
TTM_HD double nonmon_sum(const Comp& c, int fam, const XA& x) {
    double s = 0.0;
    for (int i = 0; i < c.n_nm; ++i) {
        const int* T = c.nm_terms + 4 * i;
        s = fma(c.cnm[TTM_UNI(T[3])], eval_A(T, c, fam, x), s);
    }
    return s;
}

// w[b] = sum over monotone terms with x_k-function b of c_i A_i ; slot nB collects b == -1
template <class XA, class Slots>
TTM_HD void mon_weights(const Comp& c, int fam, const XA& x, Slots& w) {
    for (int b = 0; b <= c.nB; ++b) w.set(b, 0.0);
    for (int i = 0; i < c.n_mon; ++i) {
        const int* T = c.mon_terms + 4 * i;
        const int nf = TTM_UNI(T[1]);
        int b = TTM_UNI(T[2]);
        if (b < 0) b = c.nB;
        const double ci = c.cmon[TTM_UNI(T[3])];
        const double a = (nf == 0) ? ci : ci * eval_A(T, c, fam, x);
        w.set(b, w.get(b) + a);
    }
}

// visit every distinct x_k-univariate function: f(b, B_b(t), B_b'(t))
template <bool DER, class F>
TTM_HD void for_each_B(const Comp& c, int fam, double t, F&& f) {
    int b = 0;
    if (c.nB_hf > 0) {
        const double E = exp(-0.25 * (t * t));
        double pm = 1.0, dpm = 0.0, p, dp;
        poly_first(fam, t, p, dp);
        for (int n = 1; n <= c.maxP_hf; ++n) {
            const int* B = c.bfuns + 4 * b;
            if (TTM_UNI(B[1]) == n) {
                const double a = c.dpar[TTM_UNI(B[2])];
                // d/dt [a P e^{-t^2/4}] = -1/2 e^{-t^2/4} (t aP - 2 aP')   (TM:1245)
                f(b, a * p * E, DER ? -0.5 * E * (t * (a * p) - 2.0 * (a * dp)) : 0.0);
                ++b;
            }
            if (n < c.maxP_hf) poly_next(fam, n, t, pm, p, dpm, dp);
        }
    }
    if (c.nB_poly > 0) {
        double pm = 1.0, dpm = 0.0, p, dp;
        poly_first(fam, t, p, dp);
        for (int n = 1; n <= c.maxP_poly; ++n) {
            const int* B = c.bfuns + 4 * b;
            if (TTM_UNI(B[1]) == n) {
                f(b, p, dp);
                ++b;
            }
            if (n < c.maxP_poly) poly_next(fam, n, t, pm, p, dpm, dp);
        }
    }
    for (int s = 0; s < c.nB_st; ++s, ++b) {
        const int* B = c.bfuns + 4 * b;
        const int p0 = TTM_UNI(B[2]);
        double v, dv;
        st_eval<true, DER>(TTM_UNI(B[0]), t, c.dpar[p0], c.dpar[p0 + 1], v, dv);
        f(b, v, dv);
    }
}

// g(t) = w[nB] + sum_b w[b] B_b(t)   (the argument of the rectifier, or the
// monotone part itself for separable maps) and dg/dt
template <bool DER, class Slots>
TTM_HD void g_eval(const Comp& c, int fam, double t, const Slots& w, double& g, double& dg) {
    double acc = w.get(c.nB), dacc = 0.0;
    for_each_B<DER>(c, fam, t, [&](int b, double v, double dv) {
        const double wb = w.get(b);
        acc = fma(wb, v, acc);
        if (DER) dacc = fma(wb, dv, dacc);
    });
    g = acc; dg = dacc;
}

// int_0^{xk} (r(g(t)) + delta) dt with the reference's node order and grouping (TM:4238-4258)
template <class Slots>
TTM_HD double integrate_rect(const Comp& c, const Prog& p, double xk, const Slots& w) {
    const double half = xk * 0.5;
    double res = 0.0;
    for (int q = 0; q < p.Q; ++q) {
        const double t = half * p.qx[q] + half;
        double g, dg;
        g_eval<false>(c, p.family, t, w, g, dg);
        const double fr = rect_eval(p.rect, g) + p.delta;
        const double term = half * (p.qw[q] * fr);
        res = (q == 0) ? term : res + term;
    }
    return res;
}

// monotone part of S_k at x_k = t given the sample's weights: value and dS/dx_k
template <bool DER, class Slots>
TTM_HD void mon_eval(const Comp& c, const Prog& p, double t, const Slots& w, double& m, double& dm) {
    if (p.mono == TTM_MONO_SEPARABLE) {
        g_eval<DER>(c, p.family, t, w, m, dm);
    } else {
        m = integrate_rect(c, p, t, w);
        dm = 0.0;
        if (DER) {
            double g, dg;
            g_eval<false>(c, p.family, t, w, g, dg);
            dm = rect_eval(p.rect, g) + p.delta;
        }
    }
}

// ---------------------------------------------------------------------------
// per-sample bodies
// ---------------------------------------------------------------------------

// S_k(x) and dS_k/dx_k.  scratch: nB+1 slots.
template <bool DER, class XA, class Slots>
TTM_HD void sample_forward(const Comp& c, const Prog& p, const XA& x, Slots& w, bool want_value, double& S, double& dS) {
    mon_weights(c, p.family, x, w);
    double m, dm;
    mon_eval<DER>(c, p, x(c.kc), w, m, dm);
    S = want_value ? nonmon_sum(c, p.family, x) + m : m;
    dS = dm;
}

// basis rows (inspection): which 0 Psi_nonmon, 1 Psi_mon, 2 dPsi_mon/dx_k ; out(i) = value
template <class XA, class Out>
TTM_HD void sample_basis(const Comp& c, const Prog& p, int which, const XA& x, Out&& out) {
    if (which == 0) {
        for (int i = 0; i < c.n_nm; ++i) {
            const int* T = c.nm_terms + 4 * i;
            out(TTM_UNI(T[3]), eval_A(T, c, p.family, x));
        }
        return;
    }
    const double xk = x(c.kc);
    for (int i = 0; i < c.n_mon; ++i) {
        const int* T = c.mon_terms + 4 * i;
        const int bsel = TTM_UNI(T[2]);
        double v = 1.0, dv = 0.0;
        if (bsel >= 0) {
            for_each_B<true>(c, p.family, xk, [&](int b, double bv, double bdv) {
                if (b == bsel) { v = bv; dv = bdv; }
            });
        }
        const double a = eval_A(T, c, p.family, x);
        out(TTM_UNI(T[3]), a * (which == 1 ? v : dv));
    }
}

// Objective + gradient contribution of one sample, integrated rectifier
// (TM:3343-3376, 3475-3569).  acc layout: [0] J, [1..n_nm] d/dc_nonmon, then d/dc_mon.
// scratch slots: w (nB+1) | Bv (nB+1) | I (nB+1)
template <class XA, class Slots, class Acc>
TTM_HD void sample_objective_int(const Comp& c, const Prog& p, const XA& x, Slots& w, Slots& Bv, Slots& I, Acc& acc) {
    mon_weights(c, p.family, x, w);
    const double xk = x(c.kc);
    const double half = xk * 0.5;
    double mono = 0.0;
    for (int b = 0; b <= c.nB; ++b) I.set(b, 0.0);
    for (int q = 0; q < p.Q; ++q) {
        const double t = half * p.qx[q] + half;
        double g = w.get(c.nB);
        for_each_B<false>(c, p.family, t, [&](int b, double v, double) {
            g = fma(w.get(b), v, g);
            Bv.set(b, v);
        });
        double r, dr, logr;
        rect_all(p.rect, p.delta, g, r, dr, logr);
        const double term = half * (p.qw[q] * (r + p.delta));
        mono = (q == 0) ? term : mono + term;
        const double cq = (half * p.qw[q]) * dr;       // lim_dif*0.5*W_q * r'(g_q)   (TM:4264-4278, 5127-5133)
        for (int b = 0; b < c.nB; ++b) I.set(b, fma(cq, Bv.get(b), I.get(b)));
        I.set(c.nB, I.get(c.nB) + cq);
    }
    // nonmonotone part and its gradient
    double off = 0.0;
    for (int i = 0; i < c.n_nm; ++i) {
        const int* T = c.nm_terms + 4 * i;
        off = fma(c.cnm[TTM_UNI(T[3])], eval_A(T, c, p.family, x), off);
    }
    const double S = off + mono;
    // values at x_k for the log term
    double g = w.get(c.nB);
    for_each_B<false>(c, p.family, xk, [&](int b, double v, double) {
        g = fma(w.get(b), v, g);
        Bv.set(b, v);
    });
    Bv.set(c.nB, 1.0);
    double r, dr, logr;
    rect_all(p.rect, p.delta, g, r, dr, logr);
    acc.add(0, 0.5 * S * S - logr);
    for (int i = 0; i < c.n_nm; ++i) {
        const int* T = c.nm_terms + 4 * i;
        acc.add(1 + TTM_UNI(T[3]), S * eval_A(T, c, p.family, x));
    }
    const double rinv = dr / (r + p.delta);
    for (int i = 0; i < c.n_mon; ++i) {
        const int* T = c.mon_terms + 4 * i;
        const int nf = TTM_UNI(T[1]);
        int b = TTM_UNI(T[2]);
        if (b < 0) b = c.nB;
        const double a = (nf == 0) ? 1.0 : eval_A(T, c, p.family, x);
        acc.add(1 + c.n_nm + TTM_UNI(T[3]), a * (S * I.get(b) - rinv * Bv.get(b)));
    }
}

// Separable objective pieces of one sample (TM:2990-3006):
// acc[0] += log dS, acc[1+i] += dPsi_i / dS with dS = dPsi.c + delta * rowsum(dPsi)
// scratch: dB (nB+1)
template <class XA, class Slots, class Acc>
TTM_HD void sample_objective_sep(const Comp& c, const Prog& p, const XA& x, Slots& dB, Acc& acc) {
    const double xk = x(c.kc);
    for_each_B<true>(c, p.family, xk, [&](int b, double, double dv) { dB.set(b, dv); });
    dB.set(c.nB, 0.0);
    double dS = 0.0, rowsum = 0.0;
    for (int i = 0; i < c.n_mon; ++i) {
        const int* T = c.mon_terms + 4 * i;
        const int nf = TTM_UNI(T[1]);
        int b = TTM_UNI(T[2]);
        if (b < 0) b = c.nB;
        const double a = (nf == 0) ? 1.0 : eval_A(T, c, p.family, x);
        const double d = a * dB.get(b);
        dS = fma(c.cmon[TTM_UNI(T[3])], d, dS);
        rowsum += d;
    }
    dS += rowsum * p.delta;
    acc.add(0, log(dS));
    const double inv = 1.0 / dS;
    for (int i = 0; i < c.n_mon; ++i) {
        const int* T = c.mon_terms + 4 * i;
        const int nf = TTM_UNI(T[1]);
        int b = TTM_UNI(T[2]);
        if (b < 0) b = c.nB;
        const double a = (nf == 0) ? 1.0 : eval_A(T, c, p.family, x);
        acc.add(1 + TTM_UNI(T[3]), a * dB.get(b) * inv);
    }
}

// Bisection root search of one sample for one component (TM:3842-3976).
// Returns the last trial point (what the reference leaves in X[:, kc]) and the
// number of midpoint iterations it needed.  `cap` < 0: no cap.
template <class Slots>
TTM_HD double sample_bisect(const Comp& c, const Prog& p, double off, double zk, const Slots& w, int cap, int& iters) {
    double lo = -2.0, hi = 2.0, m, dm;
    mon_eval<false>(c, p, lo, w, m, dm);
    double flo = (off + m) - zk;
    mon_eval<false>(c, p, hi, w, m, dm);
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
            mon_eval<false>(c, p, lo, w, m, dm);
            flo = (off + m) - zk;
        } else if (flo < 0.0) {
            lo = hi; hi = hi + diff * 2.0;
            flo = fhi;
            last = hi;
            mon_eval<false>(c, p, hi, w, m, dm);
            fhi = (off + m) - zk;
        } else {
            break;
        }
    }
    iters = 0;
    const int maxit = (cap >= 0 && cap < 100) ? cap : 100;
    while (iters < maxit) {
        ++iters;
        const double mid = (lo + hi) / 2.0;     // np.mean over two values
        last = mid;
        mon_eval<false>(c, p, mid, w, m, dm);
        const double fm = (off + m) - zk;
        if (fm < 0.0) lo = mid;
        if (fm > 0.0) hi = mid;
        if (!(fabs(fm) > 1e-9)) break;
    }
    return last;
}

// interp1d lookup (TM:4062-4082): xs non-decreasing table of map outputs, ys the abscissae
TTM_HD double table_lookup(const double* xs, const double* ys, int T, double target) {
    // np.searchsorted(xs, target) (left): first i with xs[i] >= target; NaN sorts last
    int lo = 0, hi = T;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (xs[mid] < target) lo = mid + 1; else hi = mid;
    }
    int i = lo < 1 ? 1 : (lo > T - 1 ? T - 1 : lo);
    const double x_lo = xs[i - 1], x_hi = xs[i], y_lo = ys[i - 1], y_hi = ys[i];
    const double slope = (y_hi - y_lo) / (x_hi - x_lo);
    return slope * (target - x_lo) + y_lo;
}

}  // namespace ttm
