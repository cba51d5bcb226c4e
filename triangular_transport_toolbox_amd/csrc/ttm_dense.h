// ttm_dense.h - the integrated-rectifier evaluator of components with a DENSE B set in monomial form.
//
// A component of an integrated-rectifier map (TM:2499-2547) is
//     S_k(x) = Psi_nonmon(x_<k) c_nonmon + int_0^{x_k} ( r(g(t)) + delta ) dt,      g(t) = sum_b w_b(x_<k) B_b(t),
// with the integral taken by Gauss-Legendre quadrature (TM:4238-4258: Q nodes, 25 in the examples).  For a polynomial map
// the B_b are Hermite-function orders <= Ph and / or plain polynomial orders <= Pp of x_k (any set of orders <= TTM_I_PMAX = 10 - the
// "dense B set" of ttm_eval.h is the case 1..P -, no special terms), so
//     g(t) = E(t) H(t) + A(t),        E(t) = exp(-t^2/4),   H, A polynomials of degree Ph, Pp.
// Everything a sample spends on a component is spent in the Q nodes - in the bisection (TM:3842-3976) Q nodes for each
// of ~33 trial points - so the node is what this file makes cheap:
//   * H and A are converted ONCE per sample and component from the family's orthogonal basis to MONOMIAL coefficients
//     (TTM_MONO_TABLE, the same NumPy conversion the U-form uses): a node evaluates them by Horner's rule, one FMA per
//     order instead of the three instructions per order of the three-term recurrence, and without the family switch;
//   * exp: Cody-Waite reduction + a degree-11 near-minimax polynomial (TTM_EXP11: 1.6e-17 relative in exact
//     arithmetic; the Taylor polynomial of fast_exp needs degree 13 for the same), 17 instructions; E(t) takes t^2
//     directly - the factor -1/4 is folded into a second coefficient set (exact: powers of two), 18 instructions from t;
//   * the node sum is sum_q W_q r(g_q) accumulated by FMA, scaled by x_k/2 once, delta added once as
//     delta (x_k/2) sum_q W_q - the reference adds x_k/2 W_q (r_q + delta) node by node; the two differ by rounding only
//     (a few 1e-16 relative; the tolerances of the integrated path are 1e-11, tests/test_transport_map.py);
//   * NODES quadrature nodes are evaluated together as a short vector: independent FMA chains, and the scalar instructions
//     of a node (loop control, loads of the node constants) are paid once per group.
// A node costs 40 + Ph vector instructions with the exponential rectifier (the generic evaluator of ttm_eval.h: ~100).
//
// The bodies compile for the host as well (tests/hostemu drives them through the C ABI's test double).
#pragma once

#include "ttm_dense_table.h"
#include "ttm_eval.h"

namespace ttm {

#ifndef TTM_I_NODES
#define TTM_I_NODES 5
#endif

#if defined(__HIPCC__)
__device__ double g_mono_table[6 * (TTM_I_PMAX + 1) * (TTM_I_PMAX + 1)] = { TTM_MONO_TABLE_VALUES };   // (not const: scalar loads, see g_exp_coef)
__device__ double g_exp11[10] = { TTM_EXP11_VALUES };
__device__ double g_expq11[11] = { TTM_EXPQ11_VALUES };
#endif
static const double h_mono_table[6 * (TTM_I_PMAX + 1) * (TTM_I_PMAX + 1)] = { TTM_MONO_TABLE_VALUES };
static const double h_exp11[10] = { TTM_EXP11_VALUES };
static const double h_expq11[11] = { TTM_EXPQ11_VALUES };

TTM_HD cdbl_p mono_table_of(int family) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (cdbl_p)g_mono_table + family * (TTM_I_PMAX + 1) * (TTM_I_PMAX + 1);
#else
    return h_mono_table + family * (TTM_I_PMAX + 1) * (TTM_I_PMAX + 1);
#endif
}
TTM_HD cdbl_p exp11_coefs() {
#if defined(__HIP_DEVICE_COMPILE__)
    return (cdbl_p)g_exp11;
#else
    return h_exp11;
#endif
}
TTM_HD cdbl_p expq11_coefs() {
#if defined(__HIP_DEVICE_COMPILE__)
    return (cdbl_p)g_expq11;
#else
    return h_expq11;
#endif
}

// exp(y) for |y| <= 800 (no guards): 17 instructions
template <class R>
TTM_HD R dense_exp_core(const R& y) {
    cdbl_p c = exp11_coefs();
    const R k = vrint(y * 1.4426950408889634);
    R r = vfma(-k, 6.93147180369123816490e-01, y);
    r = vfma(-k, 1.90821492927058770002e-10, r);
    R p(c[9]);
#pragma unroll
    for (int j = 8; j >= 0; --j) p = vfma(p, r, c[j]);
    p = vfma(p, r, 1.0);
    p = vfma(p, r, 1.0);
    return vldexp(p, vtoint(k));
}

// exp(y), any y: clamped to +-800 (0 / inf beyond, as exp); a NaN argument is NOT restored (the callers add a probe term)
template <class R>
TTM_HD R dense_exp(const R& y) { return dense_exp_core(vmin(vmax(y, -800.0), 800.0)); }

// exp(-t^2/4) from t: u = t^2, k = rint(-u log2(e)/4) <= 0, s = u + k 4 ln2 (|s| <= 2 ln2), exp(-s/4) by the same polynomial
// with the coefficients scaled by (-1/4)^j (exact).  u = inf gives NaN (k = -inf, s = NaN) - the value multiplies a
// polynomial of the same t, which is infinite there; t = NaN gives NaN.
template <class R>
TTM_HD R dense_expq(const R& t) {
    cdbl_p c = expq11_coefs();                                          // c[j-1] = c_j (-1/4)^j, j = 1 .. 11
    const R u = t * t;
    const R k = vrint(u * -0.36067376022224085);                       // -log2(e) / 4
    R s = vfma(k, 2.77258872147649526596e+00, u);                      // 4 ln2 hi / lo (Cody-Waite halves of fast_exp, x 4: exact)
    s = vfma(k, 7.63285971708235080008e-10, s);
    R p(c[10]);
#pragma unroll
    for (int j = 9; j >= 0; --j) p = vfma(p, s, c[j]);
    p = vfma(p, s, 1.0);
    return vldexp(p, vtoint(k));
}

// monomial form of g for one sample: g(t) = E(t) sum_j h[j] t^j + sum_j a[j] t^j
template <int PH, int PP>
struct DenseMono {
    double h[PH + 1];
    double a[PP + 1];
    double probe;               // 0, or NaN when x_k or a coefficient is NaN / infinite (added to every result: the clamps of
                                // the node's exp would otherwise turn a NaN sample into a finite number)
};

// from the weight slots of dense_weights: w = [a_n wh (Hermite functions, in bfun order) | wp (polynomials) | w_none]; the
// order of a B function comes from its record (any subset of the orders 1..P: a missing order has no slot)
template <int PH, int PP, class W>
TTM_HD void dense_monomials(const Comp& c, const Prog& p, const W& w, DenseMono<PH, PP>& d) {
    cdbl_p M = mono_table_of(p.family);
    constexpr int LD = TTM_I_PMAX + 1;
#pragma unroll
    for (int j = 0; j <= PH; ++j) d.h[j] = 0.0;
#pragma unroll
    for (int j = 0; j <= PP; ++j) d.a[j] = 0.0;
    d.a[0] = w.get(c.nB);
    for (int b = 0; b < c.nB_hf; ++b) {
        cdbl_p row = M + TTM_UNI(c.bfuns[4 * b + 1]) * LD;
        const double wn = w.get(b);
#pragma unroll
        for (int j = 0; j <= PH; ++j) d.h[j] = fma(wn, row[j], d.h[j]);
    }
    for (int b = c.nB_hf; b < c.nB_hf + c.nB_poly; ++b) {
        cdbl_p row = M + TTM_UNI(c.bfuns[4 * b + 1]) * LD;
        const double wn = w.get(b);
#pragma unroll
        for (int j = 0; j <= PP; ++j) d.a[j] = fma(wn, row[j], d.a[j]);
    }
    double s = 0.0;
#pragma unroll
    for (int j = 0; j <= PH; ++j) s += d.h[j];
#pragma unroll
    for (int j = 0; j <= PP; ++j) s += d.a[j];
    d.probe = s * 0.0;
}

// g at the points t (R = double or VecD<N>); E = exp(-t^2/4) is returned for the callers that need it again
template <int PH, int PP, class R>
TTM_HD R dense_g(const DenseMono<PH, PP>& d, const R& t, R& E) {
    R A(d.a[PP]);
#pragma unroll
    for (int j = PP - 1; j >= 0; --j) A = vfma(A, t, R(d.a[j]));
    if (PH == 0) { E = R(1.0); return A; }
    R H(d.h[PH]);
#pragma unroll
    for (int j = PH - 1; j >= 0; --j) H = vfma(H, t, R(d.h[j]));
    E = dense_expq(t);
    return vfma(H, E, A);
}

// r(g) for the node loop.  RECT >= 0 fixes the rectifier at compile time.
template <int RECT, class R>
TTM_HD R dense_rect(int rect_rt, const R& g) {
    const int rect = (RECT >= 0) ? RECT : rect_rt;
    if (rect == TTM_RECT_EXPONENTIAL) return dense_exp(g);
    return rect_eval(rect, g);
}

// int_0^{xk} (r(g(t)) + delta) dt by the map's Gauss-Legendre rule (TM:4238-4258); qw_sum = sum_q W_q
template <int PH, int PP, int RECT>
TTM_HD double dense_integral(const Prog& p, double qw_sum, const DenseMono<PH, PP>& d, double xk) {
    typedef VecD<TTM_I_NODES> V;
    const double half = xk * 0.5;
    double acc = 0.0;
    int q = 0;
    for (; q + TTM_I_NODES <= p.Q; q += TTM_I_NODES) {
        V t, E;
#pragma unroll
        for (int e = 0; e < TTM_I_NODES; ++e) t[e] = fma(half, p.qx[q + e], half);
        const V g = dense_g(d, t, E);
        const V r = dense_rect<RECT>(p.rect, g);
#pragma unroll
        for (int e = 0; e < TTM_I_NODES; ++e) acc = fma(p.qw[q + e], r[e], acc);
    }
    for (; q < p.Q; ++q) {
        double E;
        const double t = fma(half, p.qx[q], half);
        const double g = dense_g(d, t, E);
        acc = fma(p.qw[q], dense_rect<RECT>(p.rect, g), acc);
    }
    return half * fma(p.delta, qw_sum, acc) + (d.probe + xk * 0.0);
}

// sum of the quadrature weights, in node order (uniform)
TTM_HD double dense_qw_sum(const Prog& p) {
    double s = 0.0;
    for (int q = 0; q < p.Q; ++q) s += p.qw[q];
    return s;
}

// the weight source the root searches of ttm_eval.h evaluate through (sample_bisect / sample_newton call mon_eval)
template <int PH, int PP, int RECT>
struct DenseMonoSet {
    DenseMono<PH, PP> d;
    double qw_sum;
};
template <int MONO, bool DER, int PH, int PP, int RECT>
TTM_HD void mon_eval(const Comp&, const Prog& p, const double& t, const DenseMonoSet<PH, PP, RECT>& s, double& m, double& dm) {
    m = dense_integral<PH, PP, RECT>(p, s.qw_sum, s.d, t);
    dm = 0.0;
    if (DER) {
        double E;
        const double g = dense_g(s.d, t, E);
        dm = dense_rect<RECT>(p.rect, g) + p.delta;
    }
}

// ---------------------------------------------------------------------------
// per-sample bodies (w: nB + 1 scratch slots)
// ---------------------------------------------------------------------------

// S_k(x) and dS_k/dx_k (TM:2499-2547)
template <int PH, int PP, int RECT, bool DER, class XA, class Slots>
TTM_HD void dense_sample_forward(const Comp& c, const Prog& p, double qw_sum, VarCache<XA, double>& x, Slots& w, bool want_value,
                                 double& S, double& dS) {
#ifndef INT_X_NOW                                           /* (INT_X_*: timing experiments, results wrong by construction) */
    dense_weights<double>(c, p, x, w);
#endif
    DenseMonoSet<PH, PP, RECT> s;
#ifndef INT_X_NOMONO
    dense_monomials<PH, PP>(c, p, w, s.d);
#else
    for (int j = 0; j <= PH; ++j) s.d.h[j] = 0.01 * (j + 1);
    for (int j = 0; j <= PP; ++j) s.d.a[j] = 0.02 * (j + 1);
    s.d.probe = 0.0;
#endif
    s.qw_sum = qw_sum;
    // (the nonmonotone part first: behind the node loop it would keep the component's table pointers alive across it)
    const double xk = x.get(c.kc);
#ifndef INT_X_NONM
    const double nm = want_value ? nonmon_sum<double>(c, p, x) : 0.0;
#else
    const double nm = xk;
#endif
    double m, dm;
    mon_eval<TTM_MONO_INTEGRATED, DER>(c, p, xk, s, m, dm);
    S = want_value ? nm + m : m;
    dS = dm;
}

// bisection (the reference's sequence, TM:3842-3976) or safeguarded Newton root search of one sample
template <int PH, int PP, int RECT, bool NEWTON, class XA, class Slots>
TTM_HD double dense_sample_root(const Comp& c, const Prog& p, double qw_sum, VarCache<XA, double>& x, Slots& w, double off, double zk,
                                int cap, int& it) {
    dense_weights<double>(c, p, x, w);
    DenseMonoSet<PH, PP, RECT> s;
    dense_monomials<PH, PP>(c, p, w, s.d);
    s.qw_sum = qw_sum;
    return NEWTON ? sample_newton<TTM_MONO_INTEGRATED>(c, p, off, zk, s, it) : sample_bisect<TTM_MONO_INTEGRATED>(c, p, off, zk, s, cap, it);
}

// The node loop of the objective: sum_q W_q r(g_q) (-> mono) and the MONOMIAL moments of W_q r'(g_q) over the nodes,
//     mh[j] = sum_q W_q r'(g_q) E(t_q) t_q^j  (j <= PH),      mp[j] = sum_q W_q r'(g_q) t_q^j  (j <= PP),
// from which every integral int r'(g) B_b dt follows by the basis-conversion row of B_b (one multiplication for the power
// and one addition per order and node); NODES nodes per pass.  The factor x_k/2 of the rule follows once, outside.
template <int PH, int PP, int RECT, int NODES = TTM_I_NODES>
TTM_HD void dense_moment_nodes(const Prog& p, const DenseMono<PH, PP>& d, double xk, double& mono, double* mh, double* mp) {
    typedef VecD<NODES> V;
    const int rect = (RECT >= 0) ? RECT : p.rect;
    const double half = xk * 0.5;
    mono = 0.0;
#pragma unroll
    for (int j = 0; j <= PH; ++j) mh[j] = 0.0;
#pragma unroll
    for (int j = 0; j <= PP; ++j) mp[j] = 0.0;
    auto nodes = [&](auto tag, int q) {
        typedef decltype(tag) T;                                // V or double
        constexpr int L = lanes_of<T>::value;
        T t, E;
#pragma unroll
        for (int e = 0; e < L; ++e) set_elem(t, e, fma(half, p.qx[q + e], half));
        const T g = dense_g(d, t, E);
        T r, dr, logr;
        if (rect == TTM_RECT_EXPONENTIAL) { r = dense_exp(g); dr = r; }
        else rect_all(rect, 0.0, g, r, dr, logr);               // (logr unused here)
        T cq;
#pragma unroll
        for (int e = 0; e < L; ++e) {
            mono = fma(p.qw[q + e], elem(r, e), mono);
            set_elem(cq, e, p.qw[q + e] * elem(dr, e));          // W_q r'(g_q); the factor x_k/2 follows once (TM:4264-4278, 5127-5133)
        }
        if (PH > 0) {
            T pw = cq * E;
#pragma unroll
            for (int j = 0; j <= PH; ++j) {
#pragma unroll
                for (int e = 0; e < L; ++e) mh[j] += elem(pw, e);
                if (j < PH) pw = pw * t;
            }
        }
        {
            T pw = cq;
#pragma unroll
            for (int j = 0; j <= PP; ++j) {
#pragma unroll
                for (int e = 0; e < L; ++e) mp[j] += elem(pw, e);
                if (j < PP) pw = pw * t;
            }
        }
    };
    int q = 0;
    for (; q + NODES <= p.Q; q += NODES) nodes(V(0.0), q);
    for (; q < p.Q; ++q) nodes(0.0, q);
}

// objective + gradient contribution of one sample (TM:3343-3376, 3475-3569); scratch as sample_objective_int_dense:
// w (nB+1; reused for the B values at x_k) | I (nB+1).  The integrals int r'(g) B_b dt are taken as MONOMIAL moments
// sum_q cq E_q t_q^j (dense_moment_nodes) and converted to the basis once.  (Components without an X program: the gradient
// walks the fold recipe - csrc/ttm_xprog.h is the path of the others.)
template <int PH, int PP, int RECT, class XA, class Slots, class Acc>
TTM_HD void dense_sample_objective(const Comp& c, const Prog& p, double qw_sum, VarCache<XA, double>& x, Slots& w, Slots& Bv, Slots& I, Acc& acc) {
    const int rect = (RECT >= 0) ? RECT : p.rect;
    dense_weights<double>(c, p, x, w);
    DenseMono<PH, PP> d;
    dense_monomials<PH, PP>(c, p, w, d);
    const double xk = x.get(c.kc);
    const double half = xk * 0.5;
    double mono;
    double mh[PH + 1], mp[PP + 1];
    dense_moment_nodes<PH, PP, RECT>(p, d, xk, mono, mh, mp);
    mono = half * fma(p.delta, qw_sum, mono) + (d.probe + xk * 0.0);
    // basis integrals from the moments (the Hermite-function ones carry their normalisation constants, as in
    // sample_objective_int_dense)
    cdbl_p M = mono_table_of(p.family);
    constexpr int LD = TTM_I_PMAX + 1;
    for (int b = 0; b < c.nB_hf; ++b) {
        cdbl_p row = M + TTM_UNI(c.bfuns[4 * b + 1]) * LD;
        double s = 0.0;
#pragma unroll
        for (int j = 0; j <= PH; ++j) s = fma(row[j], mh[j], s);
        I.set(b, (c.dpar[TTM_UNI(c.bfuns[4 * b + 2])] * half) * s);
    }
    for (int b = c.nB_hf; b < c.nB_hf + c.nB_poly; ++b) {
        cdbl_p row = M + TTM_UNI(c.bfuns[4 * b + 1]) * LD;
        double s = 0.0;
#pragma unroll
        for (int j = 0; j <= PP; ++j) s = fma(row[j], mp[j], s);
        I.set(b, half * s);
    }
    I.set(c.nB, half * mp[0]);
    const double S = nonmon_sum<double>(c, p, x) + mono;
    // values at x_k for the log term (the weights are dead: Bv may be the very columns of w)
    double E;
    const double g = dense_g(d, xk, E) + (d.probe + xk * 0.0);
    for_each_B<false>(c, p, xk, [&](int b, double v, double) { Bv.set(b, v); });
    Bv.set(c.nB, 1.0);
    double r, dr, logr;
    rect_all(rect, p.delta, g, r, dr, logr);
    acc.add(0, 0.5 * S * S - logr);
    const double rinv = dr * fast_rcp(r + p.delta);
    objective_gradient(c, p, x, S, rinv, Bv, I, acc);
}

// order class of a component range for the dense kernels: 0 = not applicable; else the template pair (PH, PP) to run
struct DenseClass { int ph, pp; };
TTM_HD DenseClass dense_class_of(int max_ph, int max_pp) {
    DenseClass k{0, 0};
    if (max_ph > TTM_I_PMAX || max_pp > TTM_I_PMAX || (max_ph == 0 && max_pp == 0)) return k;
    if (max_pp == 0) { k.ph = max_ph <= 3 ? 3 : (max_ph <= 5 ? 5 : (max_ph <= 8 ? 8 : 10)); k.pp = 0; }
    else if (max_ph == 0) { k.ph = 0; k.pp = 10; }
    else { k.ph = 10; k.pp = 10; }
    return k;
}

// largest orders of a component range from the host flag words (ttm_program::h_complex); false when a component of the
// range has other functions of x_k than polynomials / Hermite functions
inline bool dense_range_class(const int32_t* h_complex, int k0, int k1, DenseClass& cls) {
    int ph = 0, pp = 0;
    for (int k = k0; k < k1; ++k) {
        const int f = h_complex[k];
        if (!(f & 4)) return false;
        const int a = (f >> 8) & 15, b = (f >> 12) & 15;
        ph = a > ph ? a : ph;
        pp = b > pp ? b : pp;
    }
    cls = dense_class_of(ph, pp);
    return cls.ph > 0 || cls.pp > 0;
}

// run CALL(PH, PP, RECT) for the class and rectifier given at run time (host-side dispatch: kernel pick, test double)
#define TTM_DENSE_DISPATCH_RECT(CALL, PH, PP, rect) \
    do { if ((rect) == TTM_RECT_EXPONENTIAL) { CALL(PH, PP, TTM_RECT_EXPONENTIAL); } else { CALL(PH, PP, -1); } } while (0)
#ifdef TTM_INT_MINI      /* (tuning builds: two classes, exponential rectifier only - a sixth of the compile time) */
#define TTM_DENSE_DISPATCH(CALL, cls, rect) \
    do { if ((cls).ph == 3) { CALL(3, 0, TTM_RECT_EXPONENTIAL); } else { CALL(5, 0, TTM_RECT_EXPONENTIAL); } } while (0)
#else
#define TTM_DENSE_DISPATCH(CALL, cls, rect)                                           \
    do {                                                                               \
        if ((cls).pp == 0 && (cls).ph == 3) TTM_DENSE_DISPATCH_RECT(CALL, 3, 0, rect); \
        else if ((cls).pp == 0 && (cls).ph == 5) TTM_DENSE_DISPATCH_RECT(CALL, 5, 0, rect); \
        else if ((cls).pp == 0 && (cls).ph == 8) TTM_DENSE_DISPATCH_RECT(CALL, 8, 0, rect); \
        else if ((cls).pp == 0) TTM_DENSE_DISPATCH_RECT(CALL, 10, 0, rect);            \
        else if ((cls).ph == 0) TTM_DENSE_DISPATCH_RECT(CALL, 0, 10, rect);            \
        else TTM_DENSE_DISPATCH_RECT(CALL, 10, 10, rect);                              \
    } while (0)
#endif

}  // namespace ttm
