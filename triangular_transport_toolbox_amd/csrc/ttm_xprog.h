// ttm_xprog.h - integrated-rectifier components through their X PROGRAM (include/ttm.h; round 5).
//
// A component of a polynomial integrated-rectifier map (TM:2499-2547) is built from a handful of UNIVARIATE factor values
// on its conditioning columns - P_n(x_j), a_n P_n(x_j) exp(-x_j^2/4) - and the functions B_b of its own column:
//     Psi_nonmon,i = prod of <= 3 factor values,      Psi_mon,i(t) = A_i B_b(i)(t),  A_i = prod of <= 3 factor values.
// The term-table interpreter of ttm_eval.h walks term, factor, group and fold-recipe records per sample, once for the
// weights w_b = sum_i c_i A_i, once for the nonmonotone sum and a third time for the gradient (objective_gradient): 2 000
// of the 3 600 vector instructions per sample of round 4's objective kernel, 331-400 spilled SGPRs.  Here every distinct
// factor value ("U value") is evaluated ONCE per sample into the sample's ROW, and
//   * every distinct product A of them once more (a multiplication);
//   * the monomial form of g and the nonmonotone sum are then matrix-vector products with matrices that depend on the
//     COEFFICIENTS alone, h_j = sum_a A_a HC[a][j], a_j = sum_a A_a HP[a][j], Psi_nonmon c_nonmon = sum_a A'_a CN[a]: the
//     matrices are slots of the fold recipe (termtable._x_program), computed once per coefficient vector;
//   * objective and gradient (TM:3343-3376, 3475-3569): with S = S_k(x), the node moments mh, mp of ttm_dense.h and
//         qH_j = S x_k/2 mh_j - r'(g)/(r(g) + delta) E(x_k) x_k^j,      qP_j = S x_k/2 mp_j - r'(g)/(r(g) + delta) x_k^j
//     written into the row, EVERY sum the evaluation needs is a product of TWO row entries summed over the samples:
//         J = sum_n [ONE][J],     sum_n A'_a S  (nonmonotone gradients),     T[a][j] = sum_n A_a q_j,
//     and dJ/dc_mon,i = sum_j CB[b(i)][j] T[a(i)][j] with the basis-conversion row CB of its B function, applied once per
//     evaluation.  The kernel (csrc/ttm_int.hip: k_int_objective) gives a SUM to a LANE: after the 64 samples of a wave have
//     written their rows, lane t walks the 64 rows with the two columns of sum t and keeps its running total in a register -
//     three instructions per sample whatever the number of coefficients, no per-coefficient wave reduction (18
//     instructions each in round 4), no LDS accumulators.
// The bodies compile for the host as well (tests/hostemu runs them behind the C ABI's test double).
#pragma once

#include "ttm_dense.h"

namespace ttm {

struct XProg {               // view of one component's X program
    cint_p vars, prods, anm, amon, terms, bfuns;
    cdbl_p dpar, urows;
    int kc, nvar, nU, nprod, na_nm, na_mon, n_nm, n_mon, nB, nB_hf, nB_poly, nrow, nsum, fold_x, ph, pp;
};

// false when the component has no X program
TTM_HD bool xprog_view(cint_p cb, cdbl_p dpar, XProg& x) {
    const int off = TTM_UNI(cb[TTM_HDR_OFF_XPROG]);
    cint_p h = cb + off;
    x.kc = TTM_UNI(cb[TTM_HDR_KC]);
    x.n_nm = TTM_UNI(cb[TTM_HDR_N_NM]);
    x.n_mon = TTM_UNI(cb[TTM_HDR_N_MON]);
    x.nB = TTM_UNI(cb[TTM_HDR_NB]);
    x.nB_hf = TTM_UNI(cb[TTM_HDR_NB_HF]);
    x.nB_poly = TTM_UNI(cb[TTM_HDR_NB_POLY]);
    x.nrow = TTM_UNI(cb[TTM_HDR_X_NROW]);
    x.nsum = TTM_UNI(cb[TTM_HDR_X_NSUM]);
    x.bfuns = cb + TTM_UNI(cb[TTM_HDR_OFF_B]);
    x.nvar = TTM_UNI(h[0]);
    x.nU = TTM_UNI(h[1]);
    x.nprod = TTM_UNI(h[2]);
    x.na_nm = TTM_UNI(h[3]);
    x.na_mon = TTM_UNI(h[4]);
    x.fold_x = TTM_UNI(h[5]);
    x.urows = dpar + TTM_UNI(h[6]);
    x.ph = TTM_UNI(h[7]) & 255;
    x.pp = (TTM_UNI(h[7]) >> 8) & 255;
    x.vars = h + TTM_XH_LEN;
    x.prods = x.vars + 4 * x.nvar;
    x.anm = x.prods + 4 * x.nprod;
    x.amon = x.anm + ((x.na_nm + 3) & ~3);
    x.terms = x.amon + ((x.na_mon + 3) & ~3);
    x.dpar = dpar;
    return off != 0;
}

// q columns of the objective kernel for the order class (PH, PP): NH of the Hermite-function part, then PP + 1
template <int PH, int PP> struct XQ { static constexpr int NH = PH > 0 ? PH + 1 : 0, NQ = NH + PP + 1; };

// U values and their products into the sample's row.  P_n(x_j): Horner's rule on the value's monomial row (dpar: a_n folded
// in; the first PM + 1 coefficients - the program holds no factor of a higher order than the component's class), times
// exp(-x_j^2/4) for the Hermite functions of a column; every row address is known without a record load
template <int PM, class XA, class Row>
TTM_HD void xprog_values(const XProg& xp, const XA& xa, Row& row) {
    constexpr int LD = TTM_I_PMAX + 1;
    row.set(TTM_XR_ONE, 1.0);
    cdbl_p ur = xp.urows;
    int col = TTM_XR_U;
    for (int v = 0; v < xp.nvar; ++v) {
        cint_p V = xp.vars + 4 * v;
        const double x = xa(TTM_UNI(V[0]));
        const int np = TTM_UNI(V[1]), nh = TTM_UNI(V[2]);
        for (int j = 0; j < np; ++j, ++col, ur += LD) {
            double val = ur[PM];
#pragma unroll
            for (int q = PM - 1; q >= 0; --q) val = fma(val, x, ur[q]);
            row.set(col, val);
        }
        if (nh > 0) {
            const double E = dense_expq(x);
            for (int j = 0; j < nh; ++j, ++col, ur += LD) {
                double val = ur[PM];
#pragma unroll
                for (int q = PM - 1; q >= 0; --q) val = fma(val, x, ur[q]);
                row.set(col, val * E);
            }
        }
    }
    for (int p = 0; p < xp.nprod; ++p, ++col) {
        cint_p P = xp.prods + 4 * p;
        row.set(col, (row.get(TTM_UNI(P[0])) * row.get(TTM_UNI(P[1]))) * row.get(TTM_UNI(P[2])));
    }
}

// monomial form of g and the nonmonotone sum of one sample from its A values and the fold's X section fx = CN | HC | HP
// (FP: pointer to uniform data - constant address space on the device for the global fold buffer, plain for LDS / the host)
template <int PH, int PP, class FP, class Row>
TTM_HD void xprog_mono(const XProg& xp, FP fx, const Row& row, DenseMono<PH, PP>& d, double& s_nm) {
    constexpr int LD = TTM_I_PMAX + 1;
#pragma unroll
    for (int j = 0; j <= PH; ++j) d.h[j] = 0.0;
#pragma unroll
    for (int j = 0; j <= PP; ++j) d.a[j] = 0.0;
    FP hc = fx + xp.na_nm;
    FP hp = hc + LD * xp.na_mon;
    for (int a = 0; a < xp.na_mon; ++a, hc += LD, hp += LD) {
        const double A = row.get(TTM_UNI(xp.amon[a]));
        if (PH > 0) {
#pragma unroll
            for (int j = 0; j <= PH; ++j) d.h[j] = fma(A, hc[j], d.h[j]);
        }
#pragma unroll
        for (int j = 0; j <= PP; ++j) d.a[j] = fma(A, hp[j], d.a[j]);
    }
    double s = 0.0;
    for (int a = 0; a < xp.na_nm; ++a) s = fma(row.get(TTM_UNI(xp.anm[a])), fx[a], s);
    s_nm = s;
    double t = 0.0;
#pragma unroll
    for (int j = 0; j <= PH; ++j) t += d.h[j];
#pragma unroll
    for (int j = 0; j <= PP; ++j) t += d.a[j];
    d.probe = t * 0.0;
}

// S_k(x) and dS_k/dx_k of one sample (TM:2499-2547) through the X program: dense_sample_forward of ttm_dense.h with the term-table
// walks (weights of the B functions, their conversion to monomials, nonmonotone groups) replaced by the row
template <int PH, int PP, int RECT, bool DER, class FP, class XA, class Row>
TTM_HD void xprog_sample_forward(const XProg& xp, const Prog& p, double qw_sum, FP fx, const XA& xa, Row& row, bool want_value,
                                 double& S, double& dS) {
    constexpr int PM = PH > PP ? PH : PP;
    xprog_values<PM>(xp, xa, row);
    DenseMonoSet<PH, PP, RECT> s;
    double s_nm;
    xprog_mono<PH, PP>(xp, fx, row, s.d, s_nm);
    s.qw_sum = qw_sum;
    const double xk = xa(xp.kc);
    double m, dm;
    Comp unused;                                            // (mon_eval of a DenseMonoSet reads nothing of the component)
    mon_eval<TTM_MONO_INTEGRATED, DER>(unused, p, xk, s, m, dm);
    S = want_value ? s_nm + m : m;
    dS = dm;
}

// bisection (the reference's sequence, TM:3842-3976) or safeguarded Newton root search of one sample through the X program;
// xa: the sample's columns with the roots of the earlier components already in place
template <int PH, int PP, int RECT, bool NEWTON, class FP, class XA, class Row>
TTM_HD double xprog_sample_root(const XProg& xp, const Prog& p, double qw_sum, FP fx, const XA& xa, Row& row, double zk, int cap, int& it) {
    constexpr int PM = PH > PP ? PH : PP;
    xprog_values<PM>(xp, xa, row);
    DenseMonoSet<PH, PP, RECT> s;
    double off;
    xprog_mono<PH, PP>(xp, fx, row, s.d, off);
    s.qw_sum = qw_sum;
    Comp unused;
    return NEWTON ? sample_newton<TTM_MONO_INTEGRATED>(unused, p, off, zk, s, it) : sample_bisect<TTM_MONO_INTEGRATED>(unused, p, off, zk, s, cap, it);
}

#ifndef XOBJ_NODES
#define XOBJ_NODES 4          /* nodes per pass of the objective's node loop: 115-123 vector registers (four waves per SIMD); 5: 125-134 */
#endif
// The row of one sample for the objective + gradient sums (see the head of this file).  active = false (a lane beyond the
// ensemble, evaluated on a clamped sample): S, J and the q columns are zero, so the row adds nothing to any sum.
template <int PH, int PP, int RECT, class FP, class XA, class Row>
TTM_HD void xobj_sample_row(const XProg& xp, const Prog& p, double qw_sum, FP fx, const XA& xa, Row& row, bool active) {
    const int rect = (RECT >= 0) ? RECT : p.rect;
    constexpr int PM = PH > PP ? PH : PP;
    xprog_values<PM>(xp, xa, row);
    DenseMono<PH, PP> d;
    double s_nm;
    xprog_mono<PH, PP>(xp, fx, row, d, s_nm);
    const double xk = xa(xp.kc);
    const double half = xk * 0.5;
    double mono;
    double mh[PH + 1], mp[PP + 1];
    dense_moment_nodes<PH, PP, RECT, XOBJ_NODES>(p, d, xk, mono, mh, mp);
    const double nanp = d.probe + xk * 0.0;
    mono = half * fma(p.delta, qw_sum, mono) + nanp;
    const double S = s_nm + mono;
    // the log term and r'/(r + delta) at x_k (TM:3370-3376, 3540-3560)
    double E;
    const double g = dense_g(d, xk, E) + nanp;
    double r, dr, logr;
    rect_all(rect, p.delta, g, r, dr, logr);
    const double rinv = dr * fast_rcp(r + p.delta);
    const double sh = S * half, re = rinv * E;
    const int qb = xp.nrow;
    double tp = 1.0;
#pragma unroll
    for (int j = 0; j <= PM; ++j) {
        if (PH > 0 && j <= PH) row.set(qb + j, active ? sh * mh[j] - re * tp : 0.0);
        if (j <= PP) row.set(qb + XQ<PH, PP>::NH + j, active ? sh * mp[j] - rinv * tp : 0.0);
        tp = tp * xk;
    }
    row.set(TTM_XR_S, active ? S : 0.0);
    row.set(TTM_XR_J, active ? 0.5 * S * S - logr : 0.0);
}

// the two row columns of sum t: 0 = J; 1 + a = nonmonotone product a times S; then (monotone product a, q column j)
TTM_HD void xobj_sum_columns(const int* anm, const int* amon, int na_nm, int nrow, int NQ, int t, int& c1, int& c2) {
    if (t == 0) { c1 = TTM_XR_ONE; c2 = TTM_XR_J; return; }
    if (t <= na_nm) { c1 = anm[t - 1]; c2 = TTM_XR_S; return; }
    const int idx = t - 1 - na_nm;
    c1 = amon[idx / NQ];
    c2 = nrow + idx % NQ;
}

// Result i of an evaluation (0: J; 1 + i: the gradient w.r.t. coefficient i of [nonmon | mon]) from the sums T: a dot product
// of n of them, from T[base] on, with the coefficients cb (n = 1, cb = 1 for J, the nonmonotone gradients and monotone terms
// without a function of x_k; else the basis-conversion row of the term's B function, a_n folded in).  xobj_result_plan reads
// the program (the kernel's finishing workgroup calls it BEFORE it waits for the sums), xobj_result_apply is arithmetic only.
template <int PH, int PP>
struct XResult {
    static constexpr int PM = PH > PP ? PH : PP;
    int base, n;
    double cb[PM + 1];
};
template <int PH, int PP>
TTM_HD void xobj_result_plan(const XProg& xp, const Prog& p, int i, XResult<PH, PP>& r) {
    constexpr int LD = TTM_I_PMAX + 1, NH = XQ<PH, PP>::NH, NQ = XQ<PH, PP>::NQ;
#pragma unroll
    for (int j = 0; j <= XResult<PH, PP>::PM; ++j) r.cb[j] = 0.0;
    r.cb[0] = 1.0;
    r.n = 1;
    if (i == 0) { r.base = 0; return; }
    const int* terms = (const int*)xp.terms;
    if (i <= xp.n_nm) { r.base = 1 + terms[4 * (i - 1)]; return; }
    const int* rec = terms + 4 * (i - 1);
    const int a = rec[0], b = rec[1];
    r.base = 1 + xp.na_nm + a * NQ + NH;
    if (b >= xp.nB) return;
    const int* B = (const int*)xp.bfuns + 4 * b;
    const double* rm = (const double*)mono_table_of(p.family) + B[1] * LD;
    if (b < xp.nB_hf) {
        const double ab = ((const double*)xp.dpar)[B[2]];
        r.base -= NH;
        r.n = PH + 1;
#pragma unroll
        for (int j = 0; j <= PH; ++j) r.cb[j] = ab * rm[j];
    } else {
        r.n = PP + 1;
#pragma unroll
        for (int j = 0; j <= PP; ++j) r.cb[j] = rm[j];
    }
}
template <int PH, int PP, class TS>
TTM_HD double xobj_result_apply(const XResult<PH, PP>& r, const TS& T) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j <= XResult<PH, PP>::PM; ++j)
        if (j < r.n) s = fma(r.cb[j], T[r.base + j], s);
    return s;
}

}  // namespace ttm
