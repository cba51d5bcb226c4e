// ttm_uform.h - the "univariate form" (U-form) of a separable transport map and its evaluators.
//
// For a separable map whose terms are all univariate (no cross terms; TM:2554-2558 with the basis of
// TM:905-1026, 1096-1150) every component is a sum of functions of ONE variable each:
//
//     S_k(x) = c0_k + sum_{groups (k,j)} f_kj(x_j) + own_k(x_k) + G_k(x_k)
//
//   f_kj(x) = A(x) + exp(-x^2/4) B(x)   the nonmonotone terms on column j: A collects the plain polynomial terms,
//                                       B the Hermite-function terms (a_n folded in), both as MONOMIAL coefficients
//                                       (degree <= TTM_U_PMAX), so a group costs one Horner pass instead of a
//                                       three-term recurrence plus per-order weights;
//   own_k    the same form for polynomial / Hermite-function terms of the monotone list (on x_k);
//   G_k(t)   the weighted sum of the component's special terms (LET / RET / RBF / iRBF).  Each of them needs an erf
//            and a Gaussian; their SUM is one smooth univariate function, linear outside the support of its terms.
//            It is compiled, whenever the coefficients change, into a piecewise polynomial: uniform intervals of
//            width h <= kappa * (smallest scale), degree 11 interpolation at Chebyshev nodes, exact linear tails.
//            Evaluating it costs one index computation, 12 table reads and 11 FMAs whatever the number of special
//            terms - the forward map of BASELINE config C5 goes from ~300 to ~60 fp64 instructions per
//            component evaluation.  The fit is verified when it is built (max error against the direct
//            evaluation at the extrema of T_12, value and derivative) and the host falls back to the direct
//            kernels when the error exceeds its tolerance.
//
// The builder (`uform_*`, run by ttm_fold after the folded coefficients) and the per-sample evaluators live here
// as host/device inline functions: the HIP kernels wrap them with LDS staging, tests/hostemu runs the same bodies
// on the host.
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/ttm.h"
#include "ttm_cheb_table.h"
#include "ttm_eval.h"

namespace ttm {

#if defined(__HIPCC__)
__device__ const double g_cheb_nodes[TTM_CHEB_N] = { TTM_CHEB_NODES };
__device__ const double g_cheb_vinv_hi[TTM_CHEB_N * TTM_CHEB_N] = { TTM_CHEB_VINV_HI };
__device__ const double g_cheb_vinv_lo[TTM_CHEB_N * TTM_CHEB_N] = { TTM_CHEB_VINV_LO };
#endif
static const double h_cheb_nodes[TTM_CHEB_N] = { TTM_CHEB_NODES };
static const double h_cheb_vinv_hi[TTM_CHEB_N * TTM_CHEB_N] = { TTM_CHEB_VINV_HI };
static const double h_cheb_vinv_lo[TTM_CHEB_N * TTM_CHEB_N] = { TTM_CHEB_VINV_LO };

#if defined(__HIP_DEVICE_COMPILE__)
#define TTM_CHEB_NODE(m) g_cheb_nodes[m]
#define TTM_CHEB_VH(i) g_cheb_vinv_hi[i]
#define TTM_CHEB_VL(i) g_cheb_vinv_lo[i]
#else
#define TTM_CHEB_NODE(m) h_cheb_nodes[m]
#define TTM_CHEB_VH(i) h_cheb_vinv_hi[i]
#define TTM_CHEB_VL(i) h_cheb_vinv_lo[i]
#endif

static_assert(TTM_CHEB_DEG == TTM_U_DEG, "spline degree");

// ---------------------------------------------------------------------------
// builder
// ---------------------------------------------------------------------------

// value and d/dt of G(t) = sum of the unified special-term records {centre, 1/(sqrt2 scale), A1, B0, B1, G, DG, DT}
// (include/ttm.h, TTM_FD_ST8), with the library erf / exp: this is the function the spline interpolates
TTM_HD void u_st_direct(const double* rec, int n_st, double t, double& v, double& dv) {
    double a = 0.0, da = 0.0;
    for (int s = 0; s < n_st; ++s) {
        const double* r = rec + 8 * s;
        const double d = t - r[0];
        const double tt = d * r[1];
        const double e = erf(tt);
        const double gs = exp(-(tt * tt));
        const double hh = fma(r[4], e, r[3]);
        a += r[2] * e + d * hh + r[5] * gs;
        da += hh + fma(tt, r[7], r[6]) * gs;
    }
    v = a; dv = da;
}

// asymptotes: G(t) -> icpt + slope * t for t below (side = -1) / above (side = +1) the support of every term
TTM_HD void u_st_tail(const double* rec, int n_st, int side, double& icpt, double& slope) {
    double ic = 0.0, sl = 0.0;
    for (int s = 0; s < n_st; ++s) {
        const double* r = rec + 8 * s;
        const double hh = r[3] + side * r[4];
        ic += side * r[2] - r[0] * hh;
        sl += hh;
    }
    icpt = ic; slope = sl;
}

// header doubles and monomial coefficients of the groups of one component.
//   uc: the component's TTM_UC record, ug: all group records, umono: TTM_U_PMAX+1 rows of monomial coefficients
//   (row n = P_n), foldk: the component's folded coefficients, fd: its fast-path descriptor, geo: {t_lo, h}
TTM_HD void uform_build_groups(const int* uc, const int* ug, const int* fd, const double* umono, const double* geo,
                               const double* foldk, double* U, int first, int stride) {
    const int ng = uc[TTM_UC_N_GRP] + (uc[TTM_UC_FLAGS] & TTM_UCF_OWN ? 1 : 0);
    double* cd = U + uc[TTM_UC_DBL_OFF];
    const int* g0 = ug + TTM_UG_LEN * uc[TTM_UC_GRP_OFF];
    for (int idx = first; idx < ng * TTM_U_GSTRIDE; idx += stride) {
        const int g = idx / TTM_U_GSTRIDE, j = idx % TTM_U_GSTRIDE, deg = j % TTM_U_GHALF;
        const int* G = g0 + TTM_UG_LEN * g;
        const int src = (j < TTM_U_GHALF) ? G[TTM_UG_SRCB] : G[TTM_UG_SRCA];
        const int P = (j < TTM_U_GHALF) ? G[TTM_UG_PB] : G[TTM_UG_PA];
        double acc = 0.0;
        if (src >= 0 && deg <= TTM_U_PMAX)
            for (int n = 1; n <= P; ++n) acc = fma(foldk[src + n - 1], umono[n * (TTM_U_PMAX + 1) + deg], acc);
        cd[4 + idx] = acc;
    }
    if (first == 0) {
        cd[0] = foldk[0] + foldk[fd[TTM_FD_ST8]];
        const double h = geo[1];
        cd[1] = uc[TTM_UC_NI] ? -geo[0] / h : 0.0;
        cd[2] = uc[TTM_UC_NI] ? 1.0 / h : 0.0;
        cd[3] = uc[TTM_UC_NI] ? 2.0 / h : 0.0;
    }
}

// phase 1 of the spline: G at the Chebyshev nodes of every interior interval -> ybuf[interval * 12 + node]
TTM_HD void uform_spline_nodes(const int* uc, const int* fd, const double* geo, const double* foldk, double* ybuf,
                               int first, int stride) {
    const int n_int = uc[TTM_UC_NI] - 2;
    const double* rec = foldk + fd[TTM_FD_ST8] + 8;
    const int n_st = fd[TTM_FD_N_ST];
    const double t_lo = geo[0], h = geo[1];
    for (int idx = first; idx < n_int * TTM_CHEB_N; idx += stride) {
        const int i = idx / TTM_CHEB_N, m = idx % TTM_CHEB_N;
        const double t = t_lo + ((double)i + 0.5 + 0.5 * TTM_CHEB_NODE(m)) * h;
        double v, dv;
        u_st_direct(rec, n_st, t, v, dv);
        ybuf[idx] = v;
    }
}

// phase 2: monomial coefficients in the local coordinate s in [-1,1] of every interval (double-double
// accumulation of VINV . y), plus the two tail columns (exactly linear in s)
TTM_HD void uform_spline_fit(const int* uc, const int* fd, const double* geo, const double* foldk, const double* ybuf,
                             double* U, int first, int stride) {
    const int nI = uc[TTM_UC_NI], n_int = nI - 2;
    double* tab = U + uc[TTM_UC_TAB_OFF];
    for (int idx = first; idx < n_int * TTM_CHEB_N; idx += stride) {
        const int i = idx / TTM_CHEB_N, j = idx % TTM_CHEB_N;
        const double* y = ybuf + i * TTM_CHEB_N;
        double sh = 0.0, sl = 0.0;
        for (int m = 0; m < TTM_CHEB_N; ++m) {
            const double vh = TTM_CHEB_VH(j * TTM_CHEB_N + m), vl = TTM_CHEB_VL(j * TTM_CHEB_N + m);
            const double p = vh * y[m];
            const double pe = fma(vh, y[m], -p) + vl * y[m];
            const double s = sh + p;
            const double bb = s - sh;
            const double se = (sh - (s - bb)) + (p - bb);
            sh = s;
            sl += se + pe;
        }
        tab[(i + 1) * TTM_U_TSTRIDE + j] = sh + sl;
    }
    // tails: column 0 is centred at t_lo - h/2, column nI-1 at t_hi + h/2; t - centre = s h / 2
    const double* rec = foldk + fd[TTM_FD_ST8] + 8;
    const int n_st = fd[TTM_FD_N_ST];
    for (int idx = first; idx < 2 * TTM_U_TSTRIDE; idx += stride) {
        const int side = idx / TTM_U_TSTRIDE, j = idx % TTM_U_TSTRIDE;
        double ic, sl;
        u_st_tail(rec, n_st, side ? 1 : -1, ic, sl);
        const double h = geo[1];
        const double centre = side ? geo[0] + ((double)n_int + 0.5) * h : geo[0] - 0.5 * h;
        const double v = j == 0 ? fma(sl, centre, ic) : (j == 1 ? sl * (0.5 * h) : 0.0);
        tab[(side ? nI - 1 : 0) * TTM_U_TSTRIDE + j] = v;
    }
    for (int idx = first; idx < n_int * (TTM_U_TSTRIDE - TTM_CHEB_N); idx += stride)      // padding of the interior columns
        tab[(idx / (TTM_U_TSTRIDE - TTM_CHEB_N) + 1) * TTM_U_TSTRIDE + TTM_CHEB_N + idx % (TTM_U_TSTRIDE - TTM_CHEB_N)] = 0.0;
}

// two int32 in the bits of one double (little endian: lo first) - the int fields of the hot records
TTM_HD double u_pack2(int lo, int hi) {
    const unsigned long long u = (unsigned long long)(unsigned int)lo | ((unsigned long long)(unsigned int)hi << 32);
    double d;
    memcpy(&d, &u, 8);
    return d;
}

TTM_HD int u_h_gs(int cls) { return cls == 1 ? 8 : (cls == 2 ? 16 : 24); }
TTM_HD int u_h_db(int cls) { return cls == 1 ? 3 : (cls == 2 ? 5 : (cls == 3 ? 7 : 10)); }
TTM_HD int u_h_da(int cls) { return cls == 1 ? 1 : (cls == 2 ? 5 : (cls == 3 ? 7 : 10)); }

// hot record of one component (include/ttm.h "H section"), from its U-form block; run after uform_build_groups.
// nm0: the constant of the nonmonotone part alone (folded[0]; c0 also carries the monotone constants)
TTM_HD void uform_build_hot(const int* uc, const int* ug, double* U, int64_t h_off, int cls, int ng, int k, double nm0,
                            int first, int stride) {
    const int GS = u_h_gs(cls), DB = u_h_db(cls), DA = u_h_da(cls);
    const int hs = TTM_H_HDR + ng * GS;
    double* rec = U + h_off + (int64_t)k * hs;
    const double* cd = U + uc[TTM_UC_DBL_OFF];
    const int n_grp = uc[TTM_UC_N_GRP];
    for (int idx = first; idx < hs; idx += stride) {
        double v = 0.0;
        if (idx < TTM_H_HDR) {
            if (idx == 0) v = u_pack2(uc[TTM_UC_KC_SLOT] >= 0 ? 2 * uc[TTM_UC_KC_SLOT] : -1, (uc[TTM_UC_FLAGS] & TTM_UCF_PUT_E) ? 1 : 0);
            else if (idx == 1) v = u_pack2(uc[TTM_UC_NI], uc[TTM_UC_KC]);
            else if (idx <= 5) v = cd[idx - 2];
            else if (idx == 6) v = u_pack2(uc[TTM_UC_TAB_OFF], n_grp);
            else if (idx == 7) v = nm0;
        } else {
            const int g = (idx - TTM_H_HDR) / GS, j = (idx - TTM_H_HDR) % GS;
            if (g < n_grp) {
                const int* G = ug + TTM_UG_LEN * (uc[TTM_UC_GRP_OFF] + g);
                const int fl = G[TTM_UG_FLAGS];
                if (j == 0) v = u_pack2(2 * TTM_PLAN_SLOT(fl), 1);
                else if (j <= 1 + DB) v = (fl & TTM_PLAN_HF) ? cd[4 + TTM_U_GSTRIDE * g + (j - 1)] : 0.0;
                else if (j <= 2 + DB + DA) v = (fl & TTM_UGF_POLY) ? cd[4 + TTM_U_GSTRIDE * g + TTM_U_GHALF + (j - 2 - DB)] : 0.0;
            } else if (j == 0) {
                v = u_pack2(0, 0);
            }
        }
        rec[idx] = v;
    }
}

// Push records (include/ttm.h "push records", csrc/ttm_band.hip) from the hot record of ONE component: every slot of the
// P section has exactly one writer - the component it describes - so the workgroup that has just built a component's hot
// record scatters its share at once (no second launch behind the U-form builder):
//   * record r = k            : [0], [1]  chain starts of component k (its constants + the constant terms of its groups)
//   * record r = k + lag      : [2..7]    spline geometry, {NI, TAB_OFF}, {k, 0}, slope of the linear own term; padding zeros
//   * record r = k + lag - lg : block lg-1: the group of component k that reads the column lg in front of it (zeros: none)
//   * the local-coordinate offset of every spline column of component k (padding slot 12 of the column)
// Slots that describe no component - the records of the lag columns in front of the first component, chain starts and
// blocks beyond the last - are zeroed by the first / last component's workgroup.  Values and summation order are those of
// the stand-alone builder it replaces (k_band_records), bit for bit.  Call after the hot record is complete (barrier).
TTM_HD void uform_scatter_push_records(const int* ucomp, const int* ugrp, double* U, int64_t h_off, int cls, int ng, int64_t p_off,
                                       int lag, int ps, int D, int k, int first, int stride) {
    const int DB = u_h_db(cls), DA = u_h_da(cls), GS = u_h_gs(cls), GP = DB + 1 + DA;
    const int hs = TTM_H_HDR + ng * GS;
    const int* uc = ucomp + k * TTM_UC_LEN;
    const double* h = U + h_off + (int64_t)k * hs;
    double* P = U + p_off;
    const int n_grp = uc[TTM_UC_N_GRP];
    const bool own = (uc[TTM_UC_FLAGS] & TTM_UCF_OWN) != 0;
    const double* ownc = U + uc[TTM_UC_DBL_OFF] + 4 + TTM_U_GSTRIDE * n_grp + TTM_U_GHALF;          // {constant, slope} of the linear own term
    // chain starts of component k -> record k
    for (int i = first; i < 2; i += stride) {
        double v = i == 0 ? h[2] : h[7];
        for (int g = 0; g < n_grp; ++g) v += h[TTM_H_HDR + g * GS + 2 + DB];
        if (i == 0) v += own ? ownc[0] : 0.0;
        P[(int64_t)k * ps + i] = v;
    }
    // header and padding of record k + lag
    {
        double* rec = P + (int64_t)(k + lag) * ps;
        for (int i = 2 + first; i < ps; i += stride) {
            if (i >= TTM_P_HDR && i < TTM_P_HDR + lag * GP) continue;
            double v = 0.0;
            if (i == 2) v = h[3] + 1.0;
            else if (i == 3) v = h[4];
            else if (i == 4) v = h[5];
            else if (i == 5) v = u_pack2(uc[TTM_UC_NI], uc[TTM_UC_TAB_OFF]);
            else if (i == 6) v = u_pack2(k, 0);
            else if (i == 7) v = own ? ownc[1] : 0.0;
            rec[i] = v;
        }
    }
    // group blocks
    for (int idx = first; idx < lag * GP; idx += stride) {
        const int l = idx / GP, j = idx % GP;
        int g = -1;
        for (int q = 0; q < n_grp; ++q)
            if (uc[TTM_UC_KC] - ugrp[(uc[TTM_UC_GRP_OFF] + q) * TTM_UG_LEN + TTM_UG_VAR] == l + 1) { g = q; break; }
        double v = 0.0;
        if (g >= 0) {
            const double* gr = h + TTM_H_HDR + g * GS;
            v = j <= DB ? gr[1 + j] : gr[2 + DB + (j - DB)];
        }
        P[(int64_t)(k - (l + 1) + lag) * ps + TTM_P_HDR + idx] = v;
    }
    // local-coordinate offsets of the spline columns
    {
        double* tab = U + uc[TTM_UC_TAB_OFF];
        for (int c = first; c < uc[TTM_UC_NI]; c += stride) tab[c * TTM_U_TSTRIDE + 12] = fma(2.0, h[3], -(double)(2 * c - 1));
    }
    // what belongs to no component
    if (k == 0) {
        for (int idx = first; idx < lag * ps; idx += stride) {
            const int r = idx / ps, i = idx % ps;
            if (i < 2) continue;                                              // (chain starts: component r's)
            if (i >= TTM_P_HDR && i < TTM_P_HDR + lag * GP) {
                const int l = (i - TTM_P_HDR) / GP;
                if ((r - lag) + l + 1 >= 0) continue;                         // (a component's block)
            }
            P[(int64_t)r * ps + i] = i == 6 ? u_pack2(-1, 0) : 0.0;
        }
    }
    if (k == D - 1) {
        for (int idx = first; idx < (D + lag) * ps; idx += stride) {
            const int r = idx / ps, i = idx % ps;
            if (i < 2) {
                if (r >= D) P[(int64_t)r * ps + i] = 0.0;
            } else if (i >= TTM_P_HDR && i < TTM_P_HDR + lag * GP) {
                const int l = (i - TTM_P_HDR) / GP;
                if ((r - lag) + l + 1 >= D) P[(int64_t)r * ps + i] = 0.0;
            }
        }
    }
}

// one column of the spline at local coordinate s: value and d/ds
TTM_HD void u_spline_column(const double* col, double s, double& p, double& dp) {
    double a = col[TTM_U_DEG], da = 0.0;
    for (int j = TTM_U_DEG - 1; j >= 0; --j) {
        da = fma(da, s, a);
        a = fma(a, s, col[j]);
    }
    p = a; dp = da;
}

// phase 3: largest error of the fit against the direct evaluation, value and derivative, relative to 1 + |exact|,
// probed at the extrema of T_12 of every interior interval and one interval into each tail.
TTM_HD void uform_spline_verify(const int* uc, const int* fd, const double* geo, const double* foldk, const double* U,
                                int first, int stride, double& err_v, double& err_d) {
    const int nI = uc[TTM_UC_NI], n_int = nI - 2;
    const double* tab = U + uc[TTM_UC_TAB_OFF];
    const double* rec = foldk + fd[TTM_FD_ST8] + 8;
    const int n_st = fd[TTM_FD_N_ST];
    const double t_lo = geo[0], h = geo[1];
    double ev = 0.0, ed = 0.0;
    const int NP = TTM_CHEB_N + 1;
    for (int idx = first; idx < nI * NP; idx += stride) {
        const int col = idx / NP, q = idx % NP;
        const double s = cos(3.14159265358979323846 * (double)q / (double)TTM_CHEB_N);
        const double t = t_lo + ((double)(col - 1) + 0.5 + 0.5 * s) * h;
        double v, dv, p, dp;
        u_st_direct(rec, n_st, t, v, dv);
        u_spline_column(tab + col * TTM_U_TSTRIDE, s, p, dp);
        dp *= 2.0 / h;
        const double a = fabs(p - v) / (1.0 + fabs(v)), b = fabs(dp - dv) / (1.0 + fabs(dv));
        ev = (a > ev || a != a) ? a : ev;
        ed = (b > ed || b != b) ? b : ed;
    }
    err_v = ev; err_d = ed;
}

// ---------------------------------------------------------------------------
// evaluators (R = double or VecD<N>)
// ---------------------------------------------------------------------------

TTM_HD double vfloor(double a) { return floor(a); }
template <int N> TTM_HD VecD<N> vfloor(const VecD<N>& a) {
    VecD<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = floor(a.v[i]);
    return r;
}

// Horner pass over P+1 uniform coefficients: value and (DER) derivative
template <int P, bool DER, class R>
TTM_HD void u_horner_fixed(cdbl_p c, const R& x, R& v, R& dv) {
    double k[P + 1];
#pragma unroll
    for (int j = 0; j <= P; ++j) k[j] = c[j];
    R a(k[P]), da(0.0);
#pragma unroll
    for (int j = P - 1; j >= 0; --j) {
        if (DER) da = vfma(da, x, a);
        a = vfma(a, x, k[j]);
    }
    v = a; dv = da;
}

template <bool DER, class R>
TTM_HD void u_horner(int P, cdbl_p c, const R& x, R& v, R& dv) {
    switch (P) {
        case 1: u_horner_fixed<1, DER>(c, x, v, dv); break;
        case 2: u_horner_fixed<2, DER>(c, x, v, dv); break;
        case 3: u_horner_fixed<3, DER>(c, x, v, dv); break;
        case 4: u_horner_fixed<4, DER>(c, x, v, dv); break;
        case 5: u_horner_fixed<5, DER>(c, x, v, dv); break;
        case 6: u_horner_fixed<6, DER>(c, x, v, dv); break;
        case 7: u_horner_fixed<7, DER>(c, x, v, dv); break;
        case 8: u_horner_fixed<8, DER>(c, x, v, dv); break;
        case 9: u_horner_fixed<9, DER>(c, x, v, dv); break;
        default: u_horner_fixed<TTM_U_PMAX, DER>(c, x, v, dv); break;
    }
}

// Horner pass of compile-time degree DEG (>= 0) or of the run-time degree P (DEG < 0)
template <int DEG, bool DER, class R>
TTM_HD void u_poly(int P, cdbl_p c, const R& x, R& v, R& dv) {
    if (DEG >= 0) u_horner_fixed<(DEG >= 0 ? DEG : 0), DER>(c, x, v, dv);
    else u_horner<DER>(P, c, x, v, dv);
}

// sum of the nonmonotone groups of a component (constant c0 included).
// DB / DA >= 0: every group is evaluated with these degrees (coefficients beyond a group's own degree are zero in
// the U section), no run-time dispatch; < 0: per-group degrees from the flag word.
// A group whose column and (for Hermite-function terms) exp(-x^2/4) are in the planned cache - every group of a
// full sweep over columns produced by the sweep itself - takes the two-read fast path.
template <int DB, int DA, class R, class Fetch>
TTM_HD R u_nonmon(cint_p uc, cint_p ug_all, cdbl_p U, Fetch& x) {
    cdbl_p cd = U + uc[TTM_UC_DBL_OFF];
    const int n_grp = uc[TTM_UC_N_GRP];
    cint_p ug = ug_all + TTM_UG_LEN * uc[TTM_UC_GRP_OFF];
    R s(cd[0]);
    for (int g = 0; g < n_grp; ++g) {
        const int var = ug[TTM_UG_LEN * g + TTM_UG_VAR], fl = ug[TTM_UG_LEN * g + TTM_UG_FLAGS];
        cdbl_p rec = cd + 4 + TTM_U_GSTRIDE * g;
        R xv, e(0.0), v, dv;
        if ((fl & (TTM_PLAN_HF | TTM_PLAN_XHIT | TTM_PLAN_EHIT)) == (TTM_PLAN_HF | TTM_PLAN_XHIT | TTM_PLAN_EHIT)) {
            const int slot = TTM_PLAN_SLOT(fl);
            xv = x.st.get(2 * slot);
            e = x.st.get(2 * slot + 1);
        } else {
            x.fetch(var, fl, xv, e);
        }
        if (fl & TTM_PLAN_HF) {
            u_poly<DB, false>(TTM_UG_DEGB(fl), rec, xv, v, dv);
            s = vfma(e, v, s);
        }
        if (fl & TTM_UGF_POLY) {
            u_poly<DA, false>(TTM_UG_DEGA(fl), rec + TTM_U_GHALF, xv, v, dv);
            s = s + v;
        }
    }
    return s;
}

// polynomial / Hermite-function terms of the monotone list (functions of x_k): value and derivative;
// ek = exp(-x_k^2/4) (only read when the group has Hermite-function terms)
template <bool DER, class R>
TTM_HD void u_own(cint_p uc, cint_p ug_all, cdbl_p U, const R& xk, const R& ek, R& m, R& dm) {
    const int g = uc[TTM_UC_N_GRP];
    cint_p G = ug_all + TTM_UG_LEN * (uc[TTM_UC_GRP_OFF] + g);
    cdbl_p rec = U + uc[TTM_UC_DBL_OFF] + 4 + TTM_U_GSTRIDE * g;
    const int fl = G[TTM_UG_FLAGS];
    R v, dv;
    if (fl & TTM_PLAN_HF) {
        u_horner<DER>(TTM_UG_DEGB(fl), rec, xk, v, dv);
        m = vfma(ek, v, m);
        if (DER) dm = vfma(ek, vfma(-0.5 * xk, v, dv), dm);      // d/dx [e^{-x^2/4} B] = e^{-x^2/4} (B' - x B / 2)
    }
    if (fl & TTM_UGF_POLY) {
        u_horner<DER>(TTM_UG_DEGA(fl), rec + TTM_U_GHALF, xk, v, dv);
        m = m + v;
        if (DER) dm = dm + dv;
    }
}

// the special-term spline at t: value and d/dt.  tab: the component's table ([column][TTM_U_TSTRIDE]), nI columns
template <bool DER, class R>
TTM_HD void u_spline(const double* tab, int nI, double sp_a, double sp_b, double sp_ds, const R& t, R& g, R& dg) {
    const R u = vfma(t, sp_b, sp_a);                         // (t - t_lo) / h
    const R fl = vfloor(vmin(vmax(u, -1.0), (double)(nI - 2)));
    const R s = vfma(2.0, u - fl, -1.0);                     // local coordinate; unbounded in the (linear) tails
    const typename int_of<R>::type col = vtoint(fl);
    R p, dp(0.0);
#pragma unroll
    for (int e = 0; e < lanes_of<R>::value; ++e) {
        const double* cp = tab + (ielem(col, e) + 1) * TTM_U_TSTRIDE;       // 16-byte aligned (even stride, even offsets)
        const double se = elem(s, e);
        double c[TTM_U_DEG + 1];
#pragma unroll
        for (int j = 0; j <= TTM_U_DEG; j += 2) load_pair(cp + j, c[j], c[j + 1]);
        double a = c[TTM_U_DEG], da = 0.0;
#pragma unroll
        for (int j = TTM_U_DEG - 1; j >= 0; --j) {
            if (DER) da = fma(da, se, a);
            a = fma(a, se, c[j]);
        }
        set_elem(p, e, a);
        set_elem(dp, e, da);
    }
    g = p;
    dg = DER ? dp * sp_ds : dp;
}

// The same spline in phases, for the straight-line hot path: index arithmetic for all samples of the thread
// (u_spline_index), the gather of the 12 coefficients of samples [E0, E0 + H) (u_spline_gather: issued together, so
// that their LDS latencies overlap each other and whatever is scheduled behind them) ...
template <class R>
struct SplineIdx {
    R s;                                         // local coordinates
    typename int_of<R>::type col;                // spline columns
};
template <class R>
TTM_HD void u_spline_index(int nI, double sp_a, double sp_b, const R& t, SplineIdx<R>& ix) {
    const R u = vfma(t, sp_b, sp_a);
    const R fl = vfloor(vmin(vmax(u, -1.0), (double)(nI - 2)));
    ix.s = vfma(2.0, u - fl, -1.0);
    ix.col = vtoint(fl);
}
template <int H>
struct SplineRegs { double c[H][TTM_U_DEG + 1]; };
template <int E0, int H, class R>
TTM_HD void u_spline_gather(const double* tab, const SplineIdx<R>& ix, SplineRegs<H>& q) {
#pragma unroll
    for (int e = 0; e < H; ++e) {
        const double* cp = tab + (ielem(ix.col, E0 + e) + 1) * TTM_U_TSTRIDE;
#pragma unroll
        for (int j = 0; j <= TTM_U_DEG; j += 2) load_pair(cp + j, q.c[e][j], q.c[e][j + 1]);
    }
}
// ... and their Horner passes, interleaved over the samples (same operations per sample as u_spline: bitwise equal);
// writes elements [E0, E0 + H) of g / dg (dg still to be scaled by the caller: d/dt = sp_ds d/ds)
template <int E0, int H, bool DER, class R>
TTM_HD void u_spline_eval(const SplineIdx<R>& ix, const SplineRegs<H>& q, R& g, R& dg) {
    double a[H], da[H];
#pragma unroll
    for (int e = 0; e < H; ++e) { a[e] = q.c[e][TTM_U_DEG]; da[e] = 0.0; }
#pragma unroll
    for (int j = TTM_U_DEG - 1; j >= 0; --j) {
#pragma unroll
        for (int e = 0; e < H; ++e) {
            const double se = elem(ix.s, E0 + e);
            if (DER) da[e] = fma(da[e], se, a[e]);
            a[e] = fma(a[e], se, q.c[e][j]);
        }
    }
#pragma unroll
    for (int e = 0; e < H; ++e) { set_elem(g, E0 + e, a[e]); set_elem(dg, E0 + e, da[e]); }
}

// Scheduling fences between the phases (tuning knob, off): with them a lone workgroup evaluates 13 % faster (every
// LDS latency is behind arithmetic), but next to the loader waves and the per-step barrier the kernel is slower
// (0.22 against 0.17 ms at C5) - the waves of a workgroup then all want the LDS, then all the VALU, at the same time.
#if defined(__HIP_DEVICE_COMPILE__) && defined(TTM_HL_PHASES)
#define TTM_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)      // nothing is scheduled across this point
#else
#define TTM_SCHED_FENCE() ((void)0)
#endif

// S_k and dS_k/dx_k of one component in U-form; afterwards x_k (and exp(-x_k^2/4) when a later group reads it)
// goes into its planned cache slot.  x: a PlanCache.
template <int DB, int DA, bool DER, class R, class Fetch>
TTM_HD void u_component(cint_p uc, cint_p ug_all, cdbl_p U, const double* tab, const R& xk, Fetch& x, bool want_value,
                        R& S, R& dS) {
    const int flags = uc[TTM_UC_FLAGS], nI = uc[TTM_UC_NI], slot = uc[TTM_UC_KC_SLOT];
    const bool own = flags & TTM_UCF_OWN;
    const bool own_hf = own && ((ug_all + TTM_UG_LEN * (uc[TTM_UC_GRP_OFF] + uc[TTM_UC_N_GRP]))[TTM_UG_FLAGS] & TTM_PLAN_HF);
    const bool put_e = (flags & TTM_UCF_PUT_E) && slot >= 0;
    R ek(0.0);
    if (put_e || own_hf) ek = exp_q_fast(xk);
    R m(0.0), dm(0.0);
    if (own) u_own<DER>(uc, ug_all, U, xk, ek, m, dm);
    if (nI > 0) {
        cdbl_p cd = U + uc[TTM_UC_DBL_OFF];
        R g, dg;
        u_spline<DER>(tab, nI, cd[1], cd[2], cd[3], xk, g, dg);
        m = m + g;
        if (DER) dm = dm + dg;
    }
    S = want_value ? u_nonmon<DB, DA, R>(uc, ug_all, U, x) + m : m;
    dS = dm;
    if (slot >= 0) {
        x.st.set(2 * slot, xk);
        if (put_e) x.st.set(2 * slot + 1, ek);
    }
}

// exp(-x^2/4) of the hot paths: from the 2^(j/32) table when the kernel staged one (CacheStore::etab - the inverse
// kernel: -6 % instructions), else the generic exp (the forward kernel: with the table lookup in its long basic
// block the compiler keeps 30 more VGPRs live and a workgroup per CU is lost - measured slower)
#define TTM_HL_EXP(st, x) ((st).etab ? exp_q_tab((st).etab, (x)) : exp_q_fast(x))

// S_k and dS_k/dx_k from a hot record (include/ttm.h "H section"): NG group records of degree (DB, DA) and stride
// GS, every column from the planned cache (st; slot s of a sample set = st.get(s)), then the put of x_k.  All scalar
// loads of the step are at fixed offsets from `rec`.  A component that uses all NG records, has a spline and
// stores exp(-x_k^2/4) - every component of a banded map but the first and last few - runs as ONE basic block, so
// the scheduler can overlap the cache / table reads with the exp and Horner chains; the others take the guarded path.
template <int NG, int DB, int DA, int GS, bool DER, bool ETAB = false, class R, class ST>
TTM_HD void h_component(cdbl_p rec, const double* tab, const R& xk, const ST& st, bool want_value, R& S, R& dS) {
    cint_p ri = (cint_p)rec;
    const int put2 = ri[0], flg = ri[1], nI = ri[2], n_grp = ri[13];
    if (n_grp == NG && nI > 0 && (flg & 1) && put2 >= 0 && want_value) {
        // phase 1: cache pairs and the spline coefficients of the first half of the samples in flight together
        constexpr int L = lanes_of<R>::value, H = L > 1 ? L / 2 : 1;
        R xv[NG], ev[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int slot2 = ((cint_p)(rec + TTM_H_HDR + g * GS))[0];
            st.get2(slot2, xv[g], ev[g]);
        }
        SplineIdx<R> ix;
        u_spline_index(nI, rec[3], rec[4], xk, ix);
        SplineRegs<H> q;
        u_spline_gather<0, H>(tab, ix, q);
        TTM_SCHED_FENCE();
        // phase 2: arithmetic that needs none of them (the exp of the column cache) behind their latency
        const R ek = (ETAB ? exp_q_tab(st.etab, xk) : exp_q_fast(xk));
        TTM_SCHED_FENCE();
        // phase 3: spline of the first half; then the second half's gather behind the groups
        R m, dm(0.0);
        u_spline_eval<0, H, DER>(ix, q, m, dm);
        if (L > 1) {
            TTM_SCHED_FENCE();
            u_spline_gather<(L > 1 ? H : 0), H>(tab, ix, q);
        }
        R s(rec[2]);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            cdbl_p gr = rec + TTM_H_HDR + g * GS;
            R b, a, dv;
            u_horner_fixed<DB, false>(gr + 1, xv[g], b, dv);
            u_horner_fixed<DA, false>(gr + 2 + DB, xv[g], a, dv);
            s = vfma(ev[g], b, s) + a;
        }
        if (L > 1) {
            TTM_SCHED_FENCE();
            u_spline_eval<(L > 1 ? H : 0), H, DER>(ix, q, m, dm);
        }
        if (DER) dm = dm * rec[5];
        S = s + m;
        dS = dm;
        st.set2(put2, xk, ek);
        return;
    }
    R s(rec[2]);
    if (want_value) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g < n_grp) {
                cdbl_p gr = rec + TTM_H_HDR + g * GS;
                const int slot2 = ((cint_p)gr)[0];
                R xv, ev;
                st.get2(slot2, xv, ev);
                R b, a, dv;
                u_horner_fixed<DB, false>(gr + 1, xv, b, dv);
                u_horner_fixed<DA, false>(gr + 2 + DB, xv, a, dv);
                s = vfma(ev, b, s) + a;
            }
        }
    }
    R ek(0.0);
    if (flg & 1) ek = (ETAB ? exp_q_tab(st.etab, xk) : exp_q_fast(xk));
    R m(0.0), dm(0.0);
    if (nI > 0) u_spline<DER>(tab, nI, rec[3], rec[4], rec[5], xk, m, dm);
    S = s + m;
    dS = dm;
    if (put2 >= 0) {
        // (no later group needs exp(-x_k^2/4): the slot next to x_k is still DEFINED - the hot-record evaluators read both
        // halves of a slot and multiply the second by the group's Hermite-function polynomial, zero for such a group: what an
        // earlier launch left in LDS (+inf sentinels of the inverse tables) times zero would be NaN)
        st.set2(put2, xk, (flg & 1) ? ek : R(0.0));
    }
}

// nonmonotone part of a component from its hot record (what the inverse subtracts from z_k): same two paths as
// h_component
template <int NG, int DB, int DA, int GS, class R, class ST>
TTM_HD R h_offset(cdbl_p rec, const ST& st) {
    const int n_grp = ((cint_p)rec)[13];
    R s(rec[7]);
    if (n_grp == NG) {
        R xv[NG], ev[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int slot2 = ((cint_p)(rec + TTM_H_HDR + g * GS))[0];
            st.get2(slot2, xv[g], ev[g]);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            cdbl_p gr = rec + TTM_H_HDR + g * GS;
            R b, a, dv;
            u_horner_fixed<DB, false>(gr + 1, xv[g], b, dv);
            u_horner_fixed<DA, false>(gr + 2 + DB, xv[g], a, dv);
            s = vfma(ev[g], b, s) + a;
        }
        return s;
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g < n_grp) {
            cdbl_p gr = rec + TTM_H_HDR + g * GS;
            const int slot2 = ((cint_p)gr)[0];
            R xv, ev;
            st.get2(slot2, xv, ev);
            R b, a, dv;
            u_horner_fixed<DB, false>(gr + 1, xv, b, dv);
            u_horner_fixed<DA, false>(gr + 2 + DB, xv, a, dv);
            s = vfma(ev, b, s) + a;
        }
    }
    return s;
}

// put of a solved x_k (and exp(-x_k^2/4) when a later group reads it) into its planned cache slot
template <class R, class ST>
TTM_HD void h_put(cdbl_p rec, const ST& st, const R& xk) {
    const int put2 = ((cint_p)rec)[0], flg = ((cint_p)rec)[1];
    if (put2 >= 0) {
        if (flg & 1) st.set2(put2, xk, TTM_HL_EXP(st, xk));
        else st.set2(put2, xk, R(0.0));                      // (defined, see h_component)
    }
}

// np.searchsorted(xs, target) (left) with a bucket index: bk[q] = first index whose entry is >= tmin + q step
// (q = 0..nb; bk[0] = 0, bk[nb] = T).  Starts one bucket below the target's own (rounding of the bucket number)
// and scans forward four entries at a time - with nb ~ T the first group almost always decides.
TTM_HD int h_search(const double* xs, const int* bk, int nb, int T, double lo, double scale, bool use_bkt, double target) {
    int a = 0;
    if (use_bkt) {
        int q = (int)((target - lo) * scale) - 1;
        q = q < 0 ? 0 : (q > nb - 1 ? nb - 1 : q);
        a = bk[q];
    }
    // count entries < target from a on (xs is non-decreasing; NaN target: every compare is false -> a)
    while (a < T) {
        const int r = T - a;
        const double v0 = xs[a], v1 = xs[r > 1 ? a + 1 : a], v2 = xs[r > 2 ? a + 2 : a], v3 = xs[r > 3 ? a + 3 : a];
        const int c = (v0 < target ? 1 : 0) + ((r > 1 && v1 < target) ? 1 : 0) + ((r > 2 && v2 < target) ? 1 : 0) +
                      ((r > 3 && v3 < target) ? 1 : 0);
        a += c;
        if (c < 4) break;
    }
    return a;
}

}  // namespace ttm
