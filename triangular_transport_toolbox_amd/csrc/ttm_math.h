// ttm_math.h - fp64 elementary functions tuned for the transport-map kernels.
//
// CDNA4 has no fp64 transcendental unit: exp/erf/log are FMA chains, and the ROCm
// device-library versions cost 62 (exp), 343 (erf) and 118 (log) instructions on
// gfx950 with two divergent paths inside erf.  The map kernels are bound by
// exactly these (profiles/r01_v1_*), so they get branch-free replacements:
//   fast_exp : Cody-Waite reduction + degree-13 Taylor/Horner, ~22 instructions, <= 1 ulp
//   erf_tab  : piecewise degree-11 interpolants (32 intervals on [0,6), 3 KB, staged in
//              LDS, bank-conflict free), 11 FMA + 12 LDS reads, max abs error 1.1e-16; the same
//              coefficients give exp(-t^2) as the polynomial's derivative (11 more FMA, no exp call)
//   fast_log : fdlibm-style log with a Newton reciprocal, ~35 instructions, <= 2 ulp
//   fast_div : v_rcp_f64 + 2 Newton steps + residual correction (not IEEE-exact, <= 1 ulp)
// Accuracy is tested against NumPy/SciPy in tests/test_math.py (host build of the same code).
#pragma once

#include <math.h>
#include <stdint.h>

#include "ttm_erf_table.h"
#include "ttm_vec.h"

namespace ttm {

TTM_HD double fast_rcp(double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    return r;
#else
    return 1.0 / b;
#endif
}

TTM_HD double fast_div(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = fast_rcp(b);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);     // one residual correction
#else
    return a / b;
#endif
}

// a / b with one Newton step on v_rcp_f64 (2^-23 -> 2^-46) and the residual correction (-> < 1 ulp of the quotient
// for normal operands): two instructions fewer than fast_div; for quotients whose consumer tolerates 2 ulp
TTM_HD double fast_div1(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
#else
    return a / b;
#endif
}

// 1 / b to the 23 bits v_rcp_f64 delivers (bucket numbers, start values)
TTM_HD double approx_rcp(double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(b);
#else
    return 1.0 / b;
#endif
}

// Taylor coefficients 1/13! .. 1/2!, kept in constant memory: read by scalar loads into SGPRs and
// fed to v_fma_f64 as scalar operands (hoisting them into VGPRs costs 24 registers and a v_mov per step)
#if defined(__HIPCC__)
__device__ double g_exp_coef[12] = {   // (not const: keeps the compiler from folding the loads into literals)
#else
static const double g_exp_coef[12] = {
#endif
    1.6059043836821613e-10, 2.08767569878681e-09, 2.505210838544172e-08, 2.755731922398589e-07,
    2.7557319223985893e-06, 2.48015873015873e-05, 0.0001984126984126984, 0.001388888888888889,
    0.008333333333333333, 0.041666666666666664, 0.16666666666666666, 0.5};

// exp(y); R = double or VecD<N> (the Horner steps of the N samples interleave: independent FMA chains)
template <class R>
TTM_HD R fast_exp(const R& y) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __attribute__((address_space(4))) double* kc = (const __attribute__((address_space(4))) double*)g_exp_coef;
#else
    const double* kc = g_exp_coef;
#endif
    const R yc = vmin(vmax(y, -800.0), 800.0);
    const R k = vrint(yc * 1.4426950408889634);
    R r = vfma(-k, 6.93147180369123816490e-01, yc);
    r = vfma(-k, 1.90821492927058770002e-10, r);
    R p(kc[0]);
#pragma unroll
    for (int j = 1; j < 12; ++j) p = vfma(p, r, kc[j]);
    p = vfma(p, r, 1.0);
    p = vfma(p, r, 1.0);
    const R res = vldexp(p, vtoint(k));
    return vnan_to(y, y, res);
}

// exp(-x^2/4) for the column cache of the forward hot kernels: the same reduction and series as fast_exp, degree 12
// (truncation 0.347^13/13! = 1.7e-16: <= 2 ulp with the roundings) and cheaper guards: the argument is never positive
// (one clamp), and NaN is restored by adding 0 * x instead of a compare and two selects (the clamp's v_max drops it).
// x = +-inf gives NaN (0 * inf) where exp gives 0: the value only ever multiplies a polynomial of the same x, which is
// infinite there - the product is NaN in the reference too.  22 instructions against 26 for fast_exp(-0.25 * (x * x)).
template <class R>
TTM_HD R exp_q_fast(const R& x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __attribute__((address_space(4))) double* kc = (const __attribute__((address_space(4))) double*)g_exp_coef;
#else
    const double* kc = g_exp_coef;
#endif
    const R y = vmax(-0.25 * (x * x), -800.0);
    const R k = vrint(y * 1.4426950408889634);
    R r = vfma(-k, 6.93147180369123816490e-01, y);
    r = vfma(-k, 1.90821492927058770002e-10, r);
    R p(kc[1]);
#pragma unroll
    for (int j = 2; j < 12; ++j) p = vfma(p, r, kc[j]);
    p = vfma(p, r, 1.0);
    p = vfma(p, r, 1.0);
    return vfma(x, 0.0, vldexp(p, vtoint(k)));
}

// erf(t) and exp(-t^2) from the staged table (TTM_ERF_TABLE_LEN doubles, [coefficient][interval]; see
// tools/gen_erf_table.py for the geometry and why it is bank-conflict free).  The Gaussian is the derivative of
// the same local polynomial (erf' = 2/sqrt(pi) exp(-t^2)): no exp call.
// abs errors: erf 1.1e-16, exp(-t^2) 3.4e-16 (relative 2e-13 up to |t| < 4); |t| >= 6 is clamped to 6 (erf = +-1 within
// 1.2e-16, exp(-t^2) = 2.3e-16).  R = double or VecD<N>.
template <bool GAUSS, class R>
TTM_HD void erf_gauss_tab(const double* tab, const R& t, R& erfv, R& gauss) {
    const R a = vmin(vabs(t), 5.9999999);
    const typename int_of<R>::type i = vtoint(a * TTM_ERF_INV_WIDTH);
    const R d = vfma(-(vfromint(i) + 0.5), TTM_ERF_WIDTH, a);
    R p = vgather(tab + TTM_ERF_DEG * TTM_ERF_NINT, i), dp(0.0);
#pragma unroll
    for (int j = TTM_ERF_DEG - 1; j >= 0; --j) {
        const R cj = vgather(tab + j * TTM_ERF_NINT, i);
        if (GAUSS) dp = vfma(dp, d, p);
        p = vfma(p, d, cj);
    }
    erfv = vnan_to(t, t, vcopysign(p, t));
    gauss = GAUSS ? vnan_to(t, t, 0.88622692545275801365 * dp) : R(0.0);      // sqrt(pi)/2
}

TTM_HD double erf_tab(const double* tab, double t) {
    double e, g;
    erf_gauss_tab<false>(tab, t, e, g);
    return e;
}

// exp(-x^2/4) from a 32-entry table of 2^(j/32) (staged in LDS by the hot kernels: per-lane gathers, 32 entries =
// one 8-byte bank each, conflict-free): y = -x^2/4 = (32 e + j) ln2/32 + r, |r| <= ln2/64, exp(y) = 2^e T[j] p(r) with
// the degree-6 Taylor polynomial (truncation 3.5e-18): 22 instructions instead of the 30 of fast_exp, <= 2 ulp.
// NaN / inf: x = NaN gives NaN (through r); x = +-inf gives 0 (y is clamped at -800).  Meant for the cached
// E(x_j) of the map kernels, which only ever multiplies a polynomial of the same x_j.
#define TTM_EXPQ_TABLE_LEN 32
#define TTM_EXPQ_TABLE_VALUES \
    0x1.0000000000000p+0, 0x1.059b0d3158574p+0, 0x1.0b5586cf9890fp+0, 0x1.11301d0125b51p+0, \
    0x1.172b83c7d517bp+0, 0x1.1d4873168b9aap+0, 0x1.2387a6e756238p+0, 0x1.29e9df51fdee1p+0, \
    0x1.306fe0a31b715p+0, 0x1.371a7373aa9cbp+0, 0x1.3dea64c123422p+0, 0x1.44e086061892dp+0, \
    0x1.4bfdad5362a27p+0, 0x1.5342b569d4f82p+0, 0x1.5ab07dd485429p+0, 0x1.6247eb03a5585p+0, \
    0x1.6a09e667f3bcdp+0, 0x1.71f75e8ec5f74p+0, 0x1.7a11473eb0187p+0, 0x1.82589994cce13p+0, \
    0x1.8ace5422aa0dbp+0, 0x1.93737b0cdc5e5p+0, 0x1.9c49182a3f090p+0, 0x1.a5503b23e255dp+0, \
    0x1.ae89f995ad3adp+0, 0x1.b7f76f2fb5e47p+0, 0x1.c199bdd85529cp+0, 0x1.cb720dcef9069p+0, \
    0x1.d5818dcfba487p+0, 0x1.dfc97337b9b5fp+0, 0x1.ea4afa2a490dap+0, 0x1.f50765b6e4540p+0

TTM_HD double exp_q_tab(const double* tab, double x) {
    double y = -0.25 * (x * x);
    y = (y < -800.0) ? -800.0 : y;                                   // (a compare keeps NaN; fmax would drop it)
    const double k = rint(y * 0x1.71547652b82fep+5);                 // 32 / ln 2
    double r = fma(k, -0x1.62e42fee00000p-6, y);                     // ln2/32 = hi + lo, hi has 32 significant bits
    r = fma(k, -0x1.a39ef35793c76p-38, r);
    const int ki = (int)k;
    const double t = tab[ki & 31];
    double p = fma(0.001388888888888889, r, 0.008333333333333333);
    p = fma(p, r, 0.041666666666666664);
    p = fma(p, r, 0.16666666666666666);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(t * p, ki >> 5);
}
template <int N> TTM_HD VecD<N> exp_q_tab(const double* tab, const VecD<N>& x) {
    VecD<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = exp_q_tab(tab, x.v[i]);
    return r;
}

TTM_HD double fast_log(double x) {
    int e;
    double m = frexp(x, &e);                          // [0.5, 1)
    const bool small = m < 0.70710678118654752440;
    m = small ? m * 2.0 : m;
    e = small ? e - 1 : e;
    const double f = m - 1.0;
    const double s = fast_div(f, 2.0 + f);
    const double z = s * s;
    const double w = z * z;
    // fdlibm e_log.c polynomial (Lg1..Lg7), |error| < 2^-58.45
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                              6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    double res = dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
    if (!(x > 1.0e-300) || !(x < 1.0e300)) res = log(x);   // rare: 0, negative, NaN, inf, denormal range
    return res;
}

// ---- N samples per thread: the scalar routines applied per element (independent chains, the
// scheduler interleaves them) ---------------------------------------------------------------------
template <int N> TTM_HD VecD<N> fast_rcp(const VecD<N>& b) {
    VecD<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = fast_rcp(b.v[i]);
    return r;
}
template <int N> TTM_HD VecD<N> fast_div(const VecD<N>& a, const VecD<N>& b) {
    VecD<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = fast_div(a.v[i], b.v[i]);
    return r;
}
template <int N> TTM_HD VecD<N> fast_div(const VecD<N>& a, double b) {
    VecD<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = fast_div(a.v[i], b);
    return r;
}
template <int N> TTM_HD VecD<N> fast_log(const VecD<N>& x) {
    VecD<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = fast_log(x.v[i]);
    return r;
}
}  // namespace ttm
