// ttm_vec.h - "several samples per thread" value type.
//
// The map kernels are interpreters: the scalar unit walks the term tables (s_load, s_cmp, s_cbranch) and
// the vector unit does the per-sample fp64 arithmetic.  With one sample per lane the scalar unit is the
// bottleneck (profiles/r01_v6: ~500 SALU + ~365 VALU instructions per component evaluation, and a CU has
// one scalar pipe for its four SIMDs).  VecD<N> lets every thread carry N samples through the same scalar
// control flow: table decoding is paid once per N*64 samples and the N independent FMA chains hide each
// other's latency.  All per-sample evaluators in ttm_math.h / ttm_eval.h are templates over the value type
// R = double (N = 1) or VecD<N>; uniform quantities (coefficients, constants, table entries) stay `double`.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#define TTM_HD __host__ __device__ __forceinline__
#else
#define TTM_HD inline
#endif

namespace ttm {

template <int N>
struct VecD {
    double v[N];
    TTM_HD VecD() {}
    TTM_HD VecD(double s) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = s;
    }
    TTM_HD double& operator[](int i) { return v[i]; }
    TTM_HD const double& operator[](int i) const { return v[i]; }
};

template <int N>
struct VecI {
    int v[N];
    TTM_HD int& operator[](int i) { return v[i]; }
    TTM_HD const int& operator[](int i) const { return v[i]; }
};

template <class R> struct lanes_of { static constexpr int value = 1; };
template <int N> struct lanes_of<VecD<N>> { static constexpr int value = N; };

// 16-byte aligned pair of doubles moved as ONE access (device: ds_read_b128 / global_load_dwordx4 - the compiler
// otherwise emits ds_read2_b64, which the LDS serves at half the rate and banks modulo 32); host: two plain loads
#if defined(__HIP_DEVICE_COMPILE__)
typedef double ttm_pair_t __attribute__((ext_vector_type(2)));
TTM_HD void load_pair(const double* p, double& a, double& b) {
    const ttm_pair_t q = *(const ttm_pair_t*)p;
    a = q.x; b = q.y;
}
TTM_HD void store_pair(double* p, double a, double b) {
    ttm_pair_t q = {a, b};
    *(ttm_pair_t*)p = q;
}
#else
TTM_HD void load_pair(const double* p, double& a, double& b) { a = p[0]; b = p[1]; }
TTM_HD void store_pair(double* p, double a, double b) { p[0] = a; p[1] = b; }
#endif

// element access that also works for plain double
TTM_HD double elem(double a, int) { return a; }
template <int N> TTM_HD double elem(const VecD<N>& a, int i) { return a.v[i]; }
TTM_HD void set_elem(double& a, int, double x) { a = x; }
template <int N> TTM_HD void set_elem(VecD<N>& a, int i, double x) { a.v[i] = x; }

#define TTM_VEC_BINOP(op)                                                                             \
    template <int N> TTM_HD VecD<N> operator op(const VecD<N>& a, const VecD<N>& b) {                  \
        VecD<N> r;                                                                                     \
        _Pragma("unroll") for (int i = 0; i < N; ++i) r.v[i] = a.v[i] op b.v[i];                       \
        return r;                                                                                      \
    }                                                                                                  \
    template <int N> TTM_HD VecD<N> operator op(const VecD<N>& a, double b) {                          \
        VecD<N> r;                                                                                     \
        _Pragma("unroll") for (int i = 0; i < N; ++i) r.v[i] = a.v[i] op b;                            \
        return r;                                                                                      \
    }                                                                                                  \
    template <int N> TTM_HD VecD<N> operator op(double a, const VecD<N>& b) {                          \
        VecD<N> r;                                                                                     \
        _Pragma("unroll") for (int i = 0; i < N; ++i) r.v[i] = a op b.v[i];                            \
        return r;                                                                                      \
    }
TTM_VEC_BINOP(+)
TTM_VEC_BINOP(-)
TTM_VEC_BINOP(*)
#undef TTM_VEC_BINOP

template <int N> TTM_HD VecD<N> operator-(const VecD<N>& a) {
    VecD<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = -a.v[i];
    return r;
}
template <int N> TTM_HD VecD<N>& operator+=(VecD<N>& a, const VecD<N>& b) {
#pragma unroll
    for (int i = 0; i < N; ++i) a.v[i] += b.v[i];
    return a;
}
template <int N> TTM_HD VecD<N>& operator*=(VecD<N>& a, const VecD<N>& b) {
#pragma unroll
    for (int i = 0; i < N; ++i) a.v[i] *= b.v[i];
    return a;
}
template <int N> TTM_HD VecD<N>& operator*=(VecD<N>& a, double b) {
#pragma unroll
    for (int i = 0; i < N; ++i) a.v[i] *= b;
    return a;
}

// ---- elementwise functions, same names for double and VecD<N> ---------------------------------------

TTM_HD double vfma(double a, double b, double c) { return fma(a, b, c); }
TTM_HD double vmin(double a, double b) { return fmin(a, b); }
TTM_HD double vmax(double a, double b) { return fmax(a, b); }
TTM_HD double vabs(double a) { return fabs(a); }
TTM_HD double vcopysign(double a, double s) { return copysign(a, s); }
TTM_HD double vrint(double a) { return rint(a); }
TTM_HD int vtoint(double a) { return (int)a; }
TTM_HD double vfromint(int a) { return (double)a; }
TTM_HD double vldexp(double a, int e) { return ldexp(a, e); }
TTM_HD double vnan_to(double x, double probe, double res) { return (probe != probe) ? probe : res; }   // NaN in -> NaN out
TTM_HD double vselect_lt0(double c, double a, double b) { return c < 0.0 ? a : b; }

#define TTM_VEC_FN3(name)                                                                                \
    template <int N> TTM_HD VecD<N> name(const VecD<N>& a, const VecD<N>& b, const VecD<N>& c) {          \
        VecD<N> r;                                                                                        \
        _Pragma("unroll") for (int i = 0; i < N; ++i) r.v[i] = name(a.v[i], b.v[i], c.v[i]);             \
        return r;                                                                                         \
    }
TTM_VEC_FN3(vfma)
TTM_VEC_FN3(vnan_to)
TTM_VEC_FN3(vselect_lt0)
#undef TTM_VEC_FN3

// fma with uniform operands (kept scalar so that they stay SGPR operands)
template <int N> TTM_HD VecD<N> vfma(double a, const VecD<N>& b, const VecD<N>& c) {
    VecD<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = fma(a, b.v[i], c.v[i]);
    return r;
}
template <int N> TTM_HD VecD<N> vfma(const VecD<N>& a, double b, const VecD<N>& c) {
    VecD<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = fma(a.v[i], b, c.v[i]);
    return r;
}
template <int N> TTM_HD VecD<N> vfma(const VecD<N>& a, const VecD<N>& b, double c) {
    VecD<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = fma(a.v[i], b.v[i], c);
    return r;
}
template <int N> TTM_HD VecD<N> vfma(double a, const VecD<N>& b, double c) {
    VecD<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = fma(a, b.v[i], c);
    return r;
}
template <int N> TTM_HD VecD<N> vfma(const VecD<N>& a, double b, double c) {
    VecD<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = fma(a.v[i], b, c);
    return r;
}

#define TTM_VEC_FN2(name)                                                                                \
    template <int N> TTM_HD VecD<N> name(const VecD<N>& a, const VecD<N>& b) {                            \
        VecD<N> r;                                                                                        \
        _Pragma("unroll") for (int i = 0; i < N; ++i) r.v[i] = name(a.v[i], b.v[i]);                     \
        return r;                                                                                         \
    }                                                                                                     \
    template <int N> TTM_HD VecD<N> name(const VecD<N>& a, double b) {                                    \
        VecD<N> r;                                                                                        \
        _Pragma("unroll") for (int i = 0; i < N; ++i) r.v[i] = name(a.v[i], b);                          \
        return r;                                                                                         \
    }
TTM_VEC_FN2(vmin)
TTM_VEC_FN2(vmax)
TTM_VEC_FN2(vcopysign)
#undef TTM_VEC_FN2

#define TTM_VEC_FN1(name)                                                                                \
    template <int N> TTM_HD VecD<N> name(const VecD<N>& a) {                                              \
        VecD<N> r;                                                                                        \
        _Pragma("unroll") for (int i = 0; i < N; ++i) r.v[i] = name(a.v[i]);                             \
        return r;                                                                                         \
    }
TTM_VEC_FN1(vabs)
TTM_VEC_FN1(vrint)
#undef TTM_VEC_FN1

template <int N> TTM_HD VecI<N> vtoint(const VecD<N>& a) {
    VecI<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = (int)a.v[i];
    return r;
}
template <int N> TTM_HD VecD<N> vfromint(const VecI<N>& a) {
    VecD<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = (double)a.v[i];
    return r;
}
template <int N> TTM_HD VecD<N> vldexp(const VecD<N>& a, const VecI<N>& e) {
    VecD<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = ldexp(a.v[i], e.v[i]);
    return r;
}

// table gather: base[i] per element
TTM_HD double vgather(const double* base, int i) { return base[i]; }
template <int N> TTM_HD VecD<N> vgather(const double* base, const VecI<N>& i) {
    VecD<N> r;
#pragma unroll
    for (int e = 0; e < N; ++e) r.v[e] = base[i.v[e]];
    return r;
}

// integer vector type matching a value type
template <class R> struct int_of { typedef int type; };
template <int N> struct int_of<VecD<N>> { typedef VecI<N> type; };
TTM_HD int ielem(int a, int) { return a; }
template <int N> TTM_HD int ielem(const VecI<N>& a, int i) { return a.v[i]; }

}  // namespace ttm
