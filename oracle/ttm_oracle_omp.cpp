// CPU ORACLE (C++ / OpenMP leg) - TEST INFRASTRUCTURE ONLY.
//
// A compiled, multi-threaded restatement of the two reference routines BASELINE.json's metric times - the forward
// map (TM:2391-2437 `map` -> TM:2439-2567 `s`, separable branch TM:2554-2558) and the table ("alternate") inverse
// (TM:3639-3796 `inverse_map` k-loop -> TM:3987-4084 `vectorized_root_search_alternate`) - for separable maps of the
// Hermite-function family (the BASELINE configurations C2b, C3, C5).  TM = /root/reference/transport_map.py.
// It follows the reference's algorithm step by step (every term evaluated on its own, Clenshaw evaluation of the
// polynomial factor as np.polynomial.hermite_e.hermeval does, libm erf / exp, a fresh 1001-point table per component
// and per call, np.searchsorted-left + scipy.interpolate.interp1d slope form), not the engine's restructured
// evaluation (folded coefficients, splines, bucket index).  Samples are row-major N x d as in the reference.
//
// Only bench.py's `cpu_baseline` leg and tests/ may load this library (oracle/omp.py), as the timed CPU baseline and
// as a checker.  It is pinned by tests/test_oracle_omp.py against the reference's outputs in tests/golden/ and against
// the NumPy oracle.  The product never links or loads it.
//
// Third-party algorithm restated: NumPy 2.2.6 `hermeval` (Clenshaw recurrence for HermiteE series).

#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

enum { F_CONST = 0, F_POLY = 1, F_HF = 2, F_LET = 3, F_RET = 4, F_RBF = 5, F_IRBF = 6 };

struct Prog {
    int D, d, skip;
    const int32_t* tstart;   // 2 D + 1: [2k] first nonmonotone term of component k, [2k+1] first monotone term, [2k+2] end
    const int32_t* foff;     // nterms + 1: factor ranges
    const int32_t* fi;       // 3 per factor: kind, variable, order
    const double* fd;        // 2 per factor: {Hermite-function constant a_n, -} or {centre, scale}
    const double* coef;      // one per term
};

// np.polynomial.hermite_e.hermeval(x, [0]*n + [top]) (Clenshaw), TM:1099-1122 builds exactly that coefficient vector
inline double hermeval_top(int n, double top, double x) {
    if (n == 0) return top;
    if (n == 1) return 0.0 + top * x;
    int nd = n + 1;
    double c0 = 0.0, c1 = top;
    for (int i = 3; i <= n + 1; ++i) {
        const double tmp = c0;
        nd -= 1;
        c0 = 0.0 - c1 * (double)(nd - 1);
        c1 = tmp + c1 * x;
    }
    return c0 + c1 * x;
}

// one factor at the value xv of its variable (TM:905-1026 special terms, TM:1096-1122 polynomial factors)
inline double factor(int kind, int order, double p0, double p1, double xv) {
    switch (kind) {
        case F_CONST: return 1.0;
        case F_POLY: return hermeval_top(order, 1.0, xv);
        case F_HF: return hermeval_top(order, p0, xv) * exp(-(xv * xv) / 4.0);
        case F_LET: {
            const double u = (xv - p0) / (sqrt(2.0) * p1);
            return ((xv - p0) * (1.0 - erf(u)) - p1 * sqrt(2.0 / M_PI) * exp(-(u * u))) / 2.0;
        }
        case F_RET: {
            const double u = (xv - p0) / (sqrt(2.0) * p1);
            return ((xv - p0) * (1.0 + erf(u)) + p1 * sqrt(2.0 / M_PI) * exp(-(u * u))) / 2.0;
        }
        case F_RBF: {
            const double u = (xv - p0) / p1;
            return exp(-(u * u) / 2.0) / (p1 * sqrt(2.0 * M_PI));
        }
        default: return (1.0 + erf((xv - p0) / (sqrt(2.0) * p1))) / 2.0;
    }
}

// sum_t coef_t prod_f factor over the terms [t0, t1) on one sample row x
inline double term_sum(const Prog& P, int t0, int t1, const double* x) {
    double s = 0.0;
    for (int t = t0; t < t1; ++t) {
        double v = 1.0;
        for (int f = P.foff[t]; f < P.foff[t + 1]; ++f)
            v *= factor(P.fi[3 * f], P.fi[3 * f + 2], P.fd[2 * f], P.fd[2 * f + 1], x[P.fi[3 * f + 1]]);
        s += P.coef[t] * v;
    }
    return s;
}

}  // namespace

extern "C" {

int ttmo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// Z[n, k] = S_k(x_n), X standardised row-major N x d, Z row-major N x D  (TM:2391-2437, 2554-2558)
int ttmo_forward(int D, int d, int skip, const int32_t* tstart, const int32_t* foff, const int32_t* fi, const double* fd,
                 const double* coef, const double* X, int64_t N, double* Z, int threads) {
    const Prog P{D, d, skip, tstart, foff, fi, fd, coef};
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
    for (int64_t n = 0; n < N; ++n) {
        const double* x = X + n * d;
        for (int k = 0; k < D; ++k)
            Z[n * D + k] = term_sum(P, tstart[2 * k], tstart[2 * k + 1], x) + term_sum(P, tstart[2 * k + 1], tstart[2 * k + 2], x);
    }
    return 0;
}

// Table inverse of components [k0, D): X (row-major N x d) holds the conditioning columns on entry and receives
// column skip + k of every inverted component; Z row-major N x (D - k0).  Per component, as TM:4039-4082: tabulate the
// monotone part at `resolution` points of [-start_distance, start_distance] with all other columns zero, subtract the
// nonmonotone offset from z_k, clip to the table's range, invert by linear interpolation (interp1d: stable sort,
// searchsorted-left clipped to [1, T-1], slope form).
int ttmo_inverse_table(int D, int d, int skip, const int32_t* tstart, const int32_t* foff, const int32_t* fi, const double* fd,
                       const double* coef, const double* Z, int k0, double* X, int64_t N, int truncate, int resolution,
                       double start_distance, int threads) {
    const Prog P{D, d, skip, tstart, foff, fi, fd, coef};
    const int T = resolution, ncomp = D - k0;
    if (T < 2 || ncomp < 1) return 1;
    std::vector<double> pts(T), tabx((size_t)ncomp * T), taby((size_t)ncomp * T), tmin(ncomp), tmax(ncomp);
    const double step = (2.0 * start_distance) / (double)(T - 1);          // np.linspace
    for (int i = 0; i < T; ++i) pts[i] = (double)i * step + (-start_distance);
    pts[T - 1] = start_distance;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (int c = 0; c < ncomp; ++c) {
        const int k = k0 + c;
        std::vector<double> x(d, 0.0), out(T);
        std::vector<int> idx(T);
        for (int i = 0; i < T; ++i) {
            x[skip + k] = pts[i];
            out[i] = term_sum(P, tstart[2 * k + 1], tstart[2 * k + 2], x.data());
            idx[i] = i;
        }
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return out[a] < out[b]; });
        double lo = out[0], hi = out[0];
        for (int i = 0; i < T; ++i) {
            tabx[(size_t)c * T + i] = out[idx[i]];
            taby[(size_t)c * T + i] = pts[idx[i]];
            lo = out[i] < lo ? out[i] : lo;
            hi = out[i] > hi ? out[i] : hi;
        }
        tmin[c] = lo;
        tmax[c] = hi;
    }
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
    for (int64_t n = 0; n < N; ++n) {
        double* x = X + n * d;
        for (int c = 0; c < ncomp; ++c) {
            const int k = k0 + c;
            const double* tx = tabx.data() + (size_t)c * T;
            const double* ty = taby.data() + (size_t)c * T;
            double target = -term_sum(P, tstart[2 * k], tstart[2 * k + 1], x) + Z[n * ncomp + c];
            if (truncate) {
                if (target < tmin[c]) target = tmin[c];
                if (target > tmax[c]) target = tmax[c];
            }
            int64_t i = std::lower_bound(tx, tx + T, target) - tx;
            i = i < 1 ? 1 : (i > T - 1 ? T - 1 : i);
            const double slope = (ty[i] - ty[i - 1]) / (tx[i] - tx[i - 1]);
            x[skip + k] = slope * (target - tx[i - 1]) + ty[i - 1];
        }
    }
    return 0;
}

}  // extern "C"
