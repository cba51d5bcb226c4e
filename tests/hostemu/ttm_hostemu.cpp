// ttm_hostemu.cpp - TEST INFRASTRUCTURE ONLY (a test double of libttm.so).
//
// Exports the C ABI of include/ttm.h with HOST pointers, implemented as plain
// loops around the very same per-sample bodies (csrc/ttm_eval.h) the HIP kernels
// run.  It lets the host-side logic (term-table compiler, the transport_map
// class, sample sharding) and the kernel bodies be checked against the oracle
// on a machine without a GPU.  It is NOT a fallback of the product: nothing
// under triangular_transport_toolbox_amd/ loads it; only tests inject it
// (tests/hostemu/emu.py), and the GPU-specific parts (LDS staging, grid loops,
// reductions, launch planning) are exercised only by the `-m gpu` tests.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../triangular_transport_toolbox_amd/csrc/ttm_eval.h"

using namespace ttm;

namespace {

struct VecSlots {
    double* base;
    double get(int i) const { return base[i]; }
    void set(int i, double v) { base[i] = v; }
};
struct VecAcc {
    double* base;
    void add(int i, double v) { base[i] += v; }
};
struct XSoA {
    const double* X;
    int64_t ld, n;
    double operator()(int var) const { return X[(int64_t)var * ld + n]; }
};
struct XFake {
    int kc;
    double t;
    double operator()(int var) const { return var == kc ? t : 0.0; }
};

Prog make_prog(const ttm_program* p) {
    Prog g;
    g.qx = p->quad_x; g.qw = p->quad_w; g.Q = p->Q; g.family = p->family; g.mono = p->monotonicity;
    g.rect = p->rectifier; g.delta = p->delta;
    return g;
}

Comp comp_of(const ttm_program* p, int k, const double* coef_k) {
    return make_comp(p->itab + p->h_comp_off[k], p->dpar + p->h_dpar_off[k], coef_k);
}

}  // namespace

extern "C" {

const char* ttm_last_error_string(void) { return "hostemu"; }
int ttm_version(void) { return TTM_VERSION; }
int ttm_device_count(int* count) { if (count) *count = 0; return TTM_E_HIP; }

int64_t ttm_colstats_work_size(int64_t, int32_t d) { return d; }

int ttm_colstats(const double* Xrow, int64_t N, int32_t d, double* mean, double* sd, double*, void*) {
    for (int j = 0; j < d; ++j) {
        double s = 0.0;
        for (int64_t n = 0; n < N; ++n) s += Xrow[n * d + j];
        mean[j] = s / (double)N;
        double q = 0.0;
        for (int64_t n = 0; n < N; ++n) { const double v = Xrow[n * d + j] - mean[j]; q += v * v; }
        sd[j] = sqrt(q / (double)N);
    }
    return 0;
}

int ttm_import(const double* Xrow, int64_t N, int32_t d, const double* mean, const double* sd, double* Xsoa, int64_t ldx, void*) {
    for (int64_t n = 0; n < N; ++n)
        for (int j = 0; j < d; ++j) {
            double v = Xrow[n * d + j];
            if (mean) v = (v - mean[j]) / sd[j];
            Xsoa[(int64_t)j * ldx + n] = v;
        }
    return 0;
}

int ttm_export(const double* Xsoa, int64_t ldx, int64_t N, int32_t j0, int32_t dout, const double* mean, const double* sd,
               double* Xrow, void*) {
    for (int64_t n = 0; n < N; ++n)
        for (int j = 0; j < dout; ++j) {
            double v = Xsoa[(int64_t)(j0 + j) * ldx + n];
            if (mean) v = v * sd[j0 + j] + mean[j0 + j];
            Xrow[n * dout + j] = v;
        }
    return 0;
}

int ttm_forward(const ttm_program* p, const double* coef, const double* X, int64_t ldx, int64_t N, int32_t k0, int32_t k1,
                double* Z, int64_t ldz, double* logdet, const double* sigma, double* sumsq, void*) {
    const Prog g = make_prog(p);
    std::vector<double> scr(4096);
    for (int64_t n = 0; n < N; ++n) {
        XSoA x{X, ldx, n};
        double ld = 0.0, ss = 0.0;
        for (int k = k0; k < k1; ++k) {
            const Comp c = comp_of(p, k, coef + p->h_coef_off[k]);
            VecSlots w{scr.data()};
            double S, dS;
            if (logdet) {
                sample_forward<true>(c, g, x, w, Z || sumsq, S, dS);
                ld += log(sigma ? dS / sigma[k - k0] : dS);
            } else {
                sample_forward<false>(c, g, x, w, true, S, dS);
            }
            if (Z) Z[(int64_t)(k - k0) * ldz + n] = S;
            ss = fma(S, S, ss);
        }
        if (logdet) logdet[n] = ld;
        if (sumsq) sumsq[n] = ss;
    }
    return 0;
}

int ttm_basis(const ttm_program* p, int32_t k, int32_t which, const double* X, int64_t ldx, int64_t N, double* out, int64_t ldo, void*) {
    const Prog g = make_prog(p);
    const Comp c = comp_of(p, k, nullptr);
    for (int64_t n = 0; n < N; ++n) {
        XSoA x{X, ldx, n};
        sample_basis(c, g, which, x, [&](int i, double v) { out[(int64_t)i * ldo + n] = v; });
    }
    return 0;
}

int64_t ttm_reduce_work_size(int32_t nout) { return nout > 0 ? nout : 1; }

int ttm_objective(const ttm_program* p, int32_t k, const double* coef_k, const double* X, int64_t ldx, int64_t N, double*,
                  double* out, void*) {
    const Prog g = make_prog(p);
    const Comp c = comp_of(p, k, coef_k);
    const int nb1 = c.nB + 1;
    std::vector<double> scr(3 * nb1 + 8);
    const int nacc = g.mono == TTM_MONO_SEPARABLE ? 1 + c.n_mon : 1 + c.n_nm + c.n_mon;
    for (int i = 0; i < nacc; ++i) out[i] = 0.0;
    VecAcc acc{out};
    for (int64_t n = 0; n < N; ++n) {
        XSoA x{X, ldx, n};
        VecSlots w{scr.data()}, Bv{scr.data() + nb1}, I{scr.data() + 2 * nb1};
        if (g.mono == TTM_MONO_SEPARABLE) sample_objective_sep(c, g, x, w, acc);
        else sample_objective_int(c, g, x, w, Bv, I, acc);
    }
    return 0;
}

int ttm_gram(const ttm_program* p, int32_t k, const double* X, int64_t ldx, int64_t N, double*, double* out, void*) {
    const Prog g = make_prog(p);
    const Comp c = comp_of(p, k, nullptr);
    const int m = c.n_nm + c.n_mon;
    std::vector<double> row(m);
    for (int i = 0; i < m * m; ++i) out[i] = 0.0;
    for (int64_t n = 0; n < N; ++n) {
        XSoA x{X, ldx, n};
        sample_basis(c, g, 0, x, [&](int i, double v) { row[i] = v; });
        sample_basis(c, g, 1, x, [&](int i, double v) { row[c.n_nm + i] = v; });
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < m; ++j) out[i * m + j] = fma(row[i], row[j], out[i * m + j]);
    }
    return 0;
}

int ttm_inverse_table_build(const ttm_program* p, const double* coef, int32_t k0, int32_t k1, const double* pts, int32_t T,
                            double* out, void*) {
    const Prog g = make_prog(p);
    for (int k = k0; k < k1; ++k) {
        const Comp c = comp_of(p, k, coef + p->h_coef_off[k]);
        std::vector<double> scr(c.nB + 2);
        for (int i = 0; i < T; ++i) {
            XFake x{c.kc, pts[i]};
            VecSlots w{scr.data()};
            mon_weights(c, g.family, x, w);
            double v, dv;
            g_eval<false>(c, g.family, pts[i], w, v, dv);
            out[(int64_t)(k - k0) * T + i] = v;
        }
    }
    return 0;
}

int ttm_inverse_table(const ttm_program* p, const double* coef, int32_t k0, int32_t k1, const double* Z, int64_t ldz, double* X,
                      int64_t ldx, int64_t N, const double* tab_x, const double* tab_y, int32_t T, const double* tmin,
                      const double* tmax, int32_t truncate, void*) {
    const Prog g = make_prog(p);
    for (int k = k0; k < k1; ++k) {
        const Comp c = comp_of(p, k, coef + p->h_coef_off[k]);
        const double* xs = tab_x + (int64_t)(k - k0) * T;
        const double* ys = tab_y + (int64_t)(k - k0) * T;
        for (int64_t n = 0; n < N; ++n) {
            XSoA x{X, ldx, n};
            const double off = nonmon_sum(c, g.family, x);
            double target = -off + Z[(int64_t)(k - k0) * ldz + n];
            if (truncate) {
                if (target < tmin[k - k0]) target = tmin[k - k0];
                if (target > tmax[k - k0]) target = tmax[k - k0];
            }
            X[(int64_t)c.kc * ldx + n] = table_lookup(xs, ys, T, target);
        }
    }
    return 0;
}

int ttm_inverse_bisect(const ttm_program* p, const double* coef, int32_t k0, int32_t k1, const double* Z, int64_t ldz, double* X,
                       int64_t ldx, int64_t N, int32_t* iters, const int32_t* cap, void*) {
    const Prog g = make_prog(p);
    std::vector<double> scr(4096);
    for (int64_t n = 0; n < N; ++n) {
        XSoA x{X, ldx, n};
        for (int k = k0; k < k1; ++k) {
            const Comp c = comp_of(p, k, coef + p->h_coef_off[k]);
            VecSlots w{scr.data()};
            const double off = nonmon_sum(c, g.family, x);
            mon_weights(c, g.family, x, w);
            int it = 0;
            X[(int64_t)c.kc * ldx + n] = sample_bisect(c, g, off, Z[(int64_t)(k - k0) * ldz + n], w, cap ? cap[k - k0] : -1, it);
            if (it > iters[k - k0]) iters[k - k0] = it;
        }
    }
    return 0;
}

}  // extern "C"
