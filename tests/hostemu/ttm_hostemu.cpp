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

#include <algorithm>
#include <vector>

#include "../../triangular_transport_toolbox_amd/csrc/ttm_eval.h"
#include "../../triangular_transport_toolbox_amd/csrc/ttm_dense.h"
#include "../../triangular_transport_toolbox_amd/csrc/ttm_xprog.h"
#include "../../triangular_transport_toolbox_amd/csrc/ttm_uform.h"
#include "../../triangular_transport_toolbox_amd/csrc/ttm_rng.h"

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
    double get(int var) const { return var == kc ? t : 0.0; }
    void get_e(int var, double& xv, double& e) const { xv = get(var); e = fast_exp(-0.25 * (xv * xv)); }
};

const double kErfTab[TTM_ERF_TABLE_LEN] = { TTM_ERF_TABLE_VALUES };
const double kExpQTab[TTM_EXPQ_TABLE_LEN] = { TTM_EXPQ_TABLE_VALUES };

Prog make_prog(const ttm_program* p) {
    Prog g;
    g.qx = p->quad_x; g.qw = p->quad_w; g.erf_tab = kErfTab; g.Q = p->Q; g.family = p->family; g.mono = p->monotonicity;
    g.rect = p->rectifier; g.delta = p->delta;
    return g;
}

// component view; fold_k = folded coefficients of this component (NULL: fold here from coef_k)
struct HostComp {
    std::vector<double> fold;
    Comp c;
};

void comp_of(const ttm_program* p, int k, const double* coef_k, HostComp& h, const double* fold_k = nullptr) {
    const int* cb = p->itab + p->h_comp_off[k];
    const double* dp = p->dpar + p->h_dpar_off[k];
    if (!fold_k) {
        const int nf = cb[TTM_HDR_N_FOLD];
        h.fold.assign(nf, 0.0);
        if (coef_k) fold_coeffs(cb, p->ftab + p->h_ftab_off[k], dp, coef_k, h.fold.data(), 0, 1);
        fold_k = h.fold.data();
    }
    h.c = make_comp(cb, dp, coef_k, fold_k);
    const int* fb = p->ftab + p->h_ftab_off[k];
    h.c.fslot = fb + cb[TTM_HDR_OFF_FSLOT];
    h.c.fsrc = fb + cb[TTM_HDR_OFF_FSRC];
}

// same dispatch rule as the library (csrc/ttm_int.hip: ttm_int::usable): integrated maps whose components all have a dense
// B set go through the monomial-form bodies of csrc/ttm_dense.h
bool int_dense(const ttm_program* p, int ka, int kb, DenseClass& cls) {
    if (getenv("TTM_INT_DENSE") && atoi(getenv("TTM_INT_DENSE")) == 0) return false;
    if (p->monotonicity != TTM_MONO_INTEGRATED || p->family < 0 || p->family > 5) return false;
    return dense_range_class(p->h_complex, ka, kb, cls);
}

// same dispatch rule as the library (csrc/ttm_int.hip: xprog_rows): the X-program kernels when every component of the range has one
bool all_xprog(const ttm_program* p, int ka, int kb) {
    if (getenv("TTM_INT_XPROG") && atoi(getenv("TTM_INT_XPROG")) == 0) return false;
    for (int k = ka; k < kb; ++k)
        if (!(p->h_complex[k] & 16)) return false;
    return true;
}

// same dispatch rule as the library: planned-cache fast path when every component of the range is simple
bool all_fast(const ttm_program* p, int ka, int kb) {
    if (getenv("TTM_NO_PLAN")) return false;
    for (int k = ka; k < kb; ++k)
        if (p->h_complex[k] & 1) return false;
    return true;
}

template <int FAM, class R, class XA>
void forward_plan(const ttm_program* p, const Prog& g, const double* fold, const XA& xa, int k0, int k1, bool want_ld,
                  bool want_val, R* S_out, R& ld, R& ss, const double* sigma) {
    double cbuf[8 * lanes_of<R>::value];
    PlanCache<XA, R> x(xa, CacheStore<R>{cbuf, 1});
    if (k0 > 0) x.warm(p->fints + p->fdesc[k0 * TTM_FDESC_LEN + TTM_FD_PLAN_OFF]);
    ld = R(0.0); ss = R(0.0);
    for (int k = k0; k < k1; ++k) {
        const int* fd = p->fdesc + k * TTM_FDESC_LEN;
        const FastComp f = make_fast(fd, p->fints, fold, 0);
        const R xk = xa(fd[TTM_FD_KC]);
        R S, dS;
        if (want_ld) sample_forward_fast<-1, FAM, true>(f, g, xk, x, want_val, S, dS);
        else sample_forward_fast<-1, FAM, false>(f, g, xk, x, true, S, dS);
        x.put(fd[TTM_FD_KC_SLOT], xk);
        if (want_ld) ld += fast_log(sigma ? fast_div(dS, sigma[k - k0]) : dS);
        S_out[k - k0] = S;
        ss = vfma(S, S, ss);
    }
}

int64_t fold_base_size_(const ttm_program* p) { return ((int64_t)p->h_fold_off[p->D] + 8 + 1) & ~(int64_t)1; }

// U-form forward of one "thread" (R = double or VecD<N>) through components [k0,k1)
template <class R, class XA>
void forward_u(const ttm_program* p, const double* fold, const XA& xa, int k0, int k1, bool want_ld, bool want_val,
               R* S_out, R& ld, R& ss, const double* sigma) {
    const double* U = fold + fold_base_size_(p);
    double cbuf[8 * lanes_of<R>::value];
    PlanCache<XA, R> x(xa, CacheStore<R>{cbuf, 1});
    if (k0 > 0) x.warm(p->ucomp + TTM_UC_STATE(p->D, k0));
    ld = R(0.0); ss = R(0.0);
    const bool fixed = getenv("TTM_EMU_U_FIXED") != nullptr;      // test knob: the fixed-degree instantiation
    for (int k = k0; k < k1; ++k) {
        const int* uc = p->ucomp + k * TTM_UC_LEN;
        const R xk = xa(uc[TTM_UC_KC]);
        R S, dS;
        if (fixed) {
            if (want_ld) u_component<TTM_U_PMAX, TTM_U_PMAX, true>(uc, p->ugrp, U, U + uc[TTM_UC_TAB_OFF], xk, x, want_val, S, dS);
            else u_component<TTM_U_PMAX, TTM_U_PMAX, false>(uc, p->ugrp, U, U + uc[TTM_UC_TAB_OFF], xk, x, true, S, dS);
        } else {
            if (want_ld) u_component<-1, -1, true>(uc, p->ugrp, U, U + uc[TTM_UC_TAB_OFF], xk, x, want_val, S, dS);
            else u_component<-1, -1, false>(uc, p->ugrp, U, U + uc[TTM_UC_TAB_OFF], xk, x, true, S, dS);
        }
        if (want_ld) ld += fast_log(sigma ? fast_div(dS, sigma[k - k0]) : dS);
        S_out[k - k0] = S;
        ss = vfma(S, S, ss);
    }
}

// hot-record forward (what k_forward_hl evaluates) of one "thread"
template <int NG, int CLS, class R, class XA>
void forward_h(const ttm_program* p, const double* fold, const XA& xa, int k0, int k1, bool want_ld, bool want_val,
               R* S_out, R& ld, R& ss, const double* sigma) {
    constexpr int DB = CLS == 1 ? 3 : (CLS == 2 ? 5 : 7), DA = CLS == 1 ? 1 : (CLS == 2 ? 5 : 7), GS = CLS == 1 ? 8 : (CLS == 2 ? 16 : 24);
    const double* U = fold + fold_base_size_(p);
    const int ways = (p->plan_ways < 1 || p->plan_ways > TTM_PLAN_WAYS) ? TTM_PLAN_WAYS : p->plan_ways;
    double cbuf[2 * TTM_PLAN_WAYS * lanes_of<R>::value];
    for (auto& c : cbuf) c = 0.0;
    (void)ways;
    CacheStore<R> st{cbuf, 1};               // (no exp table: as k_forward_hl)
    PlanCache<XA, R> x(xa, st);
    if (k0 > 0) x.warm(p->ucomp + TTM_UC_STATE(p->D, k0));
    ld = R(0.0); ss = R(0.0);
    const int hs = TTM_H_HDR + p->u_h_ng * GS;
    for (int k = k0; k < k1; ++k) {
        const double* rec = U + p->u_h_off + (int64_t)k * hs;
        const int kc = ((const int*)rec)[3];
        const int tab_off = ((const int*)rec)[12];
        const R xk = xa(kc);
        R S, dS;
        if (want_ld) h_component<NG, DB, DA, GS, true>(rec, U + tab_off, xk, st, want_val, S, dS);
        else h_component<NG, DB, DA, GS, false>(rec, U + tab_off, xk, st, true, S, dS);
        if (want_ld) ld += fast_log(sigma ? fast_div(dS, sigma[k - k0]) : dS);
        S_out[k - k0] = S;
        ss = vfma(S, S, ss);
    }
}

template <class R, class XA>
bool forward_h_dispatch(const ttm_program* p, const double* fold, const XA& xa, int k0, int k1, bool want_ld, bool want_val,
                        R* S_out, R& ld, R& ss, const double* sigma) {
#define TTM_FH(NGV, C) forward_h<NGV, C, R>(p, fold, xa, k0, k1, want_ld, want_val, S_out, ld, ss, sigma); return true
    switch (p->u_h_ng * 10 + p->u_h_cls) {
        case 11: TTM_FH(1, 1); case 12: TTM_FH(1, 2); case 13: TTM_FH(1, 3);
        case 21: TTM_FH(2, 1); case 22: TTM_FH(2, 2); case 23: TTM_FH(2, 3);
        case 31: TTM_FH(3, 1); case 32: TTM_FH(3, 2); case 33: TTM_FH(3, 3);
        case 41: TTM_FH(4, 1); case 42: TTM_FH(4, 2); case 43: TTM_FH(4, 3);
        default: return false;
    }
#undef TTM_FH
}

}  // namespace

extern "C" {

const char* ttm_last_error_string(void) { return "hostemu"; }
int ttm_set_error_string(const char*) { return TTM_OK; }
int ttm_version(void) { return TTM_VERSION; }
const char* ttm_last_kernel(void) { return "hostemu"; }
// options of the test double: the three it knows live in its environment variables (read per call)
int ttm_set_option(const char* name, int32_t value) {
    if (!name) return TTM_E_ARG;
    const char* env = !strcmp(name, "no_plan") ? "TTM_NO_PLAN" : !strcmp(name, "no_uform") ? "TTM_NO_UFORM" :
                      !strcmp(name, "u_no_hot") ? "TTM_EMU_NO_HOT" : nullptr;
    if (env) { if (value) setenv(env, "1", 1); else unsetenv(env); }
    return TTM_OK;                                  // (the other options select device kernel variants: nothing to do)
}
int ttm_reset_options(void) { unsetenv("TTM_NO_PLAN"); unsetenv("TTM_NO_UFORM"); unsetenv("TTM_EMU_NO_HOT"); return TTM_OK; }

// the collective of the path (include/ttm.h "C1"): the test double has no RCCL; the harness registers a callback that
// performs the reduction on the host buffer (tests: torch.distributed over gloo), so the class under test goes
// through the same ttm_comm_* / ttm_allreduce_* calls as on the GPU
typedef void (*ttm_hostemu_allreduce_cb)(void* buf, int64_t count, int32_t is_f64, int32_t op);
static ttm_hostemu_allreduce_cb g_allreduce_cb = nullptr;
static int64_t g_allreduce_calls = 0;
void ttm_hostemu_set_allreduce(ttm_hostemu_allreduce_cb cb) { g_allreduce_cb = cb; }
int64_t ttm_hostemu_allreduce_calls(void) { return g_allreduce_calls; }
const char* ttm_comm_last_error(void) { return g_allreduce_cb ? "" : "hostemu: no all-reduce callback registered"; }
int ttm_comm_unique_id(void* id128) { if (!id128) return TTM_E_ARG; memset(id128, 0, 128); return TTM_OK; }
int ttm_comm_create(const void* id128, int32_t rank, int32_t nranks, ttm_comm** out) {
    if (!id128 || !out || nranks < 1 || rank < 0 || rank >= nranks) return TTM_E_ARG;
    *out = (ttm_comm*)new int32_t[2]{rank, nranks};
    return TTM_OK;
}
int ttm_comm_destroy(ttm_comm* c) { delete[] (int32_t*)c; return TTM_OK; }
int ttm_comm_size(const ttm_comm* c, int32_t* rank, int32_t* nranks) {
    if (!c) return TTM_E_ARG;
    if (rank) *rank = ((const int32_t*)c)[0];
    if (nranks) *nranks = ((const int32_t*)c)[1];
    return TTM_OK;
}
int ttm_allreduce_f64(ttm_comm* c, double* buf, int64_t count, int32_t op, void*) {
    if (!c || !buf || count < 1) return TTM_E_ARG;
    if (!g_allreduce_cb) return TTM_E_UNSUPPORTED;
    ++g_allreduce_calls;
    g_allreduce_cb(buf, count, 1, op);
    return TTM_OK;
}
int ttm_allreduce_i32(ttm_comm* c, int32_t* buf, int64_t count, int32_t op, void*) {
    if (!c || !buf || count < 1) return TTM_E_ARG;
    if (!g_allreduce_cb) return TTM_E_UNSUPPORTED;
    ++g_allreduce_calls;
    g_allreduce_cb(buf, count, 0, op);
    return TTM_OK;
}
int64_t ttm_program_sizeof(void) { return (int64_t)sizeof(ttm_program); }
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

int ttm_stream_synchronize(void*) { return TTM_OK; }
// (one launch for forward + inverse: a device matter - the double declines and the class makes the two calls)
int ttm_roundtrip(const ttm_program*, const double*, const double*, const double*, int64_t, int64_t, double*, int64_t, double*, int64_t, double*,
                  const double*, double*, const double*, int32_t, const double*, const double*, const double*, const int32_t*, int32_t, void*) {
    return TTM_E_UNSUPPORTED;
}

int ttm_colstats_cols(const double* Xcols, int64_t ld, int64_t N, int32_t d, double* mean, double* sd, double*, void*) {
    if (d > 8 || N > 131072) return TTM_E_UNSUPPORTED;            // (the device library's one-launch range)
    for (int j = 0; j < d; ++j) {
        const double* c = Xcols + (int64_t)j * ld;
        double s = 0.0;
        for (int64_t n = 0; n < N; ++n) s += c[n];
        mean[j] = s / (double)N;
        double q = 0.0;
        for (int64_t n = 0; n < N; ++n) { const double v = c[n] - mean[j]; q += v * v; }
        sd[j] = sqrt(q / (double)N);
    }
    return 0;
}

int ttm_standardize_cols(const double* Xcols, int64_t ld, int64_t N, int32_t d, const double* mean, const double* sd, double* Xs,
                         int64_t ldx, void*) {
    for (int j = 0; j < d; ++j)
        for (int64_t n = 0; n < N; ++n) Xs[(int64_t)j * ldx + n] = (Xcols[(int64_t)j * ld + n] - mean[j]) / sd[j];
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

int64_t ttm_select_work_size(int32_t) { return 8; }

int ttm_order_statistics(const double* col, int64_t N, const int64_t* ranks, int32_t nr, double* out, void*, void*) {
    std::vector<double> v(col, col + N);
    for (int j = 0; j < nr; ++j) {
        std::nth_element(v.begin(), v.begin() + ranks[j], v.end());
        out[j] = v[ranks[j]];
    }
    return 0;
}

// radix select over sharded columns: the device algorithm (eight passes over the order-preserving 64-bit keys, bin
// counts all-reduced per pass) on the host
int ttm_allreduce_i32(ttm_comm* c, int32_t* buf, int64_t count, int32_t op, void*);
int ttm_order_statistics_dist(const double* col, int64_t N, const int64_t* ranks, int32_t nr, double* out, void*, ttm_comm* comm,
                              void*) {
    if (!comm) return ttm_order_statistics(col, N, ranks, nr, out, nullptr, nullptr);
    auto key = [](double x) {
        unsigned long long u;
        memcpy(&u, &x, 8);
        return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
    };
    std::vector<unsigned long long> prefix(nr, 0ull);
    std::vector<long long> rank(ranks, ranks + nr);
    for (int shift = 56; shift >= 0; shift -= 8) {
        const unsigned long long himask = (shift == 56) ? 0ull : (~0ull << (shift + 8));
        std::vector<int32_t> hist((size_t)nr * 256, 0);
        for (int64_t n = 0; n < N; ++n) {
            const unsigned long long k = key(col[n]);
            for (int j = 0; j < nr; ++j)
                if (((k ^ prefix[j]) & himask) == 0ull) ++hist[(size_t)j * 256 + ((k >> shift) & 255ull)];
        }
        const int rc = ttm_allreduce_i32(comm, hist.data(), (int64_t)nr * 256, TTM_OP_SUM, nullptr);
        if (rc) return rc;
        for (int j = 0; j < nr; ++j) {
            long long before = 0;
            int t = 0;
            for (; t < 255; ++t) {
                if (rank[j] < before + hist[(size_t)j * 256 + t]) break;
                before += hist[(size_t)j * 256 + t];
            }
            rank[j] -= before;
            prefix[j] |= (unsigned long long)t << shift;
        }
    }
    for (int j = 0; j < nr; ++j) {
        const unsigned long long u = (prefix[j] >> 63) ? (prefix[j] & 0x7fffffffffffffffull) : ~prefix[j];
        memcpy(&out[j], &u, 8);
    }
    return 0;
}

static int64_t fold_base_size(const ttm_program* p) { return ((int64_t)p->h_fold_off[p->D] + 8 + 1) & ~(int64_t)1; }
static bool u_on(const ttm_program* p) { return p->u_enabled && !getenv("TTM_NO_UFORM"); }
static int plan_ways_of(const ttm_program* p) { return (p->plan_ways < 1 || p->plan_ways > TTM_PLAN_WAYS) ? TTM_PLAN_WAYS : p->plan_ways; }

int64_t ttm_fold_size(const ttm_program* p) { return fold_base_size(p) + (p->u_enabled ? p->u_size : 0); }
int64_t ttm_uform_offset(const ttm_program* p) { return p->u_enabled ? fold_base_size(p) : -1; }

int ttm_fold(const ttm_program* p, const double* coef, double* fold, void*) {
    for (int k = 0; k < p->D; ++k)
        fold_coeffs(p->itab + p->h_comp_off[k], p->ftab + p->h_ftab_off[k], p->dpar + p->h_dpar_off[k],
                    coef + p->h_coef_off[k], fold + p->h_fold_off[k], 0, 1);
    for (int k = 0; k < p->D; ++k) fold_st8(p->fdesc + k * TTM_FDESC_LEN, p->fints, fold + p->h_fold_off[k], 0, 1);
    if (p->u_enabled) {                       // U-form section (csrc/ttm_uform.h), same builder as the k_uform kernel
        double* U = fold + fold_base_size(p);
        std::vector<double> ybuf(TTM_U_NI_MAX * TTM_CHEB_N);
        for (int k = 0; k < p->D; ++k) {
            const int* uc = p->ucomp + k * TTM_UC_LEN;
            const int* fd = p->fdesc + k * TTM_FDESC_LEN;
            const double* foldk = fold + p->h_fold_off[k];
            const double* geo = p->ugeo + 2 * k;
            if (uc[TTM_UC_NI] > TTM_U_NI_MAX) return TTM_E_LIMIT;
            uform_build_groups(uc, p->ugrp, fd, p->umono, geo, foldk, U, 0, 1);
            if (p->u_h_cls > 0) uform_build_hot(uc, p->ugrp, U, p->u_h_off, p->u_h_cls, p->u_h_ng, k, foldk[0], 0, 1);
            double ev = 0.0, ed = 0.0;
            if (uc[TTM_UC_NI] > 0) {
                uform_spline_nodes(uc, fd, geo, foldk, ybuf.data(), 0, 1);
                uform_spline_fit(uc, fd, geo, foldk, ybuf.data(), U, 0, 1);
                uform_spline_verify(uc, fd, geo, foldk, U, 0, 1, ev, ed);
            }
            U[p->u_err_off + 2 * k] = (ev != ev) ? INFINITY : ev;
            U[p->u_err_off + 2 * k + 1] = (ed != ed) ? INFINITY : ed;
        }
    }
    return 0;
}

int ttm_fold_staged(const ttm_program* p, const double* h_coef, double* coef, double* fold, double* h_err, void* stream) {
    if (!p->u_enabled) return TTM_E_UNSUPPORTED;
    memcpy(coef, h_coef, sizeof(double) * (size_t)p->h_coef_off[p->D]);
    const int rc = ttm_fold(p, coef, fold, stream);
    if (!rc && h_err) memcpy(h_err, fold + fold_base_size(p) + p->u_err_off, sizeof(double) * 2 * (size_t)p->D);
    return rc;
}

int ttm_forward(const ttm_program* p, const double* coef, const double* fold, const double* X, int64_t ldx, int64_t N,
                int32_t k0, int32_t k1, double* Z, int64_t ldz, double* logdet, const double* sigma, double* sumsq, void*) {
    const Prog g = make_prog(p);
    if (p->monotonicity == TTM_MONO_SEPARABLE && u_on(p) && all_fast(p, k0, k1)) {     // same dispatch as the library
        std::vector<double> S(k1 - k0);
        const bool hot = p->u_h_cls > 0 && p->u_p_lag <= 2 && !getenv("TTM_EMU_NO_HOT");     // hot records (what k_forward_hl evaluates)
        for (int64_t n = 0; n < N; ++n) {
            XSoA xa{X, ldx, n};
            double ld, ss;
            if (!(hot && forward_h_dispatch<double>(p, fold, xa, k0, k1, logdet != nullptr, Z || sumsq, S.data(), ld, ss, sigma)))
                forward_u<double>(p, fold, xa, k0, k1, logdet != nullptr, Z || sumsq, S.data(), ld, ss, sigma);
            if (Z) for (int k = k0; k < k1; ++k) Z[(int64_t)(k - k0) * ldz + n] = S[k - k0];
            if (logdet) logdet[n] = ld;
            if (sumsq) sumsq[n] = ss;
        }
        return 0;
    }
    DenseClass dcls;
    if (int_dense(p, k0, k1, dcls)) {
        std::vector<double> scr(4096);
        std::vector<HostComp> hc(k1 - k0);
        for (int k = k0; k < k1; ++k) comp_of(p, k, coef + p->h_coef_off[k], hc[k - k0], fold + p->h_fold_off[k]);
        const double qws = dense_qw_sum(g);
        const bool xall = all_xprog(p, k0, k1);
        for (int64_t n = 0; n < N; ++n) {
            XSoA xa{X, ldx, n};
            double cbuf[8]; VarCache<XSoA, double> x(xa, CacheStore<double>{cbuf, 1});
            double ld = 0.0, ss = 0.0;
            for (int k = k0; k < k1; ++k) {
                const Comp& c = hc[k - k0].c;
                VecSlots w{scr.data()};
                double S = 0.0, dS = 0.0;
                if (xall) {
                    XProg xp;
                    xprog_view(p->itab + p->h_comp_off[k], p->dpar + p->h_dpar_off[k], xp);
                    const double* fx = c.fold + xp.fold_x;
#define TTM_CALL(PH, PP, RECT)                                                                                  \
    do {                                                                                                        \
        if (logdet) xprog_sample_forward<PH, PP, RECT, true>(xp, g, qws, fx, xa, w, Z || sumsq, S, dS);          \
        else xprog_sample_forward<PH, PP, RECT, false>(xp, g, qws, fx, xa, w, true, S, dS);                      \
    } while (0)
                    TTM_DENSE_DISPATCH(TTM_CALL, dcls, p->rectifier);
#undef TTM_CALL
                } else {
#define TTM_CALL(PH, PP, RECT)                                                                                  \
    do {                                                                                                        \
        if (logdet) dense_sample_forward<PH, PP, RECT, true>(c, g, qws, x, w, Z || sumsq, S, dS);               \
        else dense_sample_forward<PH, PP, RECT, false>(c, g, qws, x, w, true, S, dS);                           \
    } while (0)
                TTM_DENSE_DISPATCH(TTM_CALL, dcls, p->rectifier);
#undef TTM_CALL
                }
                if (logdet) ld += fast_log(sigma ? fast_div(dS, sigma[k - k0]) : dS);
                if (Z) Z[(int64_t)(k - k0) * ldz + n] = S;
                ss = fma(S, S, ss);
            }
            if (logdet) logdet[n] = ld;
            if (sumsq) sumsq[n] = ss;
        }
        return 0;
    }
    if (all_fast(p, k0, k1)) {
        std::vector<double> S(k1 - k0);
        for (int64_t n = 0; n < N; ++n) {
            XSoA xa{X, ldx, n};
            double ld, ss;
            if (p->family == TTM_FAM_HERMITE_E)
                forward_plan<TTM_FAM_HERMITE_E, double>(p, g, fold, xa, k0, k1, logdet != nullptr, Z || sumsq, S.data(), ld, ss, sigma);
            else
                forward_plan<-1, double>(p, g, fold, xa, k0, k1, logdet != nullptr, Z || sumsq, S.data(), ld, ss, sigma);
            if (Z) for (int k = k0; k < k1; ++k) Z[(int64_t)(k - k0) * ldz + n] = S[k - k0];
            if (logdet) logdet[n] = ld;
            if (sumsq) sumsq[n] = ss;
        }
        return 0;
    }
    std::vector<double> scr(4096);
    std::vector<HostComp> hc(k1 - k0);
    for (int k = k0; k < k1; ++k) comp_of(p, k, coef + p->h_coef_off[k], hc[k - k0], fold + p->h_fold_off[k]);
    for (int64_t n = 0; n < N; ++n) {
        XSoA xa{X, ldx, n};
        double cbuf[8]; VarCache<XSoA, double> x(xa, CacheStore<double>{cbuf, 1});
        double ld = 0.0, ss = 0.0;
        for (int k = k0; k < k1; ++k) {
            const Comp& c = hc[k - k0].c;
            VecSlots w{scr.data()};
            double S, dS;
            const int* fd = p->fdesc + k * TTM_FDESC_LEN;
            if (!fd[TTM_FD_COMPLEX]) {          // same dispatch as the kernel: flat-stream fast path
                const FastComp f = make_fast(fd, p->fints, fold, 0);
                TaggedFetch<XSoA, double> xf{x};
                const double xk = x.get(f.kc);
                if (logdet) sample_forward_fast<-1, -1, true>(f, g, xk, xf, Z || sumsq, S, dS);
                else sample_forward_fast<-1, -1, false>(f, g, xk, xf, true, S, dS);
            } else if (logdet) {
                sample_forward<-1, true>(c, g, x, w, Z || sumsq, S, dS);
            } else {
                sample_forward<-1, false>(c, g, x, w, true, S, dS);
            }
            if (logdet) ld += fast_log(sigma ? fast_div(dS, sigma[k - k0]) : dS);
            if (Z) Z[(int64_t)(k - k0) * ldz + n] = S;
            ss = fma(S, S, ss);
        }
        if (logdet) logdet[n] = ld;
        if (sumsq) sumsq[n] = ss;
    }
    return 0;
}

// TEST ONLY: forward map evaluated with two samples per "thread" (VecD<2>), as the GPU kernel does
struct VecSlots2 {
    double* base;
    VecD<2> get(int i) const { VecD<2> r; r.v[0] = base[2 * i]; r.v[1] = base[2 * i + 1]; return r; }
    void set(int i, const VecD<2>& v) { base[2 * i] = v.v[0]; base[2 * i + 1] = v.v[1]; }
};
struct XSoA2 {
    const double* X;
    int64_t ld, n0, n1;
    VecD<2> operator()(int var) const { VecD<2> r; r.v[0] = X[(int64_t)var * ld + n0]; r.v[1] = X[(int64_t)var * ld + n1]; return r; }
};

int emu_forward_vec2(const ttm_program* p, const double* coef, const double* fold, const double* X, int64_t ldx, int64_t N,
                     int32_t k0, int32_t k1, double* Z, int64_t ldz, double* logdet) {
    const Prog g = make_prog(p);
    if (p->monotonicity == TTM_MONO_SEPARABLE && u_on(p) && all_fast(p, k0, k1)) {
        std::vector<VecD<2>> S(k1 - k0);
        for (int64_t n = 0; n < N; n += 2) {
            const int64_t n1 = n + 1 < N ? n + 1 : n;
            XSoA2 xa{X, ldx, n, n1};
            VecD<2> ld, ss;
            const bool hot = p->u_h_cls > 0 && p->u_p_lag <= 2 && !getenv("TTM_EMU_NO_HOT");
            if (!(hot && forward_h_dispatch<VecD<2>>(p, fold, xa, k0, k1, true, true, S.data(), ld, ss, nullptr)))
                forward_u<VecD<2>>(p, fold, xa, k0, k1, true, true, S.data(), ld, ss, nullptr);
            for (int k = k0; k < k1; ++k) {
                Z[(int64_t)(k - k0) * ldz + n] = S[k - k0].v[0];
                Z[(int64_t)(k - k0) * ldz + n1] = S[k - k0].v[1];
            }
            logdet[n] = ld.v[0];
            logdet[n1] = ld.v[1];
        }
        return 0;
    }
    if (all_fast(p, k0, k1)) {
        std::vector<VecD<2>> S(k1 - k0);
        for (int64_t n = 0; n < N; n += 2) {
            const int64_t n1 = n + 1 < N ? n + 1 : n;
            XSoA2 xa{X, ldx, n, n1};
            VecD<2> ld, ss;
            if (p->family == TTM_FAM_HERMITE_E)
                forward_plan<TTM_FAM_HERMITE_E, VecD<2>>(p, g, fold, xa, k0, k1, true, true, S.data(), ld, ss, nullptr);
            else
                forward_plan<-1, VecD<2>>(p, g, fold, xa, k0, k1, true, true, S.data(), ld, ss, nullptr);
            for (int k = k0; k < k1; ++k) {
                Z[(int64_t)(k - k0) * ldz + n] = S[k - k0].v[0];
                Z[(int64_t)(k - k0) * ldz + n1] = S[k - k0].v[1];
            }
            logdet[n] = ld.v[0];
            logdet[n1] = ld.v[1];
        }
        return 0;
    }
    std::vector<double> scr(8192);
    std::vector<HostComp> hc(k1 - k0);
    for (int k = k0; k < k1; ++k) comp_of(p, k, coef + p->h_coef_off[k], hc[k - k0], fold + p->h_fold_off[k]);
    for (int64_t n = 0; n < N; n += 2) {
        const int64_t n1 = n + 1 < N ? n + 1 : n;
        XSoA2 xa{X, ldx, n, n1};
        double cbuf[16];
        VarCache<XSoA2, VecD<2>> x(xa, CacheStore<VecD<2>>{cbuf, 1});
        VecD<2> ld(0.0);
        for (int k = k0; k < k1; ++k) {
            const Comp& c = hc[k - k0].c;
            VecSlots2 w{scr.data()};
            VecD<2> S, dS;
            const int* fd = p->fdesc + k * TTM_FDESC_LEN;
            if (!fd[TTM_FD_COMPLEX]) {
                TaggedFetch<XSoA2, VecD<2>> xf{x};
                const FastComp f = make_fast(fd, p->fints, fold, 0);
                sample_forward_fast<-1, -1, true>(f, g, x.get(f.kc), xf, true, S, dS);
            } else
                sample_forward<-1, true>(c, g, x, w, true, S, dS);
            ld += fast_log(dS);
            Z[(int64_t)(k - k0) * ldz + n] = S.v[0];
            Z[(int64_t)(k - k0) * ldz + n1] = S.v[1];
        }
        logdet[n] = ld.v[0];
        logdet[n1] = ld.v[1];
    }
    return 0;
}

// TEST ONLY: the elementary functions of csrc/ttm_math.h on arrays (which: 0 exp, 1 erf, 2 exp(-t^2) from the
// erf table, 3 log, 4 reciprocal, 5 a/b with b = second array, 6 / 7 exp(-x^2/4) by table / series)
int emu_math(int which, const double* a, const double* b, int64_t n, double* out) {
    for (int64_t i = 0; i < n; ++i) {
        double e, g;
        switch (which) {
            case 0: out[i] = fast_exp(a[i]); break;
            case 1: erf_gauss_tab<true>(kErfTab, a[i], e, g); out[i] = e; break;
            case 2: erf_gauss_tab<true>(kErfTab, a[i], e, g); out[i] = g; break;
            case 3: out[i] = fast_log(a[i]); break;
            case 4: out[i] = fast_rcp(a[i]); break;
            case 6: out[i] = exp_q_tab(kExpQTab, a[i]); break;
            case 7: out[i] = exp_q_fast(a[i]); break;
            default: out[i] = fast_div(a[i], b[i]); break;
        }
    }
    return 0;
}

int ttm_basis(const ttm_program* p, int32_t k, int32_t which, const double* X, int64_t ldx, int64_t N, double* out, int64_t ldo, void*) {
    const Prog g = make_prog(p);
    HostComp h;
    comp_of(p, k, nullptr, h);
    const Comp& c = h.c;
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
    HostComp h;
    comp_of(p, k, coef_k, h);
    const Comp& c = h.c;
    const int nb1 = c.nB + 1;
    std::vector<double> scr(3 * nb1 + 8);
    const int nacc = g.mono == TTM_MONO_SEPARABLE ? 1 + c.n_mon : 1 + c.n_nm + c.n_mon;
    for (int i = 0; i < nacc; ++i) out[i] = 0.0;
    VecAcc acc{out};
    DenseClass dcls;
    const bool dense = int_dense(p, k, k + 1, dcls);
    const double qws = dense ? dense_qw_sum(g) : 0.0;
    XProg xp;
    if (dense && xprog_view(p->itab + p->h_comp_off[k], p->dpar + p->h_dpar_off[k], xp) &&
        !(getenv("TTM_INT_XPROG") && atoi(getenv("TTM_INT_XPROG")) == 0)) {
        // the X-program path of csrc/ttm_int.hip (k_int_objective): a row per sample, every sum a product of four row
        // entries - the kernel gives a sum to a lane and walks the rows of a wave; here the same products in sample order
        const double* fx = c.fold + xp.fold_x;                  // (comp_of folded the recipe, X section included)
        std::vector<double> rowbuf(xp.nrow + 2 * (TTM_I_PMAX + 1)), T(xp.nsum, 0.0);
        VecSlots row{rowbuf.data()};
        std::vector<int> cols(2 * xp.nsum);
#define TTM_CALL(PH, PP, RECT) for (int t = 0; t < xp.nsum; ++t) xobj_sum_columns((const int*)xp.anm, (const int*)xp.amon, xp.na_nm, xp.nrow, XQ<PH, PP>::NQ, t, cols[2 * t], cols[2 * t + 1])
        TTM_DENSE_DISPATCH(TTM_CALL, dcls, p->rectifier);
#undef TTM_CALL
        for (int64_t n = 0; n < N; ++n) {
            XSoA xa{X, ldx, n};
#define TTM_CALL(PH, PP, RECT) xobj_sample_row<PH, PP, RECT>(xp, g, qws, fx, xa, row, true)
            TTM_DENSE_DISPATCH(TTM_CALL, dcls, p->rectifier);
#undef TTM_CALL
            for (int t = 0; t < xp.nsum; ++t) T[t] = fma(row.get(cols[2 * t]), row.get(cols[2 * t + 1]), T[t]);
        }
#define TTM_CALL(PH, PP, RECT) for (int i = 0; i < nacc; ++i) { XResult<PH, PP> r; xobj_result_plan<PH, PP>(xp, g, i, r); out[i] = xobj_result_apply<PH, PP>(r, T); }
        TTM_DENSE_DISPATCH(TTM_CALL, dcls, p->rectifier);
#undef TTM_CALL
        return 0;
    }
    for (int64_t n = 0; n < N; ++n) {
        XSoA xa{X, ldx, n};
        double cbuf[8]; VarCache<XSoA, double> x(xa, CacheStore<double>{cbuf, 1});
        VecSlots w{scr.data()}, Bv{scr.data() + nb1}, I{scr.data() + 2 * nb1};
        if (g.mono == TTM_MONO_SEPARABLE) sample_objective_sep(c, g, x, w, acc);
        else if (dense) {
#define TTM_CALL(PH, PP, RECT) dense_sample_objective<PH, PP, RECT>(c, g, qws, x, w, w, I, acc)
            TTM_DENSE_DISPATCH(TTM_CALL, dcls, p->rectifier);
#undef TTM_CALL
        }
        else sample_objective_int(c, g, x, w, Bv, I, acc);
    }
    return 0;
}

int ttm_objective_host(const ttm_program* p, int32_t k, const double* h_coef_k, const double* X, int64_t ldx, int64_t N,
                       double* work, uint32_t*, double* out, void* stream) {
    return ttm_objective(p, k, h_coef_k, X, ldx, N, work, out, stream);
}

int ttm_objective_sep_cached(const double* dPsi, int64_t ldp, int64_t N, int32_t m, const double* h_coef_mon, double delta,
                             double*, uint32_t*, double* out, void*) {
    for (int i = 0; i <= m; ++i) out[i] = 0.0;
    for (int64_t n = 0; n < N; ++n) {
        double dS = 0.0, rowsum = 0.0;
        for (int i = 0; i < m; ++i) {
            const double d = dPsi[(int64_t)i * ldp + n];
            dS = fma(h_coef_mon[i], d, dS);
            rowsum += d;
        }
        dS += rowsum * delta;
        out[0] += fast_log(dS);
        const double inv = fast_rcp(dS);
        for (int i = 0; i < m; ++i) out[1 + i] += dPsi[(int64_t)i * ldp + n] * inv;
    }
    return 0;
}

// the separable sums with the derivative basis recomputed from the x_k column (special terms only)
int ttm_objective_sep_direct_marked(const double* xk, int64_t N, int32_t m, const int32_t* kinds, const double* pars,
                                    const double* h_coef_mon, double delta, double*, uint32_t*, double* out, double* flag,
                                    double mark, void*) {
    if (!xk || !kinds || !pars || !h_coef_mon || !out || N < 1 || m < 1 || m > 16) return TTM_E_ARG;
    Prog g;
    g.qx = nullptr; g.qw = nullptr; g.erf_tab = kErfTab; g.Q = 0; g.family = 0; g.mono = TTM_MONO_SEPARABLE; g.rect = 0; g.delta = delta;
    std::vector<double> acc(1 + m, 0.0), d(m);
    for (int64_t n = 0; n < N; ++n) {
        double dS = 0.0, rowsum = 0.0;
        for (int i = 0; i < m; ++i) {
            double v, dv;
            st_eval<false, true>(g, kinds[i], xk[n], pars + 5 * i, v, dv);
            d[i] = dv;
            dS = fma(h_coef_mon[i], dv, dS);
            rowsum += dv;
        }
        dS += rowsum * delta;
        acc[0] += fast_log(dS);
        const double inv = fast_rcp(dS);
        for (int i = 0; i < m; ++i) acc[1 + i] += d[i] * inv;
    }
    for (int i = 0; i <= m; ++i) out[i] = acc[i];
    if (flag) *flag = mark;
    return 0;
}

// marked variants: the reduction is done when the call returns
int ttm_objective_host_marked(const ttm_program* p, int32_t k, const double* h_coef_k, const double* X, int64_t ldx, int64_t N,
                              double* work, uint32_t* counter, double* out, double* flag, double mark, void* stream) {
    const int rc = ttm_objective_host(p, k, h_coef_k, X, ldx, N, work, counter, out, stream);
    if (!rc && flag) *flag = mark;
    return rc;
}
// (the device library's evaluation with self-validating sums: the double has nothing to wait for and declines - the loops
// of csrc/ttm_optim.cpp then take the marked call)
int ttm_sentinel_fill(double*, int32_t, int64_t, void*) { return TTM_E_UNSUPPORTED; }
void* ttm_mailbox_acquire(void) { return nullptr; }
void ttm_mailbox_release(void*) {}
int ttm_objective_sep_server_start(const double*, int64_t, int64_t, int32_t, double, double*, double*, const void*, uint32_t, void*) {
    return TTM_E_UNSUPPORTED;
}
int ttm_objective_sep_cached_sent(const double*, int64_t, int64_t, int32_t, const double*, double, double*, double*, void*) {
    return TTM_E_UNSUPPORTED;
}
int ttm_objective_sep_direct_sent(const double*, int64_t, int32_t, const int32_t*, const double*, const double*, double, double*, double*,
                                  void*) {
    return TTM_E_UNSUPPORTED;
}

int ttm_objective_sep_cached_marked(const double* dPsi, int64_t ldp, int64_t N, int32_t m, const double* h_coef_mon, double delta,
                                    double* work, uint32_t* counter, double* out, double* flag, double mark, void* stream) {
    const int rc = ttm_objective_sep_cached(dPsi, ldp, N, m, h_coef_mon, delta, work, counter, out, stream);
    if (!rc && flag) *flag = mark;
    return rc;
}

int ttm_gram(const ttm_program* p, int32_t k, const double* X, int64_t ldx, int64_t N, double*, double* out, void*);
int ttm_gram_many(const ttm_program* p, const int32_t* ks, int32_t nk, const double* X, int64_t ldx, int64_t N, double* work, double* out,
                  void* stream) {
    if (nk > 8) return TTM_E_UNSUPPORTED;
    for (int y = 0; y < nk; ++y) {
        const int m = p->h_coef_off[ks[y] + 1] - p->h_coef_off[ks[y]];
        const int rc = ttm_gram(p, ks[y], X, ldx, N, work, out, stream);
        if (rc) return rc;
        out += m * m;
    }
    return TTM_OK;
}

int ttm_gram(const ttm_program* p, int32_t k, const double* X, int64_t ldx, int64_t N, double*, double* out, void*) {
    const Prog g = make_prog(p);
    HostComp h;
    comp_of(p, k, nullptr, h);
    const Comp& c = h.c;
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

int ttm_inverse_table_build(const ttm_program* p, const double* coef, const double* fold, int32_t k0, int32_t k1,
                            const double* pts, int32_t T, double* out, void*) {
    const Prog g = make_prog(p);
    for (int k = k0; k < k1; ++k) {
        HostComp h;
        comp_of(p, k, coef + p->h_coef_off[k], h, fold + p->h_fold_off[k]);
        const Comp& c = h.c;
        std::vector<double> scr(c.nB + 2);
        for (int i = 0; i < T; ++i) {
            XFake x{c.kc, pts[i]};
            VecSlots w{scr.data()};
            mon_weights<double>(c, g, x, w);
            double v, dv;
            g_eval<false>(c, g, pts[i], w, v, dv);
            out[(int64_t)(k - k0) * T + i] = v;
        }
    }
    return 0;
}

int ttm_inverse_table_index(const double* tab_x, int32_t ncomp, int32_t T, int32_t nb, double* tmin, double* tmax,
                            int32_t* bkt, int32_t* unsorted, void*) {
    for (int c = 0; c < ncomp; ++c) {
        const double* xs = tab_x + (int64_t)c * T;
        int bad = 0;
        for (int i = 1; i < T; ++i) bad |= !(xs[i - 1] <= xs[i]);
        tmin[c] = xs[0]; tmax[c] = xs[T - 1]; unsorted[c] = bad;
        const double step = (xs[T - 1] - xs[0]) / (double)nb;
        for (int q = 0; q <= nb; ++q) {
            int a = 0, b = T;
            if (q == nb) a = T;
            else if (q > 0) {
                const double u = xs[0] + (double)q * step;
                while (a < b) { const int mid = (a + b) >> 1; if (xs[mid] < u) a = mid + 1; else b = mid; }
            }
            bkt[(int64_t)c * (nb + 1) + q] = a;
        }
    }
    return 0;
}

int ttm_inverse_table_build_index(const ttm_program* p, const double* coef, const double* fold, int32_t k0, int32_t k1, const double* pts,
                                  int32_t T, int32_t nb, double* out, double* tmin, double* tmax, int32_t* bkt, int32_t* unsorted,
                                  int32_t* h_unsorted, double* img, void* stream) {
    if (img) return TTM_E_ARG;                                 // (ttm_inverse_table_image_doubles is 0 here: no images on the host)
    int rc = ttm_inverse_table_build(p, coef, fold, k0, k1, pts, T, out, stream);
    if (!rc) rc = ttm_inverse_table_index(out, k1 - k0, T, nb, tmin, tmax, bkt, unsorted, stream);
    if (!rc && h_unsorted) memcpy(h_unsorted, unsorted, sizeof(int32_t) * (size_t)(k1 - k0));
    return rc;
}

int64_t ttm_inverse_table_image_doubles(const ttm_program*, int32_t, int32_t, int32_t, int32_t) { return 0; }

int ttm_setup_staged(const ttm_program* p, const double* h_coef, double* coef, double* fold, double* fold2, double* h_err, const double* pts,
                     int32_t T, int32_t nb, double* out, double* tmin, double* tmax, int32_t* bkt, int32_t* unsorted, int32_t* h_unsorted,
                     double* img, void* stream) {
    if (!fold2 || fold2 == fold) return TTM_E_ARG;
    int rc = ttm_fold_staged(p, h_coef, coef, fold, h_err, stream);
    if (!rc) rc = ttm_inverse_table_build_index(p, coef, fold, 0, p->D, pts, T, nb, out, tmin, tmax, bkt, unsorted, h_unsorted, img, stream);
    return rc;
}

int ttm_inverse_table(const ttm_program* p, const double* coef, const double* fold, int32_t k0, int32_t k1, const double* Z,
                      int64_t ldz, double* X, int64_t ldx, int64_t N, const double* tab_x, const double* tab_y, int64_t ldy,
                      int32_t T, const double* h_y_affine, const double* tmin, const double* tmax, const int32_t* bkt, int32_t nb,
                      int32_t truncate, const double*, int64_t, void*) {
    const Prog g = make_prog(p);
    if (u_on(p) && p->u_h_cls >= 1 && p->u_h_cls <= 3 && p->u_p_lag <= 2 && (p->u_h_ng == 2 || p->u_h_ng == 4) && all_fast(p, k0, k1) && h_y_affine && ldy == 0 &&
        (nb + 1) % 4 == 0 && !getenv("TTM_EMU_NO_HOT")) {
        // hot records + bucket scan + computed linspace abscissae: what k_inverse_hl evaluates
        const double* U = fold + fold_base_size(p);
        const int cls = p->u_h_cls, ng = p->u_h_ng;
        const int GS = u_h_gs(cls), hs = TTM_H_HDR + ng * GS;
        for (int64_t n = 0; n < N; ++n) {
            XSoA xa{X, ldx, n};
            double cbuf[2 * TTM_PLAN_WAYS];
            for (auto& c : cbuf) c = 0.0;
            CacheStore<double> st{cbuf, 1, kExpQTab};
            PlanCache<XSoA, double> x(xa, st);
            if (k0 > 0) x.warm(p->ucomp + TTM_UC_STATE(p->D, k0));
            for (int k = k0; k < k1; ++k) {
                const double* rec = U + p->u_h_off + (int64_t)k * hs;
                double off;
#define TTM_HO(NGV, C, DBV, DAV, GSV) if (ng == NGV && cls == C) off = h_offset<NGV, DBV, DAV, GSV, double>(rec, st)
                TTM_HO(2, 1, 3, 1, 8); TTM_HO(2, 2, 5, 5, 16); TTM_HO(2, 3, 7, 7, 24);
                TTM_HO(4, 1, 3, 1, 8); TTM_HO(4, 2, 5, 5, 16); TTM_HO(4, 3, 7, 7, 24);
#undef TTM_HO
                const double lo = tmin[k - k0], hi = tmax[k - k0];
                const double scale = (double)nb / (hi - lo);
                const bool use_bkt = scale > 0.0 && scale < 1.0e300;
                double target = -off + Z[(int64_t)(k - k0) * ldz + n];
                if (truncate) {
                    if (target < lo) target = lo;
                    if (target > hi) target = hi;
                }
                const double* xs = tab_x + (int64_t)(k - k0) * T;
                const int a = h_search(xs, bkt + (int64_t)(k - k0) * (nb + 1), nb, T, lo, scale, use_bkt, target);
                const int i = a < 1 ? 1 : (a > T - 1 ? T - 1 : a);
                const double y_lo = (double)(i - 1) * h_y_affine[1] + h_y_affine[0];
                const double y_hi = (i == T - 1) ? h_y_affine[2] : (double)i * h_y_affine[1] + h_y_affine[0];
                const double slope = fast_div(y_hi - y_lo, xs[i] - xs[i - 1]);
                const double r = slope * (target - xs[i - 1]) + y_lo;
                h_put(rec, st, r);
                X[(int64_t)((const int*)rec)[3] * ldx + n] = r;
            }
        }
        return 0;
    }
    (void)bkt; (void)nb; (void)h_y_affine;   // below: the full search; the accelerated search must give the same index
    if (all_fast(p, k0, k1)) {               // planned column cache, one sample at a time through all components
        for (int64_t n = 0; n < N; ++n) {
            XSoA xa{X, ldx, n};
            double cbuf[8];
            PlanCache<XSoA, double> x(xa, CacheStore<double>{cbuf, 1});
            if (k0 > 0) x.warm(p->fints + p->fdesc[k0 * TTM_FDESC_LEN + TTM_FD_PLAN_OFF]);
            for (int k = k0; k < k1; ++k) {
                const int* fd = p->fdesc + k * TTM_FDESC_LEN;
                const FastComp f = make_fast(fd, p->fints, fold, 0);
                const double off = p->family == TTM_FAM_HERMITE_E ? nonmon_sum_fast<TTM_FAM_HERMITE_E, double>(f, g, x)
                                                                  : nonmon_sum_fast<-1, double>(f, g, x);
                double target = -off + Z[(int64_t)(k - k0) * ldz + n];
                if (truncate) {
                    if (target < tmin[k - k0]) target = tmin[k - k0];
                    if (target > tmax[k - k0]) target = tmax[k - k0];
                }
                const double r = table_lookup(tab_x + (int64_t)(k - k0) * T, tab_y + (int64_t)(k - k0) * ldy, T, target);
                X[(int64_t)fd[TTM_FD_KC] * ldx + n] = r;
                x.put(fd[TTM_FD_KC_SLOT], r);
            }
        }
        return 0;
    }
    for (int k = k0; k < k1; ++k) {
        HostComp h;
        comp_of(p, k, coef + p->h_coef_off[k], h, fold + p->h_fold_off[k]);
        const Comp& c = h.c;
        const double* xs = tab_x + (int64_t)(k - k0) * T;
        const double* ys = tab_y + (int64_t)(k - k0) * ldy;
        for (int64_t n = 0; n < N; ++n) {
            XSoA xa{X, ldx, n};
            double cbuf[8]; VarCache<XSoA, double> x(xa, CacheStore<double>{cbuf, 1});
            const int* fd = p->fdesc + k * TTM_FDESC_LEN;
            TaggedFetch<XSoA, double> xf{x};
            const double off = !fd[TTM_FD_COMPLEX] ? nonmon_sum_fast<-1, double>(make_fast(fd, p->fints, fold, 0), g, xf)
                                                   : nonmon_sum<double>(c, g, x);
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

int ttm_inverse_bisect(const ttm_program* p, const double* coef, const double* fold, int32_t k0, int32_t k1, const double* Z,
                       int64_t ldz, double* X, int64_t ldx, int64_t N, int32_t* iters, const int32_t* cap, void*) {
    const Prog g = make_prog(p);
    std::vector<double> scr(4096);
    std::vector<HostComp> hc(k1 - k0);
    for (int k = k0; k < k1; ++k) comp_of(p, k, coef + p->h_coef_off[k], hc[k - k0], fold + p->h_fold_off[k]);
    DenseClass dcls;
    const bool dense = int_dense(p, k0, k1, dcls);
    const double qws = dense ? dense_qw_sum(g) : 0.0;
    for (int64_t n = 0; n < N; ++n) {
        XSoA xa{X, ldx, n};
        double cbuf[8]; VarCache<XSoA, double> x(xa, CacheStore<double>{cbuf, 1});
        for (int k = k0; k < k1; ++k) {
            const Comp& c = hc[k - k0].c;
            VecSlots w{scr.data()};
            int it = 0;
            double r = 0.0;
            if (dense && all_xprog(p, k0, k1) && getenv("TTM_INT_XPROG") && atoi(getenv("TTM_INT_XPROG")) == 2) {
                XProg xp;
                xprog_view(p->itab + p->h_comp_off[k], p->dpar + p->h_dpar_off[k], xp);
                const double* fx = c.fold + xp.fold_x;
#define TTM_CALL(PH, PP, RECT) r = xprog_sample_root<PH, PP, RECT, false>(xp, g, qws, fx, xa, w, Z[(int64_t)(k - k0) * ldz + n], cap ? cap[k - k0] : -1, it)
                TTM_DENSE_DISPATCH(TTM_CALL, dcls, p->rectifier);
#undef TTM_CALL
                X[(int64_t)c.kc * ldx + n] = r;
                x.put(c.kc, r);
                if (it > iters[k - k0]) iters[k - k0] = it;
                continue;
            }
            const double off = nonmon_sum<double>(c, g, x);
            if (dense) {
#define TTM_CALL(PH, PP, RECT) r = dense_sample_root<PH, PP, RECT, false>(c, g, qws, x, w, off, Z[(int64_t)(k - k0) * ldz + n], cap ? cap[k - k0] : -1, it)
                TTM_DENSE_DISPATCH(TTM_CALL, dcls, p->rectifier);
#undef TTM_CALL
            } else {
                r = sample_root<-1, false>(c, g, x, w, off, Z[(int64_t)(k - k0) * ldz + n], cap ? cap[k - k0] : -1, it);
            }
            X[(int64_t)c.kc * ldx + n] = r;
            x.put(c.kc, r);
            if (it > iters[k - k0]) iters[k - k0] = it;
        }
    }
    return 0;
}

int ttm_inverse_newton(const ttm_program* p, const double* coef, const double* fold, int32_t k0, int32_t k1, const double* Z,
                       int64_t ldz, double* X, int64_t ldx, int64_t N, int32_t* iters, void*) {
    const Prog g = make_prog(p);
    std::vector<double> scr(4096);
    std::vector<HostComp> hc(k1 - k0);
    for (int k = k0; k < k1; ++k) comp_of(p, k, coef + p->h_coef_off[k], hc[k - k0], fold + p->h_fold_off[k]);
    DenseClass dcls;
    const bool dense = int_dense(p, k0, k1, dcls);
    const double qws = dense ? dense_qw_sum(g) : 0.0;
    for (int64_t n = 0; n < N; ++n) {
        XSoA xa{X, ldx, n};
        double cbuf[8]; VarCache<XSoA, double> x(xa, CacheStore<double>{cbuf, 1});
        for (int k = k0; k < k1; ++k) {
            const Comp& c = hc[k - k0].c;
            VecSlots w{scr.data()};
            int it = 0;
            double r = 0.0;
            if (dense && all_xprog(p, k0, k1) && getenv("TTM_INT_XPROG") && atoi(getenv("TTM_INT_XPROG")) == 2) {
                XProg xp;
                xprog_view(p->itab + p->h_comp_off[k], p->dpar + p->h_dpar_off[k], xp);
                const double* fx = c.fold + xp.fold_x;
#define TTM_CALL(PH, PP, RECT) r = xprog_sample_root<PH, PP, RECT, true>(xp, g, qws, fx, xa, w, Z[(int64_t)(k - k0) * ldz + n], -1, it)
                TTM_DENSE_DISPATCH(TTM_CALL, dcls, p->rectifier);
#undef TTM_CALL
                X[(int64_t)c.kc * ldx + n] = r;
                x.put(c.kc, r);
                if (it > iters[k - k0]) iters[k - k0] = it;
                continue;
            }
            const double off = nonmon_sum<double>(c, g, x);
            if (dense) {
#define TTM_CALL(PH, PP, RECT) r = dense_sample_root<PH, PP, RECT, true>(c, g, qws, x, w, off, Z[(int64_t)(k - k0) * ldz + n], -1, it)
                TTM_DENSE_DISPATCH(TTM_CALL, dcls, p->rectifier);
#undef TTM_CALL
            } else {
                r = sample_root<-1, true>(c, g, x, w, off, Z[(int64_t)(k - k0) * ldz + n], -1, it);
            }
            X[(int64_t)c.kc * ldx + n] = r;
            x.put(c.kc, r);
            if (it > iters[k - k0]) iters[k - k0] = it;
        }
    }
    return 0;
}

}  // extern "C"

// column utilities of the device-resident ensemble filter (same per-row bodies)
extern "C" {
int ttm_lorenz63_rk4(double* E, int64_t ld, int64_t N, double dt, int32_t nt, void*) {
    if (!E || N < 1 || ld < N || nt < 0) return TTM_E_ARG;
    for (int64_t n = 0; n < N; ++n) {
        double x = E[n], y = E[ld + n], z = E[2 * ld + n];
        for (int i = 0; i < nt; ++i) ttm::lorenz63_rk4_step(x, y, z, dt);
        E[n] = x; E[ld + n] = y; E[2 * ld + n] = z;
    }
    return TTM_OK;
}
int ttm_perturb(const double* in, const double* noise, double sd, uint64_t seed, uint32_t stream_id, int64_t row0, int64_t N,
                double* out, void*) {
    if (!in || !out || N < 1) return TTM_E_ARG;
    for (int64_t n = 0; n < N; ++n) out[n] = in[n] + sd * (noise ? noise[n] : ttm::normal_deviate(seed, stream_id, (uint64_t)(row0 + n)));
    return TTM_OK;
}
int ttm_map_columns(const double* in, int64_t ldi, const int32_t* src, const double* scale, const double* shift, int32_t ncols,
                    int64_t N, double* out, int64_t ldo, void*) {
    if (!out || !src || N < 1 || ncols < 1 || ncols > 16 || ldo < N) return TTM_E_ARG;
    for (int j = 0; j < ncols; ++j)
        for (int64_t n = 0; n < N; ++n) {
            const double v = src[j] >= 0 ? in[(int64_t)src[j] * ldi + n] : 0.0;
            out[(int64_t)j * ldo + n] = v * (scale ? scale[j] : 1.0) + (shift ? shift[j] : 0.0);
        }
    return TTM_OK;
}
}

extern "C" int ttm_signal(double* flag, double value, void*) { if (!flag) return TTM_E_ARG; *flag = value; return TTM_OK; }

// the optimiser loops of the product, compiled for the host (no streams)
#define TTM_HOST_ONLY
#include "../../triangular_transport_toolbox_amd/csrc/ttm_optim.cpp"
