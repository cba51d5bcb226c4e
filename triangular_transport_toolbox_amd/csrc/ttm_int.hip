// ttm_int.hip - kernels of integrated-rectifier maps (the reference's default monotonicity, TM:12-39) whose components all
// have a dense B set: forward map (TM:2391-2567 with the quadrature of TM:4238-4258 fused), bisection / Newton root search
// (TM:3842-3976) and the objective / gradient sums of optimize() (TM:3343-3376, 3475-3569).
//
// The per-sample bodies are csrc/ttm_dense.h (monomial form of g, Horner per node, lean exp, NODES nodes per pass); what
// is here is the kernel around them.  The generic kernels of csrc/ttm_kernels.hip inline the whole term-table interpreter
// next to the node loop: 135 VGPRs, 248 spilled SGPRs (every spill a v_writelane / v_readlane in the node loop's
// neighbourhood) and 0.54 scalar instructions per vector instruction - the one scalar unit of a CU was co-limiting
// (profiles/r03_fp64_pmc_summary.json).  Here the interpreter runs ONCE per sample and component (weights of the x_k
// functions -> monomial coefficients in registers) and the node loop is a template over the order class and the rectifier:
// straight-line FMA chains, its uniform operands (node abscissae / weights, polynomial coefficients of exp) from scalar
// loads.  One thread = one sample, samples column-major: every column access of a wave is one 512-byte transaction.
//
// Bound: fp64 vector rate (25 nodes x ~45 instructions per evaluation against 8 (d_used + D) bytes per sample:
// 1.1e3 flop per byte at C2a) - HBM is idle by construction, the roofline of these kernels is 78.6 TFLOP/s.

#include <hip/hip_runtime.h>

#include "ttm_dev.h"
#include "ttm_dense.h"
#include "ttm_xprog.h"
#include "ttm_int.h"

using namespace ttm;

namespace {

// grid: x = tiles of blockDim samples, y = chunks of `chunk` components (the components of a forward map are independent:
// a small ensemble still fills the chip; the fused log-determinant / sum of squares run over all components: one chunk)
template <int PH, int PP, int RECT, bool WANT_LD>
__global__ __launch_bounds__(256) void k_int_forward(DevProg P, int kfirst, int klast, int chunk, int erf, const double* __restrict__ coef,
                                                     const double* __restrict__ fold, const double* __restrict__ X, int64_t ldx,
                                                     int64_t N, double* __restrict__ Z, int64_t ldz, double* __restrict__ logdet,
                                                     const double* __restrict__ sigma, double* __restrict__ sumsq) {
    const int k0 = kfirst + (int)blockIdx.y * chunk;
    const int k1 = (k0 + chunk < klast) ? k0 + chunk : klast;
    double* slots;
    CacheStore<double> cst;
    const Prog g = make_prog_lds(P, cst, slots, erf != 0);
    const int bd = blockDim.x;
    LdsSlots w{slots + threadIdx.x, bd};
    const bool want_val = (Z != nullptr) || (sumsq != nullptr);
    const double qws = dense_qw_sum(g);
    cint_p fdesc = (cint_p)P.fdesc;
    for (int64_t n0 = (int64_t)blockIdx.x * bd; n0 < N; n0 += (int64_t)gridDim.x * bd) {
        const int64_t n = n0 + threadIdx.x;
        const bool active = n < N;
        const XSoA xa{X, ldx, active ? n : N - 1};
        VarCache<XSoA, double> x(xa, cst);
        double ld = 0.0, ss = 0.0;
        // the component's own column is fetched one component ahead
        double xk_next = xa(fdesc[k0 * TTM_FDESC_LEN + TTM_FD_KC]);
        for (int k = k0; k < k1; ++k) {
            cint_p fd = fdesc + k * TTM_FDESC_LEN;
            x.put(fd[TTM_FD_KC], xk_next);
            if (k + 1 < k1) xk_next = xa(fd[TTM_FDESC_LEN + TTM_FD_KC]);
            const Comp c = comp_at(P, k, 0, coef, fold);
            double S, dS;
            dense_sample_forward<PH, PP, RECT, WANT_LD>(c, g, qws, x, w, WANT_LD ? want_val : true, S, dS);
            if (WANT_LD) ld += fast_log(sigma ? fast_div(dS, ((cdbl_p)sigma)[k - kfirst]) : dS);
            if (Z && active) Z[(int64_t)(k - kfirst) * ldz + n] = S;
            ss = fma(S, S, ss);
        }
        if (active) {
            if (WANT_LD) logdet[n] = ld;
            if (sumsq) sumsq[n] = ss;
        }
    }
}

template <int PH, int PP, int RECT, bool NEWTON>
__global__ __launch_bounds__(256) void k_int_root(DevProg P, int k0, int k1, int erf, const double* __restrict__ coef,
                                                  const double* __restrict__ fold, const double* __restrict__ Z, int64_t ldz,
                                                  double* X, int64_t ldx, int64_t N, int* __restrict__ iters,
                                                  const int* __restrict__ cap) {
    double* slots;
    CacheStore<double> cst;
    const Prog g = make_prog_lds(P, cst, slots, erf != 0);
    LdsSlots w{slots + threadIdx.x, (int)blockDim.x};
    const double qws = dense_qw_sum(g);
    for (int64_t n0 = (int64_t)blockIdx.x * blockDim.x; n0 < N; n0 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = n0 + threadIdx.x;
        const bool active = n < N;
        const XSoA xa{X, ldx, active ? n : 0};
        VarCache<XSoA, double> x(xa, cst);
        for (int k = k0; k < k1; ++k) {
            const Comp c = comp_at(P, k, 0, coef, fold);
            int it = 0;
            if (active) {
                const double off = nonmon_sum<double>(c, g, x);
                const int capk = cap ? cap[k - k0] : -1;
                const double zk = Z[(int64_t)(k - k0) * ldz + n];
                const double r = dense_sample_root<PH, PP, RECT, NEWTON>(c, g, qws, x, w, off, zk, capk, it);
                X[(int64_t)c.kc * ldx + n] = r;
                x.put(c.kc, r);
            }
            // wave-level max, one atomic per wave (the sample-0 guard of the reference's loop needs the largest count)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) it = max(it, __shfl_down(it, off, 64));
            if ((threadIdx.x & 63) == 0 && it > 0) atomicMax(&iters[k - k0], it);
        }
    }
}

// LDS: erf table | column cache | per-thread scratch columns (nscr) | one row of nacc running sums per wave
template <int PH, int PP, int RECT>
__global__ __launch_bounds__(256) void k_int_objective_walk(DevProg P, int k, const double* __restrict__ coef_k,
                                                       const double* __restrict__ fold_k, const double* __restrict__ X,
                                                       int64_t ldx, int64_t N, int nscr, int nacc, double* __restrict__ partial,
                                                       unsigned int* __restrict__ counter, double* __restrict__ out, double* flag,
                                                       double mark) {
    double* slots;
    CacheStore<double> cst;
    const Prog g = make_prog_lds(P, cst, slots);
    const int bd = blockDim.x, tid = threadIdx.x;
    Comp c = comp_at(P, k, k, coef_k, fold_k);
    {   // the fold recipe of the component: the gradient walks the members of the folded sums (ttm_eval.h, objective_gradient)
        cint_p off = (cint_p)P.off;
        cint_p cb = (cint_p)P.itab + off[k];
        cint_p fb = (cint_p)P.ftab + off[4 * (P.D + 1) + k];
        c.fslot = fb + cb[TTM_HDR_OFF_FSLOT];
        c.fsrc = fb + cb[TTM_HDR_OFF_FSRC];
    }
    const int nb1 = c.nB + 1;
    const int lane = tid & 63, wv = tid >> 6, nw = bd >> 6;
    double* accbase = slots + (size_t)nscr * bd;
    for (int i = tid; i < nw * nacc; i += bd) accbase[i] = 0.0;
    __syncthreads();
    WaveAcc acc{accbase + wv * nacc, true, lane == 63};
    LdsSlots w{slots + tid, bd};                                  // (the B values at x_k reuse the weights' columns)
    LdsSlots I{slots + (size_t)nb1 * bd + tid, bd};
    const double qws = dense_qw_sum(g);
    for (int64_t n0 = (int64_t)blockIdx.x * bd; n0 < N; n0 += (int64_t)gridDim.x * bd) {
        const int64_t n = n0 + tid;
        acc.active = n < N;
        const XSoA xa{X, ldx, acc.active ? n : N - 1};
        VarCache<XSoA, double> x(xa, cst);
        dense_sample_objective<PH, PP, RECT>(c, g, qws, x, w, w, I, acc);
    }
    __syncthreads();
    for (int i = tid; i < nacc; i += bd) {
        double v = 0.0;
        for (int wq = 0; wq < nw; ++wq) v += accbase[wq * nacc + i];
        partial[(int64_t)blockIdx.x * nacc + i] = v;
    }
    if (out) {
        if (last_workgroup(counter)) {
            double* fin = slots;
            for (int i = wv; i < nacc; i += nw) {
                double v = 0.0;
                for (int b = lane; b < (int)gridDim.x; b += 64) v += partial[(int64_t)b * nacc + i];
                v = wave_sum(v);
                if (lane == 0) fin[i] = v;
            }
            publish(fin, nacc, out, flag, mark);
        }
    }
}

// ---- forward map and root searches through the X programs (csrc/ttm_xprog.h): every component of the launch has one ---------
// k_int_forward / k_int_root with the term-table walks of a sample (weights of the B functions, their conversion to monomial
// coefficients, nonmonotone groups: 620 of the 1 780 vector instructions of an evaluation at C2a, 0.5-0.7 scalar instructions
// per vector instruction) replaced by the sample's row: factor values once, the monomial form of g and the nonmonotone sum
// as products with the fold's X section (scalar loads from the fold buffer at addresses known from the component's header).
// LDS: one row per thread, column-major over the workgroup (nrow x blockDim doubles); columns come from global memory
// (a column is read by at most a few components: L2 hits; the kernels are FP64-bound).
template <int PH, int PP, int RECT, bool WANT_LD>
__global__ __launch_bounds__(256) void k_int_forward_x(DevProg P, int kfirst, int klast, int chunk, const double* __restrict__ fold,
                                                       const double* __restrict__ X, int64_t ldx, int64_t N, double* __restrict__ Z,
                                                       int64_t ldz, double* __restrict__ logdet, const double* __restrict__ sigma,
                                                       double* __restrict__ sumsq) {
    const int k0 = kfirst + (int)blockIdx.y * chunk;
    const int k1 = (k0 + chunk < klast) ? k0 + chunk : klast;
    const int bd = blockDim.x;
    LdsSlots row{g_smem + threadIdx.x, bd};
    Prog g;
    g.qx = (cdbl_p)P.qx; g.qw = (cdbl_p)P.qw; g.erf_tab = nullptr; g.Q = P.Q; g.family = P.family; g.mono = P.mono; g.rect = P.rect;
    g.delta = P.delta;
    const bool want_val = (Z != nullptr) || (sumsq != nullptr);
    const double qws = dense_qw_sum(g);
    cint_p off = (cint_p)P.off;
    const int D1 = P.D + 1;
    for (int64_t n0 = (int64_t)blockIdx.x * bd; n0 < N; n0 += (int64_t)gridDim.x * bd) {
        const int64_t n = n0 + threadIdx.x;
        const bool active = n < N;
        const XSoA xa{X, ldx, active ? n : N - 1};
        double ld = 0.0, ss = 0.0;
        for (int k = k0; k < k1; ++k) {
            XProg xp;
            xprog_view((cint_p)P.itab + off[k], (cdbl_p)P.dpar + off[D1 + k], xp);
            cdbl_p fx = (cdbl_p)fold + off[3 * D1 + k] + xp.fold_x;            // (fold: the folded coefficients of the WHOLE map)
            double S, dS;
            xprog_sample_forward<PH, PP, RECT, WANT_LD>(xp, g, qws, fx, xa, row, WANT_LD ? want_val : true, S, dS);
            if (WANT_LD) ld += fast_log(sigma ? fast_div(dS, ((cdbl_p)sigma)[k - kfirst]) : dS);
            if (Z && active) Z[(int64_t)(k - kfirst) * ldz + n] = S;
            ss = fma(S, S, ss);
        }
        if (active) {
            if (WANT_LD) logdet[n] = ld;
            if (sumsq) sumsq[n] = ss;
        }
    }
}

template <int PH, int PP, int RECT, bool NEWTON>
__global__ __launch_bounds__(256) void k_int_root_x(DevProg P, int k0, int k1, const double* __restrict__ fold, const double* __restrict__ Z,
                                                    int64_t ldz, double* X, int64_t ldx, int64_t N, int* __restrict__ iters,
                                                    const int* __restrict__ cap) {
    LdsSlots row{g_smem + threadIdx.x, (int)blockDim.x};
    Prog g;
    g.qx = (cdbl_p)P.qx; g.qw = (cdbl_p)P.qw; g.erf_tab = nullptr; g.Q = P.Q; g.family = P.family; g.mono = P.mono; g.rect = P.rect;
    g.delta = P.delta;
    const double qws = dense_qw_sum(g);
    cint_p off = (cint_p)P.off;
    const int D1 = P.D + 1;
    for (int64_t n0 = (int64_t)blockIdx.x * blockDim.x; n0 < N; n0 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = n0 + threadIdx.x;
        const bool active = n < N;
        // (columns of the components solved so far are read back from X: this lane wrote them itself)
        const XSoA xa{X, ldx, active ? n : 0};
        for (int k = k0; k < k1; ++k) {
            XProg xp;
            xprog_view((cint_p)P.itab + off[k], (cdbl_p)P.dpar + off[D1 + k], xp);
            cdbl_p fx = (cdbl_p)fold + off[3 * D1 + k] + xp.fold_x;            // (fold: the folded coefficients of the WHOLE map)
            int it = 0;
            if (active) {
                const int capk = cap ? cap[k - k0] : -1;
                const double zk = Z[(int64_t)(k - k0) * ldz + n];
                const double r = xprog_sample_root<PH, PP, RECT, NEWTON>(xp, g, qws, fx, xa, row, zk, capk, it);
                X[(int64_t)xp.kc * ldx + n] = r;
            }
            // wave-level max, one atomic per wave (the sample-0 guard of the reference's loop needs the largest count)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) it = max(it, __shfl_down(it, o, 64));
            if ((threadIdx.x & 63) == 0 && it > 0) atomicMax(&iters[k - k0], it);
        }
    }
}

// ---- objective + gradient sums through the X program (csrc/ttm_xprog.h) ---------------------------------------------------
// LDS: [coefficients (TTM_HOSTCOEF_MAX) | the component's fold (nfold) | per wave: rows of its 64 samples, COLUMN-major with a
// column stride of XOBJ_CS = 65 doubles | per wave XOBJ_SUMS running totals].  Column-major: in the per-sample phase the
// 64 lanes of a wave write one column at consecutive addresses; in the sum phase lane t reads column c(t) at the SAME sample
// s - addresses c(t) 65 + s, different columns on different banks - and s is an immediate offset of the unrolled loop.
// Per tile of blockDim samples: every lane writes the row of its sample (xobj_sample_row), then - inside each wave, no
// workgroup barrier - lane t walks the wave's 64 rows with the two columns of sum t (a component with more than 64 sums
// gives a lane up to three).  Two partial totals per sum keep the adds independent; fixed order, run-to-run deterministic.
// The coefficients travel as a kernel argument (or come from device memory); the workgroup folds them itself (fold_coeffs:
// the X section of the fold is what the per-sample phase reads): ONE launch per evaluation, finished - sums of the
// workgroups, then the basis conversion of xobj_result - by the workgroup that draws the last ticket.
#define XOBJ_SUMS TTM_X_SUM_MAX
#define XOBJ_CS 65
#define XOBJ_NCH ((TTM_X_SUM_MAX + 63) / 64)
struct XCoef { double c[TTM_HOSTCOEF_MAX]; };

struct WaveRow {             // the row of this lane's sample inside its wave's column-major block
    double* base;
    __device__ __forceinline__ double get(int c) const { return base[c * XOBJ_CS]; }
    __device__ __forceinline__ void set(int c, double v) { base[c * XOBJ_CS] = v; }
};
struct SumsView {
    const double* t;
    __device__ __forceinline__ double operator[](int i) const { return t[i]; }
};

__device__ __forceinline__ void wave_lds_sync() {
    // rows written by the lanes of this wave are read by other lanes of the SAME wave: LDS operations of a wave complete in
    // order, what is needed is that the compiler keeps them in order and waits for the writes' return
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef XOBJ_OCC
#define XOBJ_OCC 1
#endif
template <int PH, int PP, int RECT, int NCH>
__global__ __launch_bounds__(256, XOBJ_OCC) void k_int_objective(DevProg P, int k, XCoef hc, const double* __restrict__ d_coef, int ncoef,
                                                                 const double* __restrict__ X, int64_t ldx, int64_t N, int nfold, int ncols,
                                                                 double* __restrict__ partial, unsigned int* __restrict__ counter,
                                                                 double* __restrict__ out, double* flag, double mark) {
    constexpr int NQ = XQ<PH, PP>::NQ;
    const int bd = blockDim.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = bd >> 6;
    double* lcoef = g_smem;
    double* lfold = lcoef + TTM_HOSTCOEF_MAX;
    double* rows = lfold + nfold;
    double* wsum = rows + (size_t)nw * ncols * XOBJ_CS;
    // the program first (chains of dependent scalar loads: they travel while the coefficients are folded)
    const int* off = P.off;
    const int D1 = P.D + 1;
    cint_p cb = (cint_p)P.itab + off[k];
    XProg xp;
    xprog_view(cb, (cdbl_p)P.dpar + off[D1 + k], xp);
    Prog g;
    g.qx = (cdbl_p)P.qx; g.qw = (cdbl_p)P.qw; g.erf_tab = nullptr; g.Q = P.Q; g.family = P.family; g.mono = P.mono; g.rect = P.rect;
    g.delta = P.delta;
    const double qws = dense_qw_sum(g);
    const int nsum = xp.nsum;
    // the sums of this lane: t = lane + 64 c; the two columns as offsets (doubles) into the wave's block
    double* wblock = rows + (size_t)wv * ncols * XOBJ_CS;
    const int* ganm = P.itab + off[k] + (int)(xp.anm - cb);
    const int* gamon = P.itab + off[k] + (int)(xp.amon - cb);
    int o1[NCH], o2[NCH];
    double acc[NCH][2];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        int c1 = 0, c2 = 0;
        if (lane + 64 * c < nsum) xobj_sum_columns(ganm, gamon, xp.na_nm, xp.nrow, NQ, lane + 64 * c, c1, c2);
        o1[c] = c1 * XOBJ_CS; o2[c] = c2 * XOBJ_CS;
        acc[c][0] = acc[c][1] = 0.0;
    }
    for (int i = tid; i < TTM_HOSTCOEF_MAX; i += bd) lcoef[i] = d_coef ? (i < ncoef ? d_coef[i] : 0.0) : hc.c[i];
    __syncthreads();
#ifndef XOBJ_X_NOFOLD                                        /* (XOBJ_X_*: timing builds, results wrong by construction) */
    fold_coeffs(P.itab + off[k], P.ftab + off[4 * D1 + k], P.dpar + off[D1 + k], lcoef, lfold, tid, bd);
#else
    for (int i = tid; i < nfold; i += bd) lfold[i] = 0.01;
#endif
    __syncthreads();
    WaveRow row{wblock + lane};
    for (int64_t n0 = (int64_t)blockIdx.x * bd; n0 < N; n0 += (int64_t)gridDim.x * bd) {
        const int64_t n = n0 + tid;
        const bool active = n < N;
        const XSoA xa{X, ldx, active ? n : N - 1};
        // (the fold is constant over the tiles: without this the compiler keeps every entry the per-sample phase reads in a
        // vector register across the whole loop)
        const double* fx = lfold + xp.fold_x;
        asm volatile("" : "+v"(fx));
#ifndef XOBJ_X_NOROW
        xobj_sample_row<PH, PP, RECT>(xp, g, qws, fx, xa, row, active);
#else
        for (int c = 0; c < ncols; ++c) row.set(c, xa(xp.kc) + fx[c & 7]);
#endif
        wave_lds_sync();
#ifndef XOBJ_X_NOSUM
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (64 * c < nsum) {
                const double* p1 = wblock + o1[c];
                const double* p2 = wblock + o2[c];
#pragma unroll
                for (int s = 0; s < 64; ++s) acc[c][s & 1] = fma(p1[s], p2[s], acc[c][s & 1]);
            }
        }
#else
        acc[0][0] += wblock[o1[0]] + wblock[o2[0]];
#endif
        wave_lds_sync();
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) wsum[wv * XOBJ_SUMS + 64 * c + lane] = acc[c][0] + acc[c][1];
    __syncthreads();
    for (int i = tid; i < nsum; i += bd) {
        double v = 0.0;
        for (int wq = 0; wq < nw; ++wq) v += wsum[wq * XOBJ_SUMS + i];
        coherent_store(partial + (int64_t)blockIdx.x * nsum + i, v);
    }
    drain_stores();
#ifdef XOBJ_X_NOFIN
    if (blockIdx.x == 0 && tid == 0) __hip_atomic_store(flag ? flag : out, mark, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
#endif
    {
        // (LDS: the rows are done with - group / final sums in `rows`, scratch of nw x nsum in the wave totals' place)
        double* fin = rows;                                       // sums of all workgroups, then the results behind them
        double* res = rows + XOBJ_SUMS;
        __shared__ int stage_a;
        if (!finish_stage_a(partial, nsum, counter, wsum, &stage_a)) return;
        // what result `tid` is made of (program reads) travels while the last group rows arrive
        XResult<PH, PP> plan;
        if (tid <= ncoef) xobj_result_plan<PH, PP>(xp, g, tid, plan);
        if (!finish_stage_b(partial, nsum, counter, wsum, fin, &stage_a)) return;
        const SumsView T{fin};
        if (tid <= ncoef) res[tid] = xobj_result_apply<PH, PP>(plan, T);
        for (int i = tid + bd; i <= ncoef; i += bd) {             // (more results than threads: 128 coefficients, a workgroup of 128)
            xobj_result_plan<PH, PP>(xp, g, i, plan);
            res[i] = xobj_result_apply<PH, PP>(plan, T);
        }
        publish(res, 1 + ncoef, out, flag, mark);
    }
}

}  // namespace

namespace ttm_int {

// does any component of [k0, k1) evaluate a special term (erf table in LDS)?
static int needs_erf(const ttm_program* p, int k0, int k1) {
    for (int k = k0; k < k1; ++k)
        if (p->h_complex[k] & 8) return 1;
    return 0;
}

// row columns the X-program kernels need for the components [k0, k1): 0 when one of them has no X program
static int xprog_rows(const ttm_program* p, int k0, int k1) {
    int rows = 0;
    for (int k = k0; k < k1; ++k) {
        if (!(p->h_complex[k] & 16)) return 0;
        const int r = (p->h_complex[k] >> 16) & 255;
        rows = r > rows ? r : rows;
    }
    return rows;
}

bool usable(const ttm_program* p, int k0, int k1) {
    if (!p || p->monotonicity != TTM_MONO_INTEGRATED || p->family < 0 || p->family > 5) return false;
    DenseClass cls;
    return dense_range_class(p->h_complex, k0, k1, cls);
}

int forward(const ttm_program* p, const DevProg& P, int k0, int k1, const double* coef, const double* fold, const double* Xsoa,
            int64_t ldx, int64_t N, double* Zsoa, int64_t ldz, double* logdet, const double* sigma, double* sumsq, int grid,
            int chunk, int bd, size_t lds, int use_x, void* stream, const char** kernel_name) {
    DenseClass cls;
    if (!dense_range_class(p->h_complex, k0, k1, cls)) return TTM_E_UNSUPPORTED;
    if (chunk < 1 || logdet || sumsq) chunk = k1 - k0;
    const int erf = needs_erf(p, k0, k1);
    const dim3 g3(grid, (k1 - k0 + chunk - 1) / chunk);
    const int xrows = use_x ? xprog_rows(p, k0, k1) : 0;
    if (xrows > 0 && (size_t)xrows * bd * sizeof(double) <= 64 * 1024) {
        const size_t xlds = (size_t)xrows * bd * sizeof(double);
#define TTM_CALL(PH, PP, RECT)                                                                                              \
    do {                                                                                                                    \
        if (logdet) hipLaunchKernelGGL((k_int_forward_x<PH, PP, RECT, true>), g3, dim3(bd), xlds, (hipStream_t)stream, P, k0, \
                                       k1, chunk, fold, Xsoa, ldx, N, Zsoa, ldz, logdet, sigma, sumsq);                      \
        else hipLaunchKernelGGL((k_int_forward_x<PH, PP, RECT, false>), g3, dim3(bd), xlds, (hipStream_t)stream, P, k0, k1,  \
                                chunk, fold, Xsoa, ldx, N, Zsoa, ldz, logdet, sigma, sumsq);                                 \
    } while (0)
        TTM_DENSE_DISPATCH(TTM_CALL, cls, p->rectifier);
#undef TTM_CALL
        *kernel_name = "k_int_forward";
        return TTM_OK;
    }
#define TTM_CALL(PH, PP, RECT)                                                                                              \
    do {                                                                                                                    \
        if (logdet) hipLaunchKernelGGL((k_int_forward<PH, PP, RECT, true>), g3, dim3(bd), lds, (hipStream_t)stream, P, k0,  \
                                       k1, chunk, erf, coef, fold, Xsoa, ldx, N, Zsoa, ldz, logdet, sigma, sumsq);          \
        else hipLaunchKernelGGL((k_int_forward<PH, PP, RECT, false>), g3, dim3(bd), lds, (hipStream_t)stream, P, k0, k1,    \
                                chunk, erf, coef, fold, Xsoa, ldx, N, Zsoa, ldz, logdet, sigma, sumsq);                     \
    } while (0)
    TTM_DENSE_DISPATCH(TTM_CALL, cls, p->rectifier);
#undef TTM_CALL
    *kernel_name = "k_int_forward<walk>";
    return TTM_OK;
}

int root(const ttm_program* p, const DevProg& P, int k0, int k1, const double* coef, const double* fold, const double* Zsoa,
         int64_t ldz, double* Xsoa, int64_t ldx, int64_t N, int32_t* iters, const int32_t* cap, int newton, int grid, int bd,
         size_t lds, int use_x, void* stream, const char** kernel_name) {
    DenseClass cls;
    if (!dense_range_class(p->h_complex, k0, k1, cls)) return TTM_E_UNSUPPORTED;
    const int erf = needs_erf(p, k0, k1);
    const int xrows = use_x ? xprog_rows(p, k0, k1) : 0;
    if (xrows > 0 && (size_t)xrows * bd * sizeof(double) <= 64 * 1024) {
        const size_t xlds = (size_t)xrows * bd * sizeof(double);
#define TTM_CALL(PH, PP, RECT)                                                                                              \
    do {                                                                                                                    \
        if (newton) hipLaunchKernelGGL((k_int_root_x<PH, PP, RECT, true>), dim3(grid), dim3(bd), xlds, (hipStream_t)stream, P, k0, k1, \
                                       fold, Zsoa, ldz, Xsoa, ldx, N, (int*)iters, (const int*)cap);                        \
        else hipLaunchKernelGGL((k_int_root_x<PH, PP, RECT, false>), dim3(grid), dim3(bd), xlds, (hipStream_t)stream, P, k0, k1,      \
                                fold, Zsoa, ldz, Xsoa, ldx, N, (int*)iters, (const int*)cap);                               \
    } while (0)
        TTM_DENSE_DISPATCH(TTM_CALL, cls, p->rectifier);
#undef TTM_CALL
        *kernel_name = newton ? "k_int_root_x<newton>" : "k_int_root_x<bisect>";
        return TTM_OK;
    }
#define TTM_CALL(PH, PP, RECT)                                                                                              \
    do {                                                                                                                    \
        if (newton) hipLaunchKernelGGL((k_int_root<PH, PP, RECT, true>), dim3(grid), dim3(bd), lds, (hipStream_t)stream, P, k0, k1, \
                                       erf, coef, fold, Zsoa, ldz, Xsoa, ldx, N, (int*)iters, (const int*)cap);              \
        else hipLaunchKernelGGL((k_int_root<PH, PP, RECT, false>), dim3(grid), dim3(bd), lds, (hipStream_t)stream, P, k0, k1, erf,  \
                                coef, fold, Zsoa, ldz, Xsoa, ldx, N, (int*)iters, (const int*)cap);                          \
    } while (0)
    TTM_DENSE_DISPATCH(TTM_CALL, cls, p->rectifier);
#undef TTM_CALL
    *kernel_name = newton ? "k_int_root<newton>" : "k_int_root<bisect>";
    return TTM_OK;
}

int objective(const ttm_program* p, const DevProg& P, int k, const double* coef_k, const double* fold_k, const double* Xsoa,
              int64_t ldx, int64_t N, int nscr, int nacc, double* partial, unsigned int* counter, double* out, double* flag,
              double mark, int grid, int bd, size_t lds, void* stream, const char** kernel_name) {
    DenseClass cls;
    if (!dense_range_class(p->h_complex, k, k + 1, cls)) return TTM_E_UNSUPPORTED;
#define TTM_CALL(PH, PP, RECT)                                                                                             \
    hipLaunchKernelGGL((k_int_objective_walk<PH, PP, RECT>), dim3(grid), dim3(bd), lds, (hipStream_t)stream, P, k, coef_k, fold_k, Xsoa, \
                       ldx, N, nscr, nacc, partial, counter, out, flag, mark)
    TTM_DENSE_DISPATCH(TTM_CALL, cls, p->rectifier);
#undef TTM_CALL
    *kernel_name = "k_int_objective_walk";
    return TTM_OK;
}

bool has_xprog(const ttm_program* p, int k) { return usable(p, k, k + 1) && (p->h_complex[k] & 16) != 0; }

// row columns of the objective kernel: the X program's own + the q columns of the component's order class
static int objective_x_cols(const ttm_program* p, int k) {
    DenseClass cls;
    if (!dense_range_class(p->h_complex, k, k + 1, cls)) return 0;
    return ((p->h_complex[k] >> 16) & 255) + (cls.ph > 0 ? cls.ph + 1 : 0) + cls.pp + 1;
}

size_t objective_x_lds(const ttm_program* p, int k, int bd) {
    const int nfold = p->h_fold_off[k + 1] - p->h_fold_off[k];
    size_t rows = (size_t)(bd / 64) * objective_x_cols(p, k) * XOBJ_CS;
    if (rows < (size_t)2 * XOBJ_SUMS) rows = 2 * XOBJ_SUMS;       // (the finishing workgroup keeps sums and results there)
    return ((size_t)TTM_HOSTCOEF_MAX + nfold + rows + (size_t)(bd / 64) * XOBJ_SUMS) * sizeof(double);
}

int objective_x(const ttm_program* p, const DevProg& P, int k, const double* h_coef_k, const double* d_coef_k, int ncoef, const double* Xsoa,
                int64_t ldx, int64_t N, double* partial, unsigned int* counter, double* out, double* flag, double mark,
                int grid, int bd, void* stream, const char** kernel_name) {
    DenseClass cls;
    if (!dense_range_class(p->h_complex, k, k + 1, cls) || !(p->h_complex[k] & 16)) return TTM_E_UNSUPPORTED;
    if (ncoef < 1 || ncoef > TTM_HOSTCOEF_MAX || (!h_coef_k && !d_coef_k) || !counter || !out) return TTM_E_ARG;
    XCoef hc;
    for (int i = 0; i < TTM_HOSTCOEF_MAX; ++i) hc.c[i] = (h_coef_k && i < ncoef) ? h_coef_k[i] : 0.0;
    const size_t lds = objective_x_lds(p, k, bd);
    const int nfold = p->h_fold_off[k + 1] - p->h_fold_off[k];
    const int ncols = objective_x_cols(p, k);
    const int nch = (p->h_complex[k] >> 24) & 3;            // (sums of an evaluation / 64, rounded up)
#define TTM_CALL(PH, PP, RECT)                                                                                                  \
    do {                                                                                                                       \
        if (nch <= 1) hipLaunchKernelGGL((k_int_objective<PH, PP, RECT, 1>), dim3(grid), dim3(bd), lds, (hipStream_t)stream, P, k, hc, \
                                           d_coef_k, ncoef, Xsoa, ldx, N, nfold, ncols, partial, counter, out, flag, mark);    \
        else hipLaunchKernelGGL((k_int_objective<PH, PP, RECT, XOBJ_NCH>), dim3(grid), dim3(bd), lds, (hipStream_t)stream, P, k, hc,   \
                                d_coef_k, ncoef, Xsoa, ldx, N, nfold, ncols, partial, counter, out, flag, mark);               \
    } while (0)
    TTM_DENSE_DISPATCH(TTM_CALL, cls, p->rectifier);
#undef TTM_CALL
    *kernel_name = "k_int_objective";
    return TTM_OK;
}

}  // namespace ttm_int
