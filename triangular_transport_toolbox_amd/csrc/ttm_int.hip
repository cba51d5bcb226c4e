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
__global__ __launch_bounds__(256) void k_int_objective(DevProg P, int k, const double* __restrict__ coef_k,
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

}  // namespace

namespace ttm_int {

// does any component of [k0, k1) evaluate a special term (erf table in LDS)?
static int needs_erf(const ttm_program* p, int k0, int k1) {
    for (int k = k0; k < k1; ++k)
        if (p->h_complex[k] & 8) return 1;
    return 0;
}

bool usable(const ttm_program* p, int k0, int k1) {
    if (!p || p->monotonicity != TTM_MONO_INTEGRATED || p->family < 0 || p->family > 5) return false;
    DenseClass cls;
    return dense_range_class(p->h_complex, k0, k1, cls);
}

int forward(const ttm_program* p, const DevProg& P, int k0, int k1, const double* coef, const double* fold, const double* Xsoa,
            int64_t ldx, int64_t N, double* Zsoa, int64_t ldz, double* logdet, const double* sigma, double* sumsq, int grid,
            int chunk, int bd, size_t lds, void* stream, const char** kernel_name) {
    DenseClass cls;
    if (!dense_range_class(p->h_complex, k0, k1, cls)) return TTM_E_UNSUPPORTED;
    if (chunk < 1 || logdet || sumsq) chunk = k1 - k0;
    const int erf = needs_erf(p, k0, k1);
    const dim3 g3(grid, (k1 - k0 + chunk - 1) / chunk);
#define TTM_CALL(PH, PP, RECT)                                                                                              \
    do {                                                                                                                    \
        if (logdet) hipLaunchKernelGGL((k_int_forward<PH, PP, RECT, true>), g3, dim3(bd), lds, (hipStream_t)stream, P, k0,  \
                                       k1, chunk, erf, coef, fold, Xsoa, ldx, N, Zsoa, ldz, logdet, sigma, sumsq);          \
        else hipLaunchKernelGGL((k_int_forward<PH, PP, RECT, false>), g3, dim3(bd), lds, (hipStream_t)stream, P, k0, k1,    \
                                chunk, erf, coef, fold, Xsoa, ldx, N, Zsoa, ldz, logdet, sigma, sumsq);                     \
    } while (0)
    TTM_DENSE_DISPATCH(TTM_CALL, cls, p->rectifier);
#undef TTM_CALL
    *kernel_name = "k_int_forward";
    return TTM_OK;
}

int root(const ttm_program* p, const DevProg& P, int k0, int k1, const double* coef, const double* fold, const double* Zsoa,
         int64_t ldz, double* Xsoa, int64_t ldx, int64_t N, int32_t* iters, const int32_t* cap, int newton, int grid, int bd,
         size_t lds, void* stream, const char** kernel_name) {
    DenseClass cls;
    if (!dense_range_class(p->h_complex, k0, k1, cls)) return TTM_E_UNSUPPORTED;
    const int erf = needs_erf(p, k0, k1);
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
    hipLaunchKernelGGL((k_int_objective<PH, PP, RECT>), dim3(grid), dim3(bd), lds, (hipStream_t)stream, P, k, coef_k, fold_k, Xsoa, \
                       ldx, N, nscr, nacc, partial, counter, out, flag, mark)
    TTM_DENSE_DISPATCH(TTM_CALL, cls, p->rectifier);
#undef TTM_CALL
    *kernel_name = "k_int_objective";
    return TTM_OK;
}

}  // namespace ttm_int
