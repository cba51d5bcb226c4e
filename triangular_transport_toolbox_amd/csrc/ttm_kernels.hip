// ttm_kernels.hip - HIP kernels (gfx950 / CDNA4) and the C ABI of libttm.so.
//
// Geometry shared by all kernels: wave64, one thread = one sample, samples
// column-major in HBM so that every column access of a wave is one contiguous
// 512-byte transaction.  The term tables of the components a launch touches
// (int32 records + fp64 constants + coefficients + quadrature rule) are staged
// once per workgroup into LDS; per-sample scratch (the weights w_b of the
// x_k-univariate functions, quadrature partials, gradient accumulators) lives
// in per-thread LDS columns `slot*blockDim + tid` (conflict-free ds_read_b64 /
// ds_write_b64).  Grids are persistent (<= 8 workgroups per CU, grid-stride over
// sample tiles) so the staging cost is paid once per CU slot, not per tile.
// Reductions use a fixed tree (lane-strided partial sums -> wave shuffles ->
// per-block partials -> finishing kernel) and are run-to-run deterministic.
//
// No kernel here has inter-workgroup communication inside a launch, so results
// do not depend on dispatch order or workgroup->XCD placement.

#include <hip/hip_runtime.h>

#include <stdio.h>
#include <string.h>

#include "ttm_eval.h"

using namespace ttm;

// ---------------------------------------------------------------------------
// error handling
// ---------------------------------------------------------------------------

static thread_local char g_err[512] = "";

static int set_err(int code, const char* fmt, const char* a = "", long long b = 0, long long c = 0) {
    snprintf(g_err, sizeof(g_err), fmt, a, b, c);
    return code;
}

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "launch of %s failed: %s", what, hipGetErrorString(e));
        return TTM_E_HIP;
    }
    return TTM_OK;
}

// ---------------------------------------------------------------------------
// device-side helpers
// ---------------------------------------------------------------------------

struct DevProg {             // by-value kernel argument
    const int* itab;
    const double* dpar;
    const double* qx;
    const double* qw;
    int Q, family, mono, rect;
    double delta;
};

struct Stage {               // what to copy into LDS for components [k0,k1)
    int it0, nit;            // int32 range of itab
    int dp0, ndp;            // double range of dpar
    int ncf;                 // number of coefficients (pointer already offset)
    int ncomp;
    int nslots;              // per-thread scratch slots
};

struct LdsSlots {
    double* base;
    int stride;
    __device__ __forceinline__ double get(int i) const { return base[i * stride]; }
    __device__ __forceinline__ void set(int i, double v) { base[i * stride] = v; }
};

struct LdsAcc {
    double* base;
    int stride;
    __device__ __forceinline__ void add(int i, double v) { base[i * stride] += v; }
};

struct XSoA {
    const double* X;
    int64_t ld;
    int64_t n;
    __device__ __forceinline__ double operator()(int var) const { return X[(int64_t)var * ld + n]; }
};

struct XFake {               // TM:4050-4051: zeros except column kc
    int kc;
    double t;
    __device__ __forceinline__ double operator()(int var) const { return var == kc ? t : 0.0; }
};

struct Staged {
    double* slots;           // nslots * blockDim
    const double* dpar;
    const double* coef;
    const int* itab;
    Prog prog;
};

extern __shared__ __align__(16) double g_smem[];

// LDS image: [slots | dpar | coef | quad x | quad w | itab]
__device__ __forceinline__ Staged stage_program(const DevProg& P, const Stage& st, const double* coef) {
    const int tid = threadIdx.x, bd = blockDim.x;
    double* slots = g_smem;
    double* dpar = slots + (size_t)st.nslots * bd;
    double* cf = dpar + st.ndp;
    double* qx = cf + st.ncf;
    double* qw = qx + P.Q;
    int* it = reinterpret_cast<int*>(qw + P.Q);
    for (int i = tid; i < st.ndp; i += bd) dpar[i] = P.dpar[st.dp0 + i];
    for (int i = tid; i < st.ncf; i += bd) cf[i] = coef[i];
    for (int i = tid; i < P.Q; i += bd) { qx[i] = P.qx[i]; qw[i] = P.qw[i]; }
    for (int i = tid; i < st.nit; i += bd) it[i] = P.itab[st.it0 + i];
    __syncthreads();
    Staged s;
    s.slots = slots;
    s.dpar = dpar;
    s.coef = cf;
    s.itab = it;
    s.prog.qx = qx;
    s.prog.qw = qw;
    s.prog.Q = P.Q;
    s.prog.family = P.family;
    s.prog.mono = P.mono;
    s.prog.rect = P.rect;
    s.prog.delta = P.delta;
    return s;
}

// walk the staged component blocks
struct CompCursor {
    const int* cb;
    const double* dp;
    const double* cf;
    __device__ __forceinline__ Comp get() const { return make_comp(cb, dp, cf); }
    __device__ __forceinline__ void next() {
        const int n_nm = TTM_UNI(cb[TTM_HDR_N_NM]), n_mon = TTM_UNI(cb[TTM_HDR_N_MON]);
        dp += TTM_UNI(cb[TTM_HDR_N_DPAR]);
        cf += n_nm + n_mon;
        cb += TTM_UNI(cb[TTM_HDR_LEN_BLK]);
    }
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ---------------------------------------------------------------------------
// K0: layout change with fused (de)standardisation, LDS-tiled 64x64 transpose
// ---------------------------------------------------------------------------

// Xrow: N x d row-major  ->  Xsoa[j*ldx + n] = (x - mean_j) / std_j
__global__ __launch_bounds__(256) void k_import(const double* __restrict__ Xrow, int64_t N, int d,
                                                const double* __restrict__ mean, const double* __restrict__ sd,
                                                double* __restrict__ Xsoa, int64_t ldx) {
    __shared__ double tile[64][65];
    const int64_t n0 = (int64_t)blockIdx.x * 64;
    const int j0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;     // 64 x 4
    for (int r = ty; r < 64; r += 4) {                           // rows = samples, fastest index = column j
        const int64_t n = n0 + r;
        const int j = j0 + tx;
        if (n < N && j < d) tile[r][tx] = Xrow[n * d + j];
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {                           // rows = columns j, fastest index = sample n
        const int j = j0 + r;
        const int64_t n = n0 + tx;
        if (n < N && j < d) {
            double v = tile[tx][r];
            if (mean) v = (v - mean[j]) / sd[j];
            Xsoa[(int64_t)j * ldx + n] = v;
        }
    }
}

// Xsoa columns j0.. -> Xrow[n*dout + j] = x * std + mean
__global__ __launch_bounds__(256) void k_export(const double* __restrict__ Xsoa, int64_t ldx, int64_t N, int j0, int dout,
                                                const double* __restrict__ mean, const double* __restrict__ sd,
                                                double* __restrict__ Xrow) {
    __shared__ double tile[64][65];
    const int64_t n0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int j = c0 + r;
        const int64_t n = n0 + tx;
        if (n < N && j < dout) {
            double v = Xsoa[(int64_t)(j0 + j) * ldx + n];
            if (mean) v = v * sd[j0 + j] + mean[j0 + j];
            tile[r][tx] = v;
        }
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int64_t n = n0 + r;
        const int j = c0 + tx;
        if (n < N && j < dout) Xrow[n * dout + j] = tile[tx][r];
    }
}

// ---------------------------------------------------------------------------
// K1: column statistics of a row-major matrix (two passes, fixed tree)
// pass A: partial sums per block -> mean ; pass B: partial sums of (x-mean)^2 -> std
// ---------------------------------------------------------------------------

#define TTM_STAT_BLOCKS 512

// each block handles a strided set of 256-row slabs; thread (c, r): column c = tid % 64 lanes over columns
__global__ __launch_bounds__(256) void k_colsum(const double* __restrict__ Xrow, int64_t N, int d,
                                                const double* __restrict__ mean, double* __restrict__ partial) {
    // columns are processed in groups of 64 (blockIdx.y); lanes map to columns so that a wave
    // reads 64 consecutive doubles of one row (coalesced), the 4 waves take different rows
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = blockIdx.y * 64 + lane;
    double acc = 0.0;
    if (j < d) {
        const double mu = mean ? mean[j] : 0.0;
        for (int64_t n = (int64_t)blockIdx.x * 4 + wv; n < N; n += (int64_t)gridDim.x * 4) {
            const double v = Xrow[n * d + j] - mu;
            acc += mean ? v * v : v;
        }
    }
    red[wv][lane] = acc;
    __syncthreads();
    if (wv == 0 && j < d) partial[(int64_t)blockIdx.x * d + j] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

__global__ __launch_bounds__(64) void k_colfinish(const double* __restrict__ partial, int nblocks, int d, int64_t N,
                                                  double* __restrict__ out, int take_sqrt) {
    const int j = blockIdx.x;
    double acc = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) acc += partial[(int64_t)b * d + j];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) {
        const double m = acc / (double)N;
        out[j] = take_sqrt ? sqrt(m) : m;
    }
}

// ---------------------------------------------------------------------------
// K2/K3: forward map (+ fused log-determinant)
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_forward(DevProg P, Stage st, const double* __restrict__ coef,
                                                 const double* __restrict__ X, int64_t ldx, int64_t N,
                                                 double* __restrict__ Z, int64_t ldz,
                                                 double* __restrict__ logdet, const double* __restrict__ sigma,
                                                 double* __restrict__ sumsq, int accumulate) {
    const Staged s = stage_program(P, st, coef);
    LdsSlots w{s.slots + threadIdx.x, (int)blockDim.x};
    const bool want_ld = (logdet != nullptr);
    const bool want_val = (Z != nullptr) || (sumsq != nullptr);
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        const XSoA x{X, ldx, n};
        CompCursor cur{s.itab, s.dpar, s.coef};
        double ld = 0.0, ss = 0.0;
        for (int k = 0; k < st.ncomp; ++k, cur.next()) {
            const Comp c = cur.get();
            double S, dS;
            if (want_ld) {
                sample_forward<true>(c, s.prog, x, w, want_val, S, dS);
                ld += log(sigma ? dS / sigma[k] : dS);
            } else {
                sample_forward<false>(c, s.prog, x, w, true, S, dS);
            }
            if (Z) Z[(int64_t)k * ldz + n] = S;
            ss = fma(S, S, ss);
        }
        if (want_ld) logdet[n] = accumulate ? logdet[n] + ld : ld;
        if (sumsq) sumsq[n] = accumulate ? sumsq[n] + ss : ss;
    }
}

// basis matrices of one component
__global__ __launch_bounds__(256) void k_basis(DevProg P, Stage st, int which, const double* __restrict__ X, int64_t ldx,
                                               int64_t N, double* __restrict__ out, int64_t ldo) {
    const Staged s = stage_program(P, st, nullptr);
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        const XSoA x{X, ldx, n};
        const Comp c = make_comp(s.itab, s.dpar, s.coef);
        sample_basis(c, s.prog, which, x, [&](int i, double v) { out[(int64_t)i * ldo + n] = v; });
    }
}

// ---------------------------------------------------------------------------
// K4: table inverse
// ---------------------------------------------------------------------------

// blockIdx.y = component (relative to the staged range start), blockIdx.x over table points
__global__ __launch_bounds__(256) void k_table_build(DevProg P, Stage st, const double* __restrict__ coef,
                                                     const double* __restrict__ pts, int T, double* __restrict__ out) {
    const Staged s = stage_program(P, st, coef);
    LdsSlots w{s.slots + threadIdx.x, (int)blockDim.x};
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    CompCursor cur{s.itab, s.dpar, s.coef};
    for (int k = 0; k < st.ncomp; ++k, cur.next()) {
        const Comp c = cur.get();
        if (i < T) {
            const XFake x{c.kc, pts[i]};
            mon_weights(c, s.prog.family, x, w);
            double g, dg;
            g_eval<false>(c, s.prog.family, pts[i], w, g, dg);
            out[(int64_t)k * T + i] = g;
        }
    }
}

// LDS image: [staged program ... | xs (T) | ys (T)] ; the table region starts at tab_off doubles
__global__ __launch_bounds__(256) void k_inverse_table(DevProg P, Stage st, const double* __restrict__ coef,
                                                       const double* __restrict__ Z, int64_t ldz,
                                                       double* X, int64_t ldx, int64_t N,
                                                       const double* __restrict__ tab_x, const double* __restrict__ tab_y, int T,
                                                       const double* __restrict__ tmin, const double* __restrict__ tmax,
                                                       int truncate, int tab_off) {
    const Staged s = stage_program(P, st, coef);
    double* xs = g_smem + tab_off;
    double* ys = xs + T;
    CompCursor cur{s.itab, s.dpar, s.coef};
    for (int k = 0; k < st.ncomp; ++k, cur.next()) {
        __syncthreads();
        for (int i = threadIdx.x; i < T; i += blockDim.x) {
            xs[i] = tab_x[(int64_t)k * T + i];
            ys[i] = tab_y[(int64_t)k * T + i];
        }
        __syncthreads();
        const Comp c = cur.get();
        const double lo = tmin[k], hi = tmax[k];
        for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
            const XSoA x{X, ldx, n};
            const double off = nonmon_sum(c, s.prog.family, x);
            double target = -off + Z[(int64_t)k * ldz + n];
            if (truncate) {                      // TM:4074-4076 (comparisons keep NaN untouched)
                if (target < lo) target = lo;
                if (target > hi) target = hi;
            }
            X[(int64_t)c.kc * ldx + n] = table_lookup(xs, ys, T, target);
        }
    }
}

// ---------------------------------------------------------------------------
// K5: bisection inverse
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_inverse_bisect(DevProg P, Stage st, const double* __restrict__ coef,
                                                        const double* __restrict__ Z, int64_t ldz,
                                                        double* X, int64_t ldx, int64_t N,
                                                        int* __restrict__ iters, const int* __restrict__ cap) {
    const Staged s = stage_program(P, st, coef);
    LdsSlots w{s.slots + threadIdx.x, (int)blockDim.x};
    for (int64_t n0 = (int64_t)blockIdx.x * blockDim.x; n0 < N; n0 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = n0 + threadIdx.x;
        const bool active = n < N;
        const XSoA x{X, ldx, active ? n : 0};
        CompCursor cur{s.itab, s.dpar, s.coef};
        for (int k = 0; k < st.ncomp; ++k, cur.next()) {
            const Comp c = cur.get();
            int it = 0;
            if (active) {
                const double off = nonmon_sum(c, s.prog.family, x);
                mon_weights(c, s.prog.family, x, w);
                const double r = sample_bisect(c, s.prog, off, Z[(int64_t)k * ldz + n], w, cap ? cap[k] : -1, it);
                X[(int64_t)c.kc * ldx + n] = r;
            }
            // wave-level max, one atomic per wave
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) it = max(it, __shfl_down(it, off, 64));
            if ((threadIdx.x & 63) == 0 && it > 0) atomicMax(&iters[k], it);
        }
    }
}

// ---------------------------------------------------------------------------
// K6/K7: objective + gradient partial sums ; K8: Gram partial sums
// ---------------------------------------------------------------------------

#define TTM_RED_BLOCKS 1024

// LDS: [slots(nscr) | acc(nacc)] per thread columns, then the staged program
__global__ __launch_bounds__(256) void k_objective(DevProg P, Stage st, const double* __restrict__ coef_k,
                                                   const double* __restrict__ X, int64_t ldx, int64_t N,
                                                   int nacc, double* __restrict__ partial) {
    const Staged s = stage_program(P, st, coef_k);
    const int bd = blockDim.x, tid = threadIdx.x;
    const Comp c = make_comp(s.itab, s.dpar, s.coef);
    const int nb1 = c.nB + 1;
    const int nscr = st.nslots - nacc;
    double* accbase = s.slots + (size_t)nscr * bd;
    for (int i = 0; i < nacc; ++i) accbase[i * bd + tid] = 0.0;
    LdsAcc acc{accbase + tid, bd};
    LdsSlots w{s.slots + tid, bd};
    LdsSlots Bv{s.slots + (size_t)nb1 * bd + tid, bd};
    LdsSlots I{s.slots + (size_t)2 * nb1 * bd + tid, bd};
    for (int64_t n = (int64_t)blockIdx.x * bd + tid; n < N; n += (int64_t)gridDim.x * bd) {
        const XSoA x{X, ldx, n};
        if (s.prog.mono == TTM_MONO_SEPARABLE) sample_objective_sep(c, s.prog, x, w, acc);
        else sample_objective_int(c, s.prog, x, w, Bv, I, acc);
    }
    __syncthreads();
    // block reduction: wave wv takes accumulators wv, wv+nw, ... ; lanes stride over threads
    const int lane = tid & 63, wv = tid >> 6, nw = bd >> 6;
    for (int i = wv; i < nacc; i += nw) {
        double v = 0.0;
        for (int t = lane; t < bd; t += 64) v += accbase[i * bd + t];
        v = wave_sum(v);
        if (lane == 0) partial[(int64_t)blockIdx.x * nacc + i] = v;
    }
}

// out[i] = sum_b partial[b*nout + i], one wave per output
__global__ __launch_bounds__(256) void k_reduce_partials(const double* __restrict__ partial, int nblocks, int nout,
                                                         double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nout) return;
    double v = 0.0;
    for (int b = lane; b < nblocks; b += 64) v += partial[(int64_t)b * nout + i];
    v = wave_sum(v);
    if (lane == 0) out[i] = v;
}

#define TTM_GRAM_MAXPAIR 8

// LDS: [scratch (nB+1) | rows (m)] per-thread columns, then the staged program
__global__ __launch_bounds__(256) void k_gram(DevProg P, Stage st, const double* __restrict__ X, int64_t ldx, int64_t N,
                                              int m, double* __restrict__ partial) {
    const Staged s = stage_program(P, st, nullptr);
    const int bd = blockDim.x, tid = threadIdx.x;
    const Comp c = make_comp(s.itab, s.dpar, s.coef);
    double* rows = s.slots + (size_t)(c.nB + 1) * bd;
    const int npair = m * m;
    double g[TTM_GRAM_MAXPAIR];
#pragma unroll
    for (int q = 0; q < TTM_GRAM_MAXPAIR; ++q) g[q] = 0.0;
    for (int64_t n0 = (int64_t)blockIdx.x * bd; n0 < N; n0 += (int64_t)gridDim.x * bd) {
        const int64_t n = n0 + tid;
        if (n < N) {
            const XSoA x{X, ldx, n};
            sample_basis(c, s.prog, 0, x, [&](int i, double v) { rows[i * bd + tid] = v; });
            sample_basis(c, s.prog, 1, x, [&](int i, double v) { rows[(c.n_nm + i) * bd + tid] = v; });
        } else {
            for (int i = 0; i < m; ++i) rows[i * bd + tid] = 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < TTM_GRAM_MAXPAIR; ++q) {
            const int pr = tid + q * bd;
            if (pr < npair) {
                const double* ri = rows + (pr / m) * bd;
                const double* rj = rows + (pr % m) * bd;
                double a = g[q];
                for (int t = 0; t < bd; ++t) {
                    const int tt = (t + tid) & (bd - 1);    // skewed start: conflict-free LDS columns
                    a = fma(ri[tt], rj[tt], a);
                }
                g[q] = a;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < TTM_GRAM_MAXPAIR; ++q) {
        const int pr = tid + q * bd;
        if (pr < npair) partial[(int64_t)blockIdx.x * npair + pr] = g[q];
    }
}

// ---------------------------------------------------------------------------
// host side: launch planning
// ---------------------------------------------------------------------------

static const int kLdsBudget = 64 * 1024;      // bytes per workgroup (2 workgroups/CU still fit in 160 KiB)

static int grid_for(int64_t N, int bd) {
    int64_t tiles = (N + bd - 1) / bd;
    const int64_t cap = 256 * 8;              // 256 CUs x up to 8 resident workgroups
    if (tiles > cap) tiles = cap;
    if (tiles < 1) tiles = 1;
    return (int)tiles;
}

static DevProg dev_prog(const ttm_program* p) {
    DevProg P;
    P.itab = p->itab; P.dpar = p->dpar; P.qx = p->quad_x; P.qw = p->quad_w;
    P.Q = p->Q; P.family = p->family; P.mono = p->monotonicity; P.rect = p->rectifier; P.delta = p->delta;
    return P;
}

static int validate(const ttm_program* p, int k0, int k1) {
    if (!p || !p->itab || !p->h_comp_off || !p->h_dpar_off || !p->h_coef_off || !p->h_nslots)
        return set_err(TTM_E_ARG, "ttm_program has null tables%s");
    if (k0 < 0 || k1 > p->D || k0 >= k1) return set_err(TTM_E_ARG, "component range [%s%lld,%lld) invalid", "", k0, k1);
    if (p->Q < 0 || p->Q > 4096) return set_err(TTM_E_ARG, "quadrature order %s%lld out of range", "", p->Q);
    if (p->monotonicity == TTM_MONO_INTEGRATED && (p->Q < 1 || !p->quad_x || !p->quad_w))
        return set_err(TTM_E_ARG, "integrated rectifier needs quadrature nodes%s");
    return TTM_OK;
}

// stage description + LDS bytes for components [ka,kb) with `extra_slots` per-thread slots on top of max(nB+1)
static Stage make_stage(const ttm_program* p, int ka, int kb, int slots_mult, int extra_slots) {
    Stage st;
    st.it0 = p->h_comp_off[ka]; st.nit = p->h_comp_off[kb] - st.it0;
    st.dp0 = p->h_dpar_off[ka]; st.ndp = p->h_dpar_off[kb] - st.dp0;
    st.ncf = p->h_coef_off[kb] - p->h_coef_off[ka];
    st.ncomp = kb - ka;
    int ns = 0;
    for (int k = ka; k < kb; ++k) ns = p->h_nslots[k] > ns ? p->h_nslots[k] : ns;
    st.nslots = ns * slots_mult + extra_slots;
    return st;
}

static size_t lds_bytes(const ttm_program* p, const Stage& st, int bd, int extra_doubles) {
    size_t dbl = (size_t)st.nslots * bd + st.ndp + st.ncf + 2 * (size_t)p->Q + extra_doubles;
    return dbl * 8 + (size_t)st.nit * 4;
}

// pick the largest block size whose LDS image fits; 0 if none
static int pick_block(const ttm_program* p, const Stage& st, int extra_doubles) {
    for (int bd = 256; bd >= 64; bd >>= 1)
        if (lds_bytes(p, st, bd, extra_doubles) <= (size_t)kLdsBudget) return bd;
    return 0;
}

// greedy split of [k0,k1) into chunks whose program fits the LDS budget at blockDim 256
// (a single oversized component falls back to smaller blocks)
template <class Fn>
static int for_each_chunk(const ttm_program* p, int k0, int k1, int slots_mult, int extra_slots, int extra_doubles, Fn fn) {
    int ka = k0;
    while (ka < k1) {
        int kb = ka + 1;
        while (kb < k1) {
            Stage st = make_stage(p, ka, kb + 1, slots_mult, extra_slots);
            if (lds_bytes(p, st, 256, extra_doubles) > (size_t)kLdsBudget) break;
            ++kb;
        }
        Stage st = make_stage(p, ka, kb, slots_mult, extra_slots);
        int bd = pick_block(p, st, extra_doubles);
        if (!bd) return set_err(TTM_E_LIMIT, "component %s%lld does not fit the LDS budget", "", ka);
        int rc = fn(ka, kb, st, bd);
        if (rc != TTM_OK) return rc;
        ka = kb;
    }
    return TTM_OK;
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------

extern "C" {

const char* ttm_last_error_string(void) { return g_err; }

int ttm_version(void) { return TTM_VERSION; }

int ttm_device_count(int* count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        n = 0;
        snprintf(g_err, sizeof(g_err), "hipGetDeviceCount: %s", hipGetErrorString(e));
        (void)hipGetLastError();
    }
    if (count) *count = n;
    return n > 0 ? TTM_OK : TTM_E_HIP;
}

int64_t ttm_colstats_work_size(int64_t N, int32_t d) { (void)N; return (int64_t)TTM_STAT_BLOCKS * d; }

int ttm_colstats(const double* Xrow, int64_t N, int32_t d, double* mean, double* sd, double* work, void* stream) {
    if (!Xrow || !mean || !sd || !work || N < 1 || d < 1) return set_err(TTM_E_ARG, "ttm_colstats: bad arguments%s");
    hipStream_t s = (hipStream_t)stream;
    int nb = (int)((N + 3) / 4 < TTM_STAT_BLOCKS ? (N + 3) / 4 : TTM_STAT_BLOCKS);
    dim3 grid(nb, (d + 63) / 64);
    hipLaunchKernelGGL(k_colsum, grid, dim3(256), 0, s, Xrow, N, (int)d, (const double*)nullptr, work);
    hipLaunchKernelGGL(k_colfinish, dim3(d), dim3(64), 0, s, work, nb, (int)d, N, mean, 0);
    hipLaunchKernelGGL(k_colsum, grid, dim3(256), 0, s, Xrow, N, (int)d, (const double*)mean, work);
    hipLaunchKernelGGL(k_colfinish, dim3(d), dim3(64), 0, s, work, nb, (int)d, N, sd, 1);
    return check_launch("k_colsum/k_colfinish");
}

int ttm_import(const double* Xrow, int64_t N, int32_t d, const double* mean, const double* sd, double* Xsoa,
               int64_t ldx, void* stream) {
    if (!Xrow || !Xsoa || N < 1 || d < 1 || ldx < N) return set_err(TTM_E_ARG, "ttm_import: bad arguments%s");
    if ((mean == nullptr) != (sd == nullptr)) return set_err(TTM_E_ARG, "ttm_import: mean and std must both be given%s");
    dim3 grid((unsigned)((N + 63) / 64), (d + 63) / 64);
    hipLaunchKernelGGL(k_import, grid, dim3(256), 0, (hipStream_t)stream, Xrow, N, (int)d, mean, sd, Xsoa, ldx);
    return check_launch("k_import");
}

int ttm_export(const double* Xsoa, int64_t ldx, int64_t N, int32_t j0, int32_t dout, const double* mean,
               const double* sd, double* Xrow, void* stream) {
    if (!Xrow || !Xsoa || N < 1 || dout < 1 || j0 < 0 || ldx < N) return set_err(TTM_E_ARG, "ttm_export: bad arguments%s");
    if ((mean == nullptr) != (sd == nullptr)) return set_err(TTM_E_ARG, "ttm_export: mean and std must both be given%s");
    dim3 grid((unsigned)((N + 63) / 64), (dout + 63) / 64);
    hipLaunchKernelGGL(k_export, grid, dim3(256), 0, (hipStream_t)stream, Xsoa, ldx, N, (int)j0, (int)dout, mean, sd, Xrow);
    return check_launch("k_export");
}

int ttm_forward(const ttm_program* p, const double* coef, const double* Xsoa, int64_t ldx, int64_t N, int32_t k0,
                int32_t k1, double* Zsoa, int64_t ldz, double* logdet, const double* sigma, double* sumsq, void* stream) {
    int rc = validate(p, k0, k1);
    if (rc) return rc;
    if (!coef || !Xsoa || N < 1 || ldx < N || (Zsoa && ldz < N) || (!Zsoa && !logdet && !sumsq))
        return set_err(TTM_E_ARG, "ttm_forward: bad arguments%s");
    const DevProg P = dev_prog(p);
    return for_each_chunk(p, k0, k1, 1, 0, 0, [&](int ka, int kb, Stage st, int bd) {
        hipLaunchKernelGGL(k_forward, dim3(grid_for(N, bd)), dim3(bd), lds_bytes(p, st, bd, 0), (hipStream_t)stream, P, st,
                           coef + p->h_coef_off[ka], Xsoa, ldx, N, Zsoa ? Zsoa + (int64_t)(ka - k0) * ldz : nullptr, ldz,
                           logdet, sigma ? sigma + (ka - k0) : nullptr, sumsq, ka > k0 ? 1 : 0);
        return check_launch("k_forward");
    });
}

int ttm_basis(const ttm_program* p, int32_t k, int32_t which, const double* Xsoa, int64_t ldx, int64_t N, double* out,
              int64_t ldo, void* stream) {
    int rc = validate(p, k, k + 1);
    if (rc) return rc;
    if (!Xsoa || !out || N < 1 || ldx < N || ldo < N || which < 0 || which > 2) return set_err(TTM_E_ARG, "ttm_basis: bad arguments%s");
    Stage st = make_stage(p, k, k + 1, 1, 0);
    st.ncf = 0;
    int bd = pick_block(p, st, 0);
    if (!bd) return set_err(TTM_E_LIMIT, "component %s%lld does not fit the LDS budget", "", k);
    hipLaunchKernelGGL(k_basis, dim3(grid_for(N, bd)), dim3(bd), lds_bytes(p, st, bd, 0), (hipStream_t)stream, dev_prog(p), st,
                       (int)which, Xsoa, ldx, N, out, ldo);
    return check_launch("k_basis");
}

int ttm_inverse_table_build(const ttm_program* p, const double* coef, int32_t k0, int32_t k1, const double* pts, int32_t T,
                            double* out, void* stream) {
    int rc = validate(p, k0, k1);
    if (rc) return rc;
    if (!coef || !pts || !out || T < 2) return set_err(TTM_E_ARG, "ttm_inverse_table_build: bad arguments%s");
    const DevProg P = dev_prog(p);
    DevProg Psep = P;
    Psep.mono = TTM_MONO_SEPARABLE;
    return for_each_chunk(p, k0, k1, 1, 0, 0, [&](int ka, int kb, Stage st, int bd) {
        hipLaunchKernelGGL(k_table_build, dim3((T + bd - 1) / bd), dim3(bd), lds_bytes(p, st, bd, 0), (hipStream_t)stream, Psep, st,
                           coef + p->h_coef_off[ka], pts, (int)T, out + (int64_t)(ka - k0) * T);
        return check_launch("k_table_build");
    });
}

int ttm_inverse_table(const ttm_program* p, const double* coef, int32_t k0, int32_t k1, const double* Zsoa, int64_t ldz,
                      double* Xsoa, int64_t ldx, int64_t N, const double* tab_x, const double* tab_y, int32_t T,
                      const double* tmin, const double* tmax, int32_t truncate, void* stream) {
    int rc = validate(p, k0, k1);
    if (rc) return rc;
    if (!coef || !Zsoa || !Xsoa || !tab_x || !tab_y || !tmin || !tmax || N < 1 || ldx < N || ldz < N || T < 2 || T > 2048)
        return set_err(TTM_E_ARG, "ttm_inverse_table: bad arguments%s");
    if (p->monotonicity != TTM_MONO_SEPARABLE) return set_err(TTM_E_UNSUPPORTED, "table inverse needs separable monotonicity%s");
    const DevProg P = dev_prog(p);
    return for_each_chunk(p, k0, k1, 0, 0, 2 * T, [&](int ka, int kb, Stage st, int bd) {
        const size_t bytes = lds_bytes(p, st, bd, 2 * T);
        const int tab_off = st.nslots * bd + st.ndp + st.ncf + 2 * p->Q;
        // the int table follows the doubles; place xs/ys after it, 8-byte aligned
        const int tab_off_d = tab_off + (st.nit + 1) / 2;
        const size_t total = (size_t)(tab_off_d + 2 * T) * 8;
        (void)bytes;
        hipLaunchKernelGGL(k_inverse_table, dim3(grid_for(N, bd)), dim3(bd), total, (hipStream_t)stream, P, st,
                           coef + p->h_coef_off[ka], Zsoa + (int64_t)(ka - k0) * ldz, ldz, Xsoa, ldx, N,
                           tab_x + (int64_t)(ka - k0) * T, tab_y + (int64_t)(ka - k0) * T, (int)T, tmin + (ka - k0), tmax + (ka - k0),
                           (int)truncate, tab_off_d);
        return check_launch("k_inverse_table");
    });
}

int ttm_inverse_bisect(const ttm_program* p, const double* coef, int32_t k0, int32_t k1, const double* Zsoa, int64_t ldz,
                       double* Xsoa, int64_t ldx, int64_t N, int32_t* iters, const int32_t* cap, void* stream) {
    int rc = validate(p, k0, k1);
    if (rc) return rc;
    if (!coef || !Zsoa || !Xsoa || !iters || N < 1 || ldx < N || ldz < N) return set_err(TTM_E_ARG, "ttm_inverse_bisect: bad arguments%s");
    const DevProg P = dev_prog(p);
    return for_each_chunk(p, k0, k1, 1, 0, 0, [&](int ka, int kb, Stage st, int bd) {
        hipLaunchKernelGGL(k_inverse_bisect, dim3(grid_for(N, bd)), dim3(bd), lds_bytes(p, st, bd, 0), (hipStream_t)stream, P, st,
                           coef + p->h_coef_off[ka], Zsoa + (int64_t)(ka - k0) * ldz, ldz, Xsoa, ldx, N, iters + (ka - k0),
                           cap ? cap + (ka - k0) : nullptr);
        return check_launch("k_inverse_bisect");
    });
}

int64_t ttm_reduce_work_size(int32_t nout) { return (int64_t)TTM_RED_BLOCKS * (nout > 0 ? nout : 1); }

int ttm_objective(const ttm_program* p, int32_t k, const double* coef_k, const double* Xsoa, int64_t ldx, int64_t N,
                  double* work, double* out, void* stream) {
    int rc = validate(p, k, k + 1);
    if (rc) return rc;
    if (!coef_k || !Xsoa || !work || !out || N < 1 || ldx < N) return set_err(TTM_E_ARG, "ttm_objective: bad arguments%s");
    if (p->monotonicity == TTM_MONO_INTEGRATED && p->rectifier != TTM_RECT_EXPONENTIAL && p->rectifier != TTM_RECT_SOFTPLUS &&
        p->rectifier != TTM_RECT_EXPNEG)
        return set_err(TTM_E_UNSUPPORTED, "rectifier has no evaluate_dfdc in the reference (TM:5119-5163)%s");
    const int sep = p->monotonicity == TTM_MONO_SEPARABLE;
    const int n_nm = p->h_n_nm[k];
    const int n_mon = p->h_coef_off[k + 1] - p->h_coef_off[k] - n_nm;
    const int nacc = sep ? 1 + n_mon : 1 + n_nm + n_mon;
    Stage st = make_stage(p, k, k + 1, sep ? 1 : 3, nacc);
    const int bd = pick_block(p, st, 0);
    if (!bd) return set_err(TTM_E_LIMIT, "component %s%lld does not fit the LDS budget", "", k);
    int nb = grid_for(N, bd);
    if (nb > TTM_RED_BLOCKS) nb = TTM_RED_BLOCKS;
    hipLaunchKernelGGL(k_objective, dim3(nb), dim3(bd), lds_bytes(p, st, bd, 0), (hipStream_t)stream, dev_prog(p), st, coef_k,
                       Xsoa, ldx, N, nacc, work);
    hipLaunchKernelGGL(k_reduce_partials, dim3((nacc + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const double*)work, nb, nacc, out);
    return check_launch("k_objective");
}

int ttm_gram(const ttm_program* p, int32_t k, const double* Xsoa, int64_t ldx, int64_t N, double* work, double* out,
             void* stream) {
    int rc = validate(p, k, k + 1);
    if (rc) return rc;
    if (!Xsoa || !work || !out || N < 1 || ldx < N) return set_err(TTM_E_ARG, "ttm_gram: bad arguments%s");
    const int m = p->h_coef_off[k + 1] - p->h_coef_off[k];
    Stage st = make_stage(p, k, k + 1, 1, m);
    st.ncf = 0;
    int bd = pick_block(p, st, 0);
    while (bd && m * m > TTM_GRAM_MAXPAIR * bd) bd = 0;
    if (!bd) return set_err(TTM_E_LIMIT, "ttm_gram: %s%lld basis functions exceed the kernel limits", "", m);
    int nb = grid_for(N, bd);
    if (nb > TTM_RED_BLOCKS) nb = TTM_RED_BLOCKS;
    hipLaunchKernelGGL(k_gram, dim3(nb), dim3(bd), lds_bytes(p, st, bd, 0), (hipStream_t)stream, dev_prog(p), st, Xsoa, ldx, N, m, work);
    hipLaunchKernelGGL(k_reduce_partials, dim3((m * m + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const double*)work, nb, m * m, out);
    return check_launch("k_gram");
}

}  // extern "C"
