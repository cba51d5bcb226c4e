// ttm_kernels.hip - HIP kernels (gfx950 / CDNA4) and the C ABI of libttm.so.
//
// Geometry shared by all kernels: wave64, one thread = one sample, samples
// column-major in HBM so that every column access of a wave is one contiguous
// 512-byte transaction.  Everything that is identical for all lanes - term
// tables, fp64 constants, coefficients, folded coefficients, the quadrature
// rule - is read through constant-address-space pointers, i.e. by scalar loads
// into SGPRs (the scalar unit runs the table interpreter, the VALU only does
// per-sample fp64 math).  LDS holds what lanes index individually: the erf
// interpolation table (3 KB) and per-sample scratch in per-thread columns
// `slot*blockDim + tid` (weights w_b of the x_k-univariate functions for
// components with cross terms, quadrature partials, gradient accumulators),
// plus the 1001-point inverse table of the table root search.
// Reductions use a fixed tree (lane-strided partial sums -> wave shuffles ->
// per-block partials -> finishing kernel) and are run-to-run deterministic.
//
// No kernel here has inter-workgroup communication inside a launch, so results
// do not depend on dispatch order or workgroup->XCD placement.

#include <hip/hip_runtime.h>
#include <mutex>
#include <setjmp.h>
#include <signal.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "ttm_eval.h"
#include "ttm_dev.h"
#include "ttm_rng.h"
#include "ttm_uform.h"
#include "ttm_band.h"
#include "ttm_band_image.h"
#include "ttm_int.h"

using namespace ttm;

// ---------------------------------------------------------------------------
// error handling
// ---------------------------------------------------------------------------

static thread_local char g_err[512] = "";

static int set_err(int code, const char* fmt, const char* a = "", long long b = 0, long long c = 0) {
    snprintf(g_err, sizeof(g_err), fmt, a, b, c);
    return code;
}

static thread_local const char* g_last_kernel = "";      // name of the kernel the last entry point launched

static int check_launch(const char* what) {
    g_last_kernel = what;
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

// ---------------------------------------------------------------------------
// folded coefficients: block k handles component kfirst + k, one thread per slot
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(64) void k_fold(DevProg P, int kfirst, int kbase, const double* __restrict__ coef,
                                             double* __restrict__ fold) {
    const int k = kfirst + blockIdx.x;
    const int D1 = P.D + 1;
    const int* off = P.off;
    fold_coeffs(P.itab + off[k], P.ftab + off[4 * D1 + k], P.dpar + off[D1 + k],
                coef + (off[2 * D1 + k] - off[2 * D1 + kbase]), fold + (off[3 * D1 + k] - off[3 * D1 + kbase]),
                threadIdx.x, blockDim.x);
    __syncthreads();
    fold_st8(P.fdesc + k * TTM_FDESC_LEN, P.fints, fold + (off[3 * D1 + k] - off[3 * D1 + kbase]), threadIdx.x, blockDim.x);
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

// The same for matrices of at most 64 columns - every map of the reference's examples and benchmarks: a slab of 64 rows is
// 64 d CONSECUTIVE doubles of the row-major matrix, so it is read (written) as a flat stream with every lane busy; the
// tiled kernels above give a column to a lane (d = 40: 24 of 64 lanes idle; d = 4 - the filter's map input - 60 of 64).
// tile[c][r] with a row stride of 65 doubles: the flat side walks c fastest, the column side r fastest, both conflict-free.
__global__ __launch_bounds__(256) void k_import_flat(const double* __restrict__ Xrow, int64_t N, int d,
                                                     const double* __restrict__ mean, const double* __restrict__ sd,
                                                     double* __restrict__ Xsoa, int64_t ldx) {
    __shared__ double tile[64 * 65];
    const int64_t n0 = (int64_t)blockIdx.x * 64;
    const int rows = (int)(N - n0 < 64 ? N - n0 : 64), total = rows * d, tid = threadIdx.x;
    const double* src = Xrow + n0 * d;
    const int dq = 256 / d, dr = 256 - dq * d;                  // 256 = dq d + dr: how (row, column) moves from one pass to the next
    int r = tid / d, c = tid - r * d;
    for (int e = tid; e < total; e += 256) {
        tile[c * 65 + r] = src[e];
        r += dq; c += dr;
        if (c >= d) { c -= d; ++r; }
    }
    __syncthreads();
    for (int idx = tid; idx < d * 64; idx += 256) {
        const int cc = idx >> 6, rr = idx & 63;
        if (rr < rows) {
            double v = tile[cc * 65 + rr];
            if (mean) v = (v - mean[cc]) / sd[cc];
            Xsoa[(int64_t)cc * ldx + n0 + rr] = v;
        }
    }
}

__global__ __launch_bounds__(256) void k_export_flat(const double* __restrict__ Xsoa, int64_t ldx, int64_t N, int j0, int dout,
                                                     const double* __restrict__ mean, const double* __restrict__ sd,
                                                     double* __restrict__ Xrow) {
    __shared__ double tile[64 * 65];
    const int64_t n0 = (int64_t)blockIdx.x * 64;
    const int rows = (int)(N - n0 < 64 ? N - n0 : 64), total = rows * dout, tid = threadIdx.x;
    for (int idx = tid; idx < dout * 64; idx += 256) {
        const int cc = idx >> 6, rr = idx & 63;
        if (rr < rows) {
            double v = Xsoa[(int64_t)(j0 + cc) * ldx + n0 + rr];
            if (mean) v = v * sd[j0 + cc] + mean[j0 + cc];
            tile[cc * 65 + rr] = v;
        }
    }
    __syncthreads();
    double* dst = Xrow + n0 * dout;
    const int dq = 256 / dout, dr = 256 - dq * dout;
    int r = tid / dout, c = tid - r * dout;
    for (int e = tid; e < total; e += 256) {
        dst[e] = tile[c * 65 + r];
        r += dq; c += dr;
        if (c >= dout) { c -= dout; ++r; }
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

// Column moments of up to 8 columns and 131 072 rows in ONE launch (the filter's ensembles; the four launches above are 34 us
// of an update and leave 60 of 64 lanes idle at d = 4): a thread keeps its <= 4 rows in registers, the workgroup takes its own
// mean and the squared deviations from it (two passes over registers), and the workgroup that draws the last ticket combines the
// rows {count, mean_b, M2_b} of all workgroups (Chan et al.: M2 = sum M2_b + count_b (mean_b - mean)^2) in workgroup order - no
// cancellation whatever the offset of the data, run-to-run deterministic.  COLS: column-major input X[j ld + n], else row-major
// X[n D + j]; both layouts make the same sums.  counter: one uint32, zero (the host clears it in front of the launch).
#define TTM_CS1_ROWS 4
#define TTM_CS1_DMAX 8
#define TTM_CS1_WGS 128

template <int D, bool COLS>
__global__ __launch_bounds__(256) void k_colstats_one(const double* __restrict__ X, int64_t ld, int64_t N, double* __restrict__ partial,
                                                      unsigned int* __restrict__ counter, double* __restrict__ mean, double* __restrict__ sd) {
    constexpr int NS = 1 + 2 * D;
    __shared__ double red[4][D + 1];
    __shared__ double mb[D + 1];
    __shared__ double all[TTM_CS1_WGS * NS];
    __shared__ int verdict;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double v[TTM_CS1_ROWS][D];
    bool have[TTM_CS1_ROWS];
#pragma unroll
    for (int r = 0; r < TTM_CS1_ROWS; ++r) {
        const int64_t n = ((int64_t)r * gridDim.x + blockIdx.x) * 256 + tid;
        have[r] = n < N;
#pragma unroll
        for (int j = 0; j < D; ++j) v[r][j] = have[r] ? (COLS ? X[(int64_t)j * ld + n] : X[n * D + j]) : 0.0;
    }
    double s[D + 1];
#pragma unroll
    for (int j = 0; j <= D; ++j) s[j] = 0.0;
#pragma unroll
    for (int r = 0; r < TTM_CS1_ROWS; ++r)
        if (have[r]) {
#pragma unroll
            for (int j = 0; j < D; ++j) s[j] += v[r][j];
            s[D] += 1.0;
        }
#pragma unroll
    for (int j = 0; j <= D; ++j) {
        const double t = wave_sum(s[j]);
        if (lane == 0) red[wv][j] = t;
    }
    __syncthreads();
    if (tid <= D) {
        const double cnt = (red[0][D] + red[1][D]) + (red[2][D] + red[3][D]);
        const double tot = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        mb[tid] = tid < D ? tot / cnt : cnt;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < D; ++j) s[j] = 0.0;
#pragma unroll
    for (int r = 0; r < TTM_CS1_ROWS; ++r)
        if (have[r]) {
#pragma unroll
            for (int j = 0; j < D; ++j) { const double dl = v[r][j] - mb[j]; s[j] += dl * dl; }
        }
#pragma unroll
    for (int j = 0; j < D; ++j) {
        const double t = wave_sum(s[j]);
        if (lane == 0) red[wv][j] = t;
    }
    __syncthreads();
    if (tid < NS) {
        const int j = tid - 1 - D;
        const double val = tid == 0 ? mb[D] : tid <= D ? mb[tid - 1] : (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
        coherent_store(partial + (int64_t)blockIdx.x * NS + tid, val);
    }
    drain_stores();
    __syncthreads();
    if (tid == 0) verdict = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u ? 1 : 0;
    __syncthreads();
    if (!verdict) return;
    if (tid == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int nb = (int)gridDim.x;
    for (int i = tid; i < nb * NS; i += 256) all[i] = coherent_load(partial + i);
    __syncthreads();
    // column j by wave j mod 4: lanes = workgroups (a serial walk over 98 rows of LDS costs more than the rest of the kernel)
    for (int j = wv; j < D; j += 4) {
        double sw = 0.0;
        for (int b = lane; b < nb; b += 64) sw += all[b * NS] * all[b * NS + 1 + j];
        const double mu = __shfl(wave_sum(sw), 0, 64) / (double)N;
        double m2 = 0.0;
        for (int b = lane; b < nb; b += 64) {
            const double dl = all[b * NS + 1 + j] - mu;
            m2 += all[b * NS + 1 + D + j] + all[b * NS] * (dl * dl);
        }
        m2 = wave_sum(m2);
        if (lane == 0) {
            mean[j] = mu;
            sd[j] = sqrt(m2 / (double)N);
        }
    }
}

// Xs[j ldx + n] = (X[j ld + n] - mean[j]) / sd[j]: standardisation of a column-major matrix that is already on the device
__global__ __launch_bounds__(256) void k_standardize_cols(const double* __restrict__ X, int64_t ld, int64_t N, int d,
                                                          const double* __restrict__ mean, const double* __restrict__ sd,
                                                          double* __restrict__ Xs, int64_t ldx) {
    const int j = blockIdx.y;
    const double mu = mean[j], sg = sd[j];
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < N; n += (int64_t)gridDim.x * 256)
        Xs[(int64_t)j * ldx + n] = (X[(int64_t)j * ld + n] - mu) / sg;
}

// ---------------------------------------------------------------------------
// K9: exact order statistics by radix select (8 passes of 8 bits over order-preserving 64-bit keys)
// state per requested rank: prefix (determined high bits), remaining rank; hist: nr x 256 counters
// ---------------------------------------------------------------------------

#define TTM_SEL_MAX 16

__device__ __forceinline__ unsigned long long f64_key(double x) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);      // total order: -inf < ... < -0 < +0 < ... < +inf < NaN
}

struct SelState {
    unsigned long long prefix[TTM_SEL_MAX];
    long long rank[TTM_SEL_MAX];
};

__global__ __launch_bounds__(256) void k_select_init(const long long* __restrict__ ranks, int nr, SelState* st,
                                                     unsigned int* __restrict__ hist) {
    for (int i = threadIdx.x; i < nr * 256; i += blockDim.x) hist[i] = 0u;
    if (threadIdx.x < nr) { st->prefix[threadIdx.x] = 0ull; st->rank[threadIdx.x] = ranks[threadIdx.x]; }
}

// histogram of byte `shift/8` over the elements whose higher bytes equal the rank's prefix
__global__ __launch_bounds__(256) void k_select_hist(const double* __restrict__ col, int64_t N, int nr, int shift,
                                                     const SelState* __restrict__ st, unsigned int* __restrict__ hist) {
    __shared__ unsigned int lh[TTM_SEL_MAX * 256];
    __shared__ unsigned long long pf[TTM_SEL_MAX];
    for (int i = threadIdx.x; i < nr * 256; i += blockDim.x) lh[i] = 0u;
    if (threadIdx.x < nr) pf[threadIdx.x] = st->prefix[threadIdx.x];
    __syncthreads();
    const unsigned long long himask = (shift == 56) ? 0ull : (~0ull << (shift + 8));
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long key = f64_key(col[n]);
        const unsigned int bin = (unsigned int)((key >> shift) & 255ull);
        for (int j = 0; j < nr; ++j)
            if (((key ^ pf[j]) & himask) == 0ull) atomicAdd(&lh[j * 256 + bin], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nr * 256; i += blockDim.x)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// choose the bin that holds the requested rank, extend the prefix, clear the histogram
// (one workgroup per requested rank, one thread per bin, LDS inclusive scan)
__global__ __launch_bounds__(256) void k_select_pick(int nr, int shift, SelState* st, unsigned int* __restrict__ hist,
                                                     double* __restrict__ out) {
    __shared__ long long cum[256];
    const int j = blockIdx.x, t = threadIdx.x;
    unsigned int* h = hist + j * 256;
    const long long mine = (long long)h[t];
    cum[t] = mine;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const long long add = t >= d ? cum[t - d] : 0;
        __syncthreads();
        cum[t] += add;
        __syncthreads();
    }
    const long long r = st->rank[j];
    const long long before = cum[t] - mine;
    const bool last_nonempty_guard = (t == 255);
    if ((r >= before && r < cum[t]) || (last_nonempty_guard && r >= cum[255])) {
        st->rank[j] = r - before;
        const unsigned long long pfx = st->prefix[j] | ((unsigned long long)t << shift);
        st->prefix[j] = pfx;
        if (shift == 0) {
            const unsigned long long u = (pfx >> 63) ? (pfx & 0x7fffffffffffffffull) : ~pfx;
            out[j] = __longlong_as_double((long long)u);
        }
    }
    h[t] = 0u;
}

// The same select in ONE launch for columns of up to 131 072 rows (the filter's ensembles: the 17 launches above are 117 us of
// an update, each pass a launch floor).  Every thread keeps its <= 8 keys in registers, the workgroups meet at a grid barrier
// per pass (agent-scope atomics; at most 64 workgroups of 256 threads: one per CU on a quarter of the chip), and every
// workgroup picks the bins itself from the summed histogram.  Ranks that still share a prefix share a histogram row (the two
// neighbours a quantile interpolates between do until the last passes).  As soon as every rank's bin holds at most 32 keys
// (three passes for 1e5 values of a continuous distribution) the passes end: the keys of those bins are gathered into lists
// and workgroup 0 ranks them by counting - four barriers instead of eight (a pass is four device-scope round trips, ~11 us).
//   ghist: three regions of TTM_SEL_MAX x 256 counters - pass p adds into region p % 3 and clears region (p + 1) % 3, which
//          was last read before barrier p - 1;
//   bar:   [0] arrivals, monotone within a launch, [16 + j] length of candidate list j; bar and region 0 are cleared by the
//          host (ONE memset node in front of the launch: `work` needs no initial state, and a launch that was cut short leaves
//          nothing behind);
//   cand:  TTM_SEL_MAX lists of 32 keys.
// Every wait is bounded: a workgroup whose partners do not arrive (a grid that is not co-resident because someone else holds
// the CUs) stops waiting; workgroup 0 then does the selection by itself from the column in memory (select_solo) - slower,
// same result -, so the grid always drains and `out` is always the exact order statistic.
#define TTM_SELC_ROWS 8
#define TTM_SELC_WGS 64
#define TTM_SELC_SPINS (1 << 17)
#define TTM_SELC_CAND 32

__device__ __forceinline__ void coherent_store_u32(unsigned int* p, unsigned int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned int coherent_load_u32(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// all threads of the workgroup call; wait = false: arrive only.  false: the others did not arrive within the bound
__device__ __forceinline__ bool grid_barrier(unsigned int* bar, unsigned int target, int* verdict, bool wait = true) {
    drain_stores();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = wait ? 0 : 1;
        for (int spin = 0; wait && spin < TTM_SELC_SPINS; ++spin)
            if (coherent_load_u32(bar) >= target) { ok = 1; break; }
        *verdict = ok;
    }
    __syncthreads();
    return *verdict != 0;
}

// bin of rank j from histogram row h (256 counters, read by `load`): one wave, four bins per lane; the lane that holds the
// bin updates prefix and remaining rank (the guard of k_select_pick: a rank beyond the total lands in the last bin) and
// notes how many keys the bin holds
template <typename Load>
__device__ __forceinline__ void wave_pick(Load load, int shift, unsigned long long* pf, long long* rk, long long* held) {
    const int lane = threadIdx.x & 63;
    long long c[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) c[q] = (long long)load(lane * 4 + q);
    const long long mine = (c[0] + c[1]) + (c[2] + c[3]);
    long long incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    const long long total = __shfl(incl, 63, 64);
    const long long r = *rk;
    long long before = incl - mine;
    const bool here = (r >= before && r < incl) || (lane == 63 && r >= total);
    // (every lane has read *rk before any lane writes it: the wave runs in lock step, and the writes sit behind the reads)
    if (here) {
        int q = 0;
        while (q < 3 && r >= before + c[q]) { before += c[q]; ++q; }
        *rk = r - before;
        *pf = *pf | ((unsigned long long)(lane * 4 + q) << shift);
        *held = c[q];
    }
}

__device__ __forceinline__ double key_f64(unsigned long long pfx) {
    const unsigned long long u = (pfx >> 63) ? (pfx & 0x7fffffffffffffffull) : ~pfx;
    return __longlong_as_double((long long)u);
}

// the whole selection by ONE workgroup from the column in memory (the fallback of k_select_coop)
__device__ void select_solo(const double* __restrict__ col, int64_t N, const long long* __restrict__ ranks, int nr,
                            unsigned int* lh, unsigned long long* pf, long long* rk, long long* held, double* __restrict__ out) {
    const int tid = threadIdx.x, wv = tid >> 6;
    __syncthreads();
    if (tid < nr) { pf[tid] = 0ull; rk[tid] = ranks[tid]; }
    for (int shift = 56; shift >= 0; shift -= 8) {
        for (int i = tid; i < nr * 256; i += blockDim.x) lh[i] = 0u;
        __syncthreads();
        const unsigned long long himask = (shift == 56) ? 0ull : (~0ull << (shift + 8));
        for (int64_t n = tid; n < N; n += blockDim.x) {
            const unsigned long long key = f64_key(col[n]);
            const unsigned int bin = (unsigned int)((key >> shift) & 255ull);
            for (int j = 0; j < nr; ++j)
                if (((key ^ pf[j]) & himask) == 0ull) atomicAdd(&lh[j * 256 + bin], 1u);
        }
        __syncthreads();
        for (int j = wv; j < nr; j += (int)(blockDim.x >> 6)) {
            const unsigned int* h = lh + j * 256;
            wave_pick([&](int b) { return h[b]; }, shift, pf + j, rk + j, held + j);
        }
        __syncthreads();
    }
    if (tid < nr) out[tid] = key_f64(pf[tid]);
}

__global__ __launch_bounds__(256) void k_select_coop(const double* __restrict__ col, int64_t N, const long long* __restrict__ ranks,
                                                     int nr, unsigned int* __restrict__ ghist, unsigned int* __restrict__ bar,
                                                     unsigned long long* __restrict__ cand, double* __restrict__ out, int give_up) {
    __shared__ unsigned int lh[TTM_SEL_MAX * 256];
    __shared__ unsigned long long pf[TTM_SEL_MAX];
    __shared__ long long rk[TTM_SEL_MAX];
    __shared__ long long held[TTM_SEL_MAX];    // keys in the bin the rank was found in
    __shared__ int rep[TTM_SEL_MAX];           // first rank with the same prefix (its histogram row)
    __shared__ int verdict;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned int nb = gridDim.x;
    constexpr int REG = TTM_SEL_MAX * 256;
    unsigned long long key[TTM_SELC_ROWS];
    bool have[TTM_SELC_ROWS];
#pragma unroll
    for (int r = 0; r < TTM_SELC_ROWS; ++r) {
        const int64_t n = ((int64_t)r * nb + blockIdx.x) * 256 + tid;
        have[r] = n < N;
        key[r] = have[r] ? f64_key(col[n]) : 0ull;
    }
    if (tid < nr) { pf[tid] = 0ull; rk[tid] = ranks[tid]; rep[tid] = 0; held[tid] = N; }
    unsigned int* ncand = bar + 16;
    bool alive = give_up == 0;                 // (give_up: tests - every workgroup behaves as if its partners had not arrived)
    int gather_shift = -1;                     // >= 0: the passes ended early; bits >= gather_shift of every rank are known
    unsigned int arrivals = 0;
    for (int pass = 0; pass < 8 && alive; ++pass) {
        const int shift = 56 - 8 * pass;
        for (int i = tid; i < nr * 256; i += 256) lh[i] = 0u;
        __syncthreads();
        const unsigned long long himask = (shift == 56) ? 0ull : (~0ull << (shift + 8));
#pragma unroll
        for (int r = 0; r < TTM_SELC_ROWS; ++r) {
            if (!have[r]) continue;
            const unsigned int bin = (unsigned int)((key[r] >> shift) & 255ull);
            for (int j = 0; j < nr; ++j)
                if (rep[j] == j && ((key[r] ^ pf[j]) & himask) == 0ull) atomicAdd(&lh[j * 256 + bin], 1u);
        }
        __syncthreads();
        unsigned int* mine = ghist + (pass % 3) * REG;
        unsigned int* next = ghist + ((pass + 1) % 3) * REG;
        for (int i = tid; i < nr * 256; i += 256)
            if (lh[i]) __hip_atomic_fetch_add(mine + i, lh[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = blockIdx.x * 256 + tid; i < nr * 256; i += (int)nb * 256) coherent_store_u32(next + i, 0u);
        arrivals += nb;
        alive = grid_barrier(bar, arrivals, &verdict);
        if (!alive) break;
        for (int j = wv; j < nr; j += 4) {
            const unsigned int* h = mine + rep[j] * 256;
            wave_pick([&](int b) { return coherent_load_u32(h + b); }, shift, pf + j, rk + j, held + j);
        }
        __syncthreads();
        if (tid < nr) {
            int first = tid;
            for (int i = tid - 1; i >= 0; --i)
                if (pf[i] == pf[tid]) first = i;
            rep[tid] = first;
        }
        __syncthreads();
        long long most = 0;
        for (int j = 0; j < nr; ++j) most = held[j] > most ? held[j] : most;
        if (shift > 0 && most <= TTM_SELC_CAND) { gather_shift = shift; break; }
    }
    if (alive && gather_shift >= 0) {
        // the keys of the bins the ranks were found in, list by list (a list per distinct prefix), then workgroup 0 ranks them
        const unsigned long long known = ~0ull << gather_shift;
#pragma unroll
        for (int r = 0; r < TTM_SELC_ROWS; ++r) {
            if (!have[r]) continue;
            for (int j = 0; j < nr; ++j)
                if (rep[j] == j && ((key[r] ^ pf[j]) & known) == 0ull) {
                    const unsigned int slot = __hip_atomic_fetch_add(ncand + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (slot < TTM_SELC_CAND) __hip_atomic_store(cand + j * TTM_SELC_CAND + slot, key[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
        }
        arrivals += nb;
        alive = grid_barrier(bar, arrivals, &verdict, blockIdx.x == 0);
        if (alive && blockIdx.x == 0) {
            for (int j = wv; j < nr; j += 4) {
                const int list = rep[j];
                const int nc = (int)held[list];
                const unsigned long long k = lane < nc ? __hip_atomic_load(cand + list * TTM_SELC_CAND + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
                int pos = 0;
                for (int l = 0; l < nc; ++l) {
                    const unsigned long long kl = __shfl(k, l, 64);
                    pos += (kl < k || (kl == k && l < lane)) ? 1 : 0;
                }
                if (lane < nc && pos == (int)rk[j]) out[j] = key_f64(k);
            }
        }
    } else if (alive && blockIdx.x == 0) {
        if (tid < nr) out[tid] = key_f64(pf[tid]);
    }
    if (!alive && blockIdx.x == 0) select_solo(col, N, ranks, nr, lh, pf, rk, held, out);
}

// ---------------------------------------------------------------------------
// K2/K3: forward map (+ fused log-determinant / squared norm)
// ---------------------------------------------------------------------------

template <int MONO, bool WANT_LD, int NS>
__global__ __launch_bounds__(256) void k_forward(DevProg P, int k0, int k1, const double* __restrict__ coef,
                                                 const double* __restrict__ fold,
                                                 const double* __restrict__ X, int64_t ldx, int64_t N,
                                                 double* __restrict__ Z, int64_t ldz,
                                                 double* __restrict__ logdet, const double* __restrict__ sigma,
                                                 double* __restrict__ sumsq) {
    typedef typename real_of<NS>::type R;
    double* slots;
    CacheStore<R> cst;
    const Prog g = make_prog_lds(P, cst, slots);
    const int bd = blockDim.x;
    LdsSlotsN<R> w{slots + threadIdx.x, bd};
    const bool want_val = (Z != nullptr) || (sumsq != nullptr);
    cint_p off = (cint_p)P.off;
    cint_p itab = (cint_p)P.itab;
    // a thread carries NS samples (rows tile + e*blockDim + tid) through one pass of the table interpreter
    for (int64_t tile = (int64_t)blockIdx.x * NS * bd; tile < N; tile += (int64_t)gridDim.x * NS * bd) {
        XSoAN<NS> xa;
        xa.X = X; xa.ld = ldx;
        bool act[NS];
#pragma unroll
        for (int e = 0; e < NS; ++e) {
            const int64_t n = tile + (int64_t)e * bd + threadIdx.x;
            act[e] = n < N;
            xa.n[e] = act[e] ? n : N - 1;
        }
        VarCache<XSoAN<NS>, R> x(xa, cst);
        R ld(0.0), ss(0.0);
        // the component's own column is fetched one component ahead, so its HBM latency overlaps
        // the arithmetic of the current component
        R xk_next = xa(((cint_p)P.fdesc)[k0 * TTM_FDESC_LEN + TTM_FD_KC]);
        for (int k = k0; k < k1; ++k) {
            cint_p fd = (cint_p)P.fdesc + k * TTM_FDESC_LEN;
            x.put(fd[TTM_FD_KC], xk_next);
            if (k + 1 < k1) xk_next = xa(fd[TTM_FDESC_LEN + TTM_FD_KC]);
            R S, dS;
            if (!fd[TTM_FD_COMPLEX]) {
                // all terms univariate: flat streams, records prefetched by scalar loads
                const FastComp f = make_fast(fd, (cint_p)P.fints, (cdbl_p)fold, 0);
                TaggedFetch<XSoAN<NS>, R> xf{x};
                const R xk = x.get(f.kc);
                if (WANT_LD) sample_forward_fast<MONO, -1, true>(f, g, xk, xf, want_val, S, dS);
                else sample_forward_fast<MONO, -1, false>(f, g, xk, xf, true, S, dS);
            } else {
                const Comp c = comp_at(P, k, 0, coef, fold);
                if (WANT_LD) sample_forward<MONO, true>(c, g, x, w, want_val, S, dS);
                else sample_forward<MONO, false>(c, g, x, w, true, S, dS);
            }
            if (WANT_LD) ld += fast_log(sigma ? fast_div(dS, ((cdbl_p)sigma)[k - k0]) : dS);
            if (Z) {
#pragma unroll
                for (int e = 0; e < NS; ++e)
                    if (act[e]) Z[(int64_t)(k - k0) * ldz + xa.n[e]] = elem(S, e);
            }
            ss = vfma(S, S, ss);
        }
#pragma unroll
        for (int e = 0; e < NS; ++e) {
            if (act[e]) {
                if (WANT_LD) logdet[xa.n[e]] = elem(ld, e);
                if (sumsq) sumsq[xa.n[e]] = elem(ss, e);
            }
        }
    }
}

// Forward map of a range of components that all take the fast path (univariate terms only): no generic
// interpreter in the kernel, the column cache is statically planned (PlanCache), the polynomial family is
// a template parameter (FAM = -1: run-time), special terms come as unified branch-free records.
template <int MONO, bool WANT_LD, int NS, int FAM>
__global__ __launch_bounds__(256) void k_forward_plan(DevProg P, int k0, int k1, const double* __restrict__ fold,
                                                      const double* __restrict__ X, int64_t ldx, int64_t N,
                                                      double* __restrict__ Z, int64_t ldz,
                                                      double* __restrict__ logdet, const double* __restrict__ sigma,
                                                      double* __restrict__ sumsq) {
    typedef typename real_of<NS>::type R;
    double* unused;
    CacheStore<R> cst;
    const Prog g = make_prog_lds(P, cst, unused);
    const int bd = blockDim.x;
    const bool want_val = (Z != nullptr) || (sumsq != nullptr);
    cint_p fd0 = (cint_p)P.fdesc + k0 * TTM_FDESC_LEN;
    for (int64_t tile = (int64_t)blockIdx.x * NS * bd; tile < N; tile += (int64_t)gridDim.x * NS * bd) {
        XSoAN<NS> xa;
        xa.X = X; xa.ld = ldx;
        bool act[NS];
#pragma unroll
        for (int e = 0; e < NS; ++e) {
            const int64_t n = tile + (int64_t)e * bd + threadIdx.x;
            act[e] = n < N;
            xa.n[e] = act[e] ? n : N - 1;
        }
        PlanCache<XSoAN<NS>, R> x(xa, cst);
        if (k0 > 0) x.warm((cint_p)P.fints + fd0[TTM_FD_PLAN_OFF]);
        R ld(0.0), ss(0.0);
        // the component's own column is fetched one component ahead, so its HBM latency overlaps
        // the arithmetic of the current component
        R xk_next = xa(fd0[TTM_FD_KC]);
        for (int k = k0; k < k1; ++k) {
            cint_p fd = (cint_p)P.fdesc + k * TTM_FDESC_LEN;
            const R xk = xk_next;
            if (k + 1 < k1) xk_next = xa(fd[TTM_FDESC_LEN + TTM_FD_KC]);
            const FastComp f = make_fast(fd, (cint_p)P.fints, (cdbl_p)fold, 0);
            R S, dS;
            if (WANT_LD) sample_forward_fast<MONO, FAM, true>(f, g, xk, x, want_val, S, dS);
            else sample_forward_fast<MONO, FAM, false>(f, g, xk, x, true, S, dS);
            x.put(fd[TTM_FD_KC_SLOT], xk);
            if (WANT_LD) ld += fast_log(sigma ? fast_div(dS, ((cdbl_p)sigma)[k - k0]) : dS);
            if (Z) {
#pragma unroll
                for (int e = 0; e < NS; ++e)
                    if (act[e]) Z[(int64_t)(k - k0) * ldz + xa.n[e]] = elem(S, e);
            }
            ss = vfma(S, S, ss);
        }
#pragma unroll
        for (int e = 0; e < NS; ++e) {
            if (act[e]) {
                if (WANT_LD) logdet[xa.n[e]] = elem(ld, e);
                if (sumsq) sumsq[xa.n[e]] = elem(ss, e);
            }
        }
    }
}


// ---------------------------------------------------------------------------
// U-form (csrc/ttm_uform.h): builder and forward kernel
// ---------------------------------------------------------------------------

struct UTabs {               // by-value kernel argument
    const int* ucomp;
    const int* ugrp;
    const double* umono;
    const double* ugeo;
};

// one workgroup per component: monomial coefficients of the groups, spline of the summed special terms, fit check
// (1024 threads per component: the node, fit and verification phases each hand out nI x 12..13 independent evaluations -
// ~450 at C5 - and a workgroup of four waves walked them in two dependent rounds: 15.8 us per launch, 9 with sixteen)
// coef != nullptr: the workgroup first folds the component's coefficients itself (what k_fold does: one thread per slot, fixed
// summation order - the same bits) - a new coefficient vector then costs ONE launch for fold + U-form + push records instead
// of three (k_fold 6.3 us, k_uform 10 us, k_band_records 4.9 us and the gaps between them: profiles/r03_v2_*).
// p_lag > 0: the component's share of the push records (uform_scatter_push_records) behind its hot record.
// coef_src != nullptr: `coef_src` is page-locked HOST memory the device can read (the packed coefficient vector as the host
// wrote it): the workgroup first copies its component's slice into the device vector `coef` - the upload of a new vector
// costs no copy in the stream (an H2D copy between two kernels sits between two engine switches: 3.7 us + 10.8 us of gap in
// front of it, tools/trace_uncached.sh).  err_host != nullptr: the component's two fit errors also go to page-locked host
// memory by system-scope stores (the host reads them behind an event instead of behind a D2H copy).
// what one workgroup does for component k of a new coefficient vector (k_uform; the uform half of k_setup)
__device__ __forceinline__ void uform_body(const DevProg& P, const UTabs& T, int k, const double* __restrict__ coef_src, double* coef,
                                           double* __restrict__ fold, double* __restrict__ U, int64_t err_off, int64_t h_off, int h_cls,
                                           int h_ng, int64_t p_off, int p_lag, int p_stride, double* err_host) {
    __shared__ double ybuf[TTM_U_NI_MAX * TTM_CHEB_N];
    __shared__ double red[2][16];
    const int tid = threadIdx.x, bd = blockDim.x;
    const int D1 = P.D + 1;
    const int* uc = T.ucomp + k * TTM_UC_LEN;
    const int* fd = P.fdesc + k * TTM_FDESC_LEN;
    if (coef) {
        const int* off = P.off;
        if (coef_src) {
            for (int i = off[2 * D1 + k] + tid; i < off[2 * D1 + k + 1]; i += bd) coef[i] = coef_src[i];
            __syncthreads();
        }
        fold_coeffs(P.itab + off[k], P.ftab + off[4 * D1 + k], P.dpar + off[D1 + k], coef + off[2 * D1 + k], fold + off[3 * D1 + k], tid, bd);
        __syncthreads();
        fold_st8(fd, P.fints, fold + off[3 * D1 + k], tid, bd);
        __syncthreads();
    }
    const double* foldk = fold + P.off[3 * D1 + k];
    const double* geo = T.ugeo + 2 * k;
    uform_build_groups(uc, T.ugrp, fd, T.umono, geo, foldk, U, tid, bd);
    if (h_cls > 0) {
        __syncthreads();
        uform_build_hot(uc, T.ugrp, U, h_off, h_cls, h_ng, k, foldk[0], tid, bd);
    }
    double ev = 0.0, ed = 0.0;
    if (uc[TTM_UC_NI] > 0) {
        uform_spline_nodes(uc, fd, geo, foldk, ybuf, tid, bd);
        __syncthreads();
        uform_spline_fit(uc, fd, geo, foldk, ybuf, U, tid, bd);
        __syncthreads();
        uform_spline_verify(uc, fd, geo, foldk, U, tid, bd, ev, ed);
    }
    if (h_cls > 0 && p_lag > 0) {
        // (behind the spline fit: it writes whole columns, the padding slot that takes the column's offset included)
        __syncthreads();
        uform_scatter_push_records(T.ucomp, T.ugrp, U, h_off, h_cls, h_ng, p_off, p_lag, p_stride, P.D, k, tid, bd);
    }
    if (ev != ev) ev = INFINITY;
    if (ed != ed) ed = INFINITY;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ev = fmax(ev, __shfl_down(ev, off, 64));
        ed = fmax(ed, __shfl_down(ed, off, 64));
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = ev; red[1][tid >> 6] = ed; }
    __syncthreads();
    if (tid == 0) {
        const int nw = bd >> 6;
        for (int w = 1; w < nw; ++w) { ev = fmax(ev, red[0][w]); ed = fmax(ed, red[1][w]); }
        U[err_off + 2 * k] = ev;
        U[err_off + 2 * k + 1] = ed;
        if (err_host) {
            __hip_atomic_store(err_host + 2 * k, ev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(err_host + 2 * k + 1, ed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ __launch_bounds__(1024) void k_uform(DevProg P, UTabs T, const double* __restrict__ coef_src, double* coef,
                                                double* __restrict__ fold, double* __restrict__ U, int64_t err_off, int64_t h_off,
                                                int h_cls, int h_ng, int64_t p_off, int p_lag, int p_stride, double* err_host) {
    uform_body(P, T, (int)blockIdx.x, coef_src, coef, fold, U, err_off, h_off, h_cls, h_ng, p_off, p_lag, p_stride, err_host);
}

// NS rows of one thread inside the current tile (32-bit row numbers: the column base stays a scalar pointer and
// every access is `global_load_dwordx2 v, v_off, s[base]`)
template <int NS>
struct XRowsN {
    typedef typename real_of<NS>::type R;
    const double* X;
    int64_t ld;
    unsigned int n[NS];
    __device__ __forceinline__ R operator()(int var) const {
        const double* col = X + (int64_t)var * ld;
        R r;
#pragma unroll
        for (int e = 0; e < NS; ++e) set_elem(r, e, col[n[e]]);
        return r;
    }
};

typedef double D2 __attribute__((ext_vector_type(2)));      // 16-byte staging unit of the spline tables

// rows of one thread as BYTE offsets into a column (32-bit): every access is `global_load_dwordx2 v, v_off, s[base]`
template <int NS>
struct XOffN {
    typedef typename real_of<NS>::type R;
    const char* X;
    int64_t ldb;                 // column stride in bytes
    unsigned int off[NS];
    __device__ __forceinline__ R operator()(int var) const {
        const char* col = X + (int64_t)var * ldb;
        R r;
#pragma unroll
        for (int e = 0; e < NS; ++e) set_elem(r, e, *(const double*)(col + off[e]));
        return r;
    }
};

// Forward map in U-form.  The workgroup walks a flat sequence of steps (tile, component).  While step s is
// evaluated, the x_k column and the spline table of step s + 1 are in flight to registers; the table is written
// to the other LDS buffer at the end of the step: one barrier per step, no pipeline drain at tile boundaries.
// DB / DA: Horner degrees of the Hermite-function / plain part of every nonmonotone group (-1: per group).
// LDS: [table buffer 0 | table buffer 1 | column cache (2 x ways x NS x blockDim)]
template <bool WANT_LD, int NS, int DB, int DA>
__global__ __launch_bounds__(256) void k_forward_u(const int* __restrict__ ucomp_, const int* __restrict__ ugrp_,
                                                   const double* __restrict__ U_, int D, int k0, int k1,
                                                   const double* __restrict__ X, int64_t ldx, int64_t N,
                                                   double* __restrict__ Z, int64_t ldz, double* __restrict__ logdet,
                                                   const double* __restrict__ sigma, double* __restrict__ sumsq, int tab_cap) {
    typedef typename real_of<NS>::type R;
    cint_p ucomp = (cint_p)ucomp_;
    cint_p ugrp = (cint_p)ugrp_;
    cdbl_p U = (cdbl_p)U_;
    const int bd = blockDim.x, tid = threadIdx.x;
    const int ncomp = k1 - k0;
    const int64_t rpt = (int64_t)NS * bd;
    const int64_t ntiles = (N + rpt - 1) / rpt;
    if ((int64_t)blockIdx.x >= ntiles) return;
    const int64_t S = ((ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x) * ncomp;
    double* tabbuf = g_smem;
    CacheStore<R> cst;
    cst.base = g_smem + 2 * (size_t)tab_cap + tid;
    cst.stride = bd;
    const bool want_val = (Z != nullptr) || (sumsq != nullptr);

    XOffN<NS> cx;                // rows of the tile being evaluated
    cx.X = (const char*)X; cx.ldb = ldx * 8;
    bool act[NS];
    int64_t ctile = blockIdx.x;
#pragma unroll
    for (int e = 0; e < NS; ++e) {
        const int64_t n = ctile * rpt + (int64_t)e * bd + tid;
        act[e] = n < N;
        cx.off[e] = (unsigned int)(act[e] ? n : N - 1) * 8u;
    }
    // first column, first table
    R xk_next = cx(ucomp[k0 * TTM_UC_LEN + TTM_UC_KC]);
    {
        const int n16 = (TTM_U_TSTRIDE / 2) * ucomp[k0 * TTM_UC_LEN + TTM_UC_NI];
        const D2* src = (const D2*)(U_ + ucomp[k0 * TTM_UC_LEN + TTM_UC_TAB_OFF]);
        D2* dst = (D2*)tabbuf;
        for (int i = tid; i < n16; i += bd) dst[i] = src[i];
    }
    __syncthreads();

    int k = k0;
    R ld(0.0), ss(0.0);
    for (int64_t s = 0; s < S; ++s) {
        cint_p uc = ucomp + k * TTM_UC_LEN;
        const R xk = xk_next;
        // next step: component, rows, its column and its spline table
        const bool wrap = (k + 1 == k1);
        const int knext = wrap ? k0 : k + 1;
        const bool has_next = s + 1 < S;
        XOffN<NS> nx = cx;
        bool nact[NS];
#pragma unroll
        for (int e = 0; e < NS; ++e) nact[e] = act[e];
        if (wrap) {
            const int64_t t2 = ctile + gridDim.x;
#pragma unroll
            for (int e = 0; e < NS; ++e) {
                const int64_t n = t2 * rpt + (int64_t)e * bd + tid;
                nact[e] = n < N;
                nx.off[e] = (unsigned int)(nact[e] ? n : N - 1) * 8u;
            }
        }
        cint_p ucn = ucomp + knext * TTM_UC_LEN;
        const int n16 = has_next ? (TTM_U_TSTRIDE / 2) * ucn[TTM_UC_NI] : 0;
        // (every lane loads, with a clamped index, inside uniform branches: the values stay in registers and the
        // loads stay asynchronous; a per-lane predicate here makes the compiler park them in scratch)
        const D2* tsrc = (const D2*)(U_ + ucn[TTM_UC_TAB_OFF]);
        D2 stg0 = {0.0, 0.0}, stg1 = {0.0, 0.0}, stg2 = {0.0, 0.0}, stg3 = {0.0, 0.0};
        if (n16 > 0) stg0 = tsrc[min(tid, n16 - 1)];
        if (n16 > bd) stg1 = tsrc[min(tid + bd, n16 - 1)];
        if (n16 > 2 * bd) stg2 = tsrc[min(tid + 2 * bd, n16 - 1)];
        if (n16 > 3 * bd) stg3 = tsrc[min(tid + 3 * bd, n16 - 1)];
        if (has_next) xk_next = nx(ucn[TTM_UC_KC]);

        PlanCache<XOffN<NS>, R> x(cx, cst);
        if (k == k0) {
            ld = R(0.0); ss = R(0.0);
            if (k0 > 0) x.warm(ucomp + TTM_UC_STATE(D, k0));
        }
        const double* tab = tabbuf + (size_t)(s & 1) * tab_cap;
        R Sv, dS;
        u_component<DB, DA, WANT_LD>(uc, ugrp, U, tab, xk, x, WANT_LD ? want_val : true, Sv, dS);
        if (WANT_LD) ld += fast_log(sigma ? fast_div(dS, ((cdbl_p)sigma)[k - k0]) : dS);
        // staged table -> the other buffer (its last readers passed the previous barrier); before the stores of
        // this step so that waiting for the table loads does not wait for the stores
        D2* tdst = (D2*)(tabbuf + (size_t)((s + 1) & 1) * tab_cap);
        if (tid < n16) tdst[tid] = stg0;
        if (tid + bd < n16) tdst[tid + bd] = stg1;
        if (tid + 2 * bd < n16) tdst[tid + 2 * bd] = stg2;
        if (tid + 3 * bd < n16) tdst[tid + 3 * bd] = stg3;
        if (Z) {
            char* zc = (char*)(Z + (int64_t)(k - k0) * ldz);
#pragma unroll
            for (int e = 0; e < NS; ++e)
                if (act[e]) *(double*)(zc + cx.off[e]) = elem(Sv, e);
        }
        ss = vfma(Sv, Sv, ss);
        if (wrap) {
#pragma unroll
            for (int e = 0; e < NS; ++e) {
                if (act[e]) {
                    if (WANT_LD) *(double*)((char*)logdet + cx.off[e]) = elem(ld, e);
                    if (sumsq) *(double*)((char*)sumsq + cx.off[e]) = elem(ss, e);
                }
            }
            ctile += gridDim.x;
        }
        __syncthreads();
        k = knext;
        cx = nx;
#pragma unroll
        for (int e = 0; e < NS; ++e) act[e] = nact[e];
    }
}

// ---------------------------------------------------------------------------
// Forward map in U-form with dedicated loader waves (the kernel the large-ensemble path runs).
//
// Workgroup = 4 evaluating waves (256 threads, two adjacent samples each: a 512-row tile) + wave 4, which streams
// the x_k columns, + wave 5, which streams the spline tables - both by LDS-DMA (global_load_lds_dwordx4: no
// registers, no LDS instructions), each into its own ring of LDS slots, xlead / tlead steps ahead of the
// evaluation.  The evaluating waves issue NO vector loads: they read their column pair from the ring with one
// ds_read_b128, so nothing they do waits on HBM latency, and the compiler's conservative vmcnt placement has
// nothing to serialise.  One s_barrier per step hands a slot over: a loader arrives at barrier A(s) only after
// its counted `s_waitcnt vmcnt(n)` says the data of step s has landed, and, having passed A(s), may overwrite the
// slot of step s-1 (every evaluating wave finished it before arriving).  Loaders keep issuing (harmless
// duplicate) DMAs past the last step so that the counts behind the immediates stay valid.
// Requires 16-byte aligned X / Z columns (even leading dimensions) and ldx >= N rounded up to even.
// LDS (doubles): [x ring: (LEAD+1) x 512 | table ring: 3 x tab_slot | column cache: 2 x ways x 2 x 256]
// ---------------------------------------------------------------------------
#ifndef TTM_UL_CW
#define TTM_UL_CW 4           // evaluating waves per workgroup
#endif
#ifndef TTM_FWD_ETAB          // tuning knob: which forward hot kernels take exp(-x^2/4) from the 2^(j/32) table
#define TTM_FWD_ETAB(NS) false
#endif
#define TTM_UL_THREADS ((TTM_UL_CW + 2) * 64)
#define TTM_UL_CT (TTM_UL_CW * 64)             // evaluating threads
#define TTM_UL_ROWS (TTM_UL_CW * 128)          // rows per tile (evaluating waves x 64 lanes x 2 samples)
// Evaluating waves per workgroup of the hot FORWARD kernel: two for the plain map (four workgroups per CU instead of
// two: the waves of a workgroup move in lock step from barrier to barrier and want the same unit at the same time -
// more, smaller workgroups interleave better: 0.172 -> 0.162 ms at C5), four with the fused log-determinant (more
// arithmetic per step; the per-step table is then shared by twice the rows).  The inverse stays at TTM_UL_CW (its
// 12 KB table per step wants the larger tile).
#ifndef TTM_HL_CW_PLAIN
#define TTM_HL_CW_PLAIN 2
#endif
#ifndef TTM_HL_CW_LD
#define TTM_HL_CW_LD 4
#endif
#define TTM_HL_FWD_CW(WANT_LD) ((WANT_LD) ? TTM_HL_CW_LD : TTM_HL_CW_PLAIN)
#define TTM_HL_FWD_BOUNDS(WANT_LD) __launch_bounds__((TTM_HL_FWD_CW(WANT_LD) + 2) * 64)
// xlead / tlead (kernel arguments): how many steps ahead of the evaluation the x / table loaders run; the rings have
// xlead + 1 and tlead + 1 slots.  xlead <= 4 and tlead <= 2 (the vmcnt immediates).

#define TTM_RAW_BARRIER()                       \
    do {                                        \
        asm volatile("" ::: "memory");          \
        __builtin_amdgcn_s_barrier();           \
        asm volatile("" ::: "memory");          \
    } while (0)

__device__ __forceinline__ void ul_wait_vmcnt(int n) {      // all but the n youngest vector-memory operations are done
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
        case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
        case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
        case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
        case 27: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break;
        case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
        case 29: asm volatile("s_waitcnt vmcnt(29)" ::: "memory"); break;
        case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;      // (more than 30 younger operations: drain)
    }
}

__device__ __forceinline__ void ul_dma16(const void* g, double* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Column loader wave: streams, for every step (tile, k) of the workgroup's flat step sequence, the 16-byte row pairs
// of column col_of(k) of the tile into ring slot step % XSLOTS, xlead steps ahead of the evaluation, and meets the
// evaluating waves at one barrier per step (A(0) before the first step, A(s + 1) after step s): it arrives only
// after its counted vmcnt wait says the data of the next step has landed, and having passed A(s) it may overwrite
// the slot of step s - 1.  Past the last step it keeps issuing (harmless duplicates of the last column) so that
// the number of younger operations behind the vmcnt immediate stays what it is in the steady state.
template <int ROWS, class ColOf>
__device__ __forceinline__ void ul_column_loader(ColOf col_of, int k0, int k1, int64_t S, int64_t N, double* ring,
                                                 int XSLOTS, int xlead, int lane) {
    const int64_t last_pair = ((N + 1) & ~(int64_t)1) - 2;               // first row of the last readable pair
    int64_t ptile = blockIdx.x;
    int pk = k0;
    int64_t pstep = 0;                                                   // step the cursor (ptile, pk) stands for
    int xs = 0;                                                          // ring slot of the next issue
    auto issue = [&](int64_t step) {
        while (pstep < step && pstep < S - 1) {                          // advance the cursor to min(step, S - 1)
            ++pstep;
            if (++pk == k1) { pk = k0; ptile += gridDim.x; }
        }
        const double* col = col_of(pk);
        double* slot = ring + (size_t)xs * ROWS;
        xs = (xs + 1 == XSLOTS) ? 0 : xs + 1;
#pragma unroll
        for (int c = 0; c < ROWS / 128; ++c) {
            int64_t pair = ptile * ROWS + c * 128 + lane * 2;
            pair = pair < last_pair ? pair : last_pair;
            ul_dma16(col + pair, slot + c * 128);
        }
    };
    for (int j = 0; j < xlead; ++j) issue(j);
    ul_wait_vmcnt((xlead - 1) * (ROWS / 128));
    TTM_RAW_BARRIER();                                                   // A(0)
    for (int64_t s = 0; s < S; ++s) {
        issue(s + xlead);
        ul_wait_vmcnt((xlead - 1) * (ROWS / 128));                       // the column of step s + 1 has landed
        TTM_RAW_BARRIER();                                               // A(s + 1)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // drain before the LDS is released
}

// Table loader wave: issue_tab(k, slot) DMAs the per-component table of component k into `slot` and returns the
// number of operations it issued (<= 15); same barrier protocol, tlead (1 or 2) steps ahead.
template <class IssueTab>
__device__ __forceinline__ void ul_table_loader(IssueTab issue_tab, int k0, int k1, int64_t S, double* tabs, int tab_slot,
                                                int TSLOTS, int tlead) {
    int pk = k0;
    int64_t pstep = 0;
    int ts = 0;                                                          // table slot of the next issue
    auto issue = [&](int64_t step) -> int {
        while (pstep < step && pstep < S - 1) {
            ++pstep;
            if (++pk == k1) pk = k0;
        }
        double* slot = tabs + (size_t)ts * tab_slot;
        ts = (ts + 1 == TSLOTS) ? 0 : ts + 1;
        return issue_tab(pk, slot);
    };
    issue(0);
    const int n1 = tlead > 1 ? issue(1) : 0;
    ul_wait_vmcnt(n1);
    TTM_RAW_BARRIER();                                                   // A(0)
    for (int64_t s = 0; s < S; ++s) {
        const int n = issue(s + tlead);
        ul_wait_vmcnt(tlead > 1 ? n : 0);                                // the table of step s + 1 has landed
        TTM_RAW_BARRIER();                                               // A(s + 1)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// DMA of `bytes` (multiple of 16) from src to an LDS slot, 1 KB per wave-instruction; returns the instruction count
__device__ __forceinline__ int ul_dma_block(const char* src, int bytes, double* slot, int lane) {
    const int nch = (bytes + 1023) >> 10;
    for (int c = 0; c < nch; ++c) {
        const int off = c * 1024 + lane * 16;
        if (off < bytes) ul_dma16(src + off, slot + c * 128);
    }
    return nch;
}

__device__ __forceinline__ void ul_dma4(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}

template <bool WANT_LD, int DB, int DA>
__global__ __launch_bounds__(TTM_UL_THREADS) void k_forward_ul(const int* __restrict__ ucomp_, const int* __restrict__ ugrp_,
                                                    const double* __restrict__ U_, int D, int k0, int k1,
                                                    const double* __restrict__ X, int64_t ldx, int64_t N,
                                                    double* __restrict__ Z, int64_t ldz, double* __restrict__ logdet,
                                                    const double* __restrict__ sigma, double* __restrict__ sumsq,
                                                    int tab_slot, int xlead, int tlead) {
    typedef VecD<2> R;
    const int XSLOTS = xlead + 1, TSLOTS = tlead + 1;
    cint_p ucomp = (cint_p)ucomp_;
    cint_p ugrp = (cint_p)ugrp_;
    cdbl_p U = (cdbl_p)U_;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int ncomp = k1 - k0;
    const int64_t ntiles = (N + TTM_UL_ROWS - 1) / TTM_UL_ROWS;
    if ((int64_t)blockIdx.x >= ntiles) return;
    const int64_t my_tiles = (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const int64_t S = my_tiles * ncomp;
    double* ring = g_smem;
    double* tabs = g_smem + (size_t)XSLOTS * TTM_UL_ROWS;
    double* cache = tabs + (size_t)TSLOTS * tab_slot;

    if (wv == TTM_UL_CW) {
        ul_column_loader<TTM_UL_ROWS>([&](int kk) { return X + (int64_t)ucomp[kk * TTM_UC_LEN + TTM_UC_KC] * ldx; }, k0, k1, S, N,
                                      ring, XSLOTS, xlead, lane);
        return;
    }
    if (wv == TTM_UL_CW + 1) {
        ul_table_loader([&](int kk, double* slot) {
            const int bytes = ucomp[kk * TTM_UC_LEN + TTM_UC_NI] * (TTM_U_TSTRIDE * 8);
            return ul_dma_block((const char*)(U_ + ucomp[kk * TTM_UC_LEN + TTM_UC_TAB_OFF]), bytes, slot, lane);
        }, k0, k1, S, tabs, tab_slot, TSLOTS, tlead);
        return;
    }

    // ---- evaluating waves -------------------------------------------------------------------------------------------
    CacheStore<R> cst;
    cst.base = cache + tid;
    cst.stride = TTM_UL_CT;
    const bool want_val = (Z != nullptr) || (sumsq != nullptr);
    XOffN<2> cx;
    cx.X = (const char*)X; cx.ldb = ldx * 8;
    bool act0 = false, act1 = false;
    int64_t ctile = blockIdx.x;
    int k = k0;
    R ld(0.0), ss(0.0);
    int xs = 0, ts = 0;
    TTM_RAW_BARRIER();                                                   // A(0)
    for (int64_t s = 0; s < S; ++s) {
        cint_p uc = ucomp + k * TTM_UC_LEN;
        if (k == k0) {
            const int64_t n = ctile * TTM_UL_ROWS + 2 * tid;
            act0 = n < N; act1 = n + 1 < N;
            cx.off[0] = (unsigned int)(act0 ? n : N - 1) * 8u;
            cx.off[1] = (unsigned int)(act1 ? n + 1 : N - 1) * 8u;
            ld = R(0.0); ss = R(0.0);
        }
        PlanCache<XOffN<2>, R> x(cx, cst);
        if (k == k0 && k0 > 0) x.warm(ucomp + TTM_UC_STATE(D, k0));
        const D2 xp = *(const D2*)(ring + (size_t)xs * TTM_UL_ROWS + 2 * tid);
        R xk;
        xk.v[0] = xp.x; xk.v[1] = xp.y;
        const double* tab = tabs + (size_t)ts * tab_slot;
        xs = (xs + 1 == XSLOTS) ? 0 : xs + 1;
        ts = (ts + 1 == TSLOTS) ? 0 : ts + 1;
        R Sv, dS;
        u_component<DB, DA, WANT_LD>(uc, ugrp, U, tab, xk, x, WANT_LD ? want_val : true, Sv, dS);
        if (WANT_LD) ld += fast_log(sigma ? fast_div(dS, ((cdbl_p)sigma)[k - k0]) : dS);
        if (Z) {
            double* zc = Z + (int64_t)(k - k0) * ldz + ctile * TTM_UL_ROWS + 2 * tid;
            if (act1) { D2 o = {Sv.v[0], Sv.v[1]}; *(D2*)zc = o; }
            else if (act0) *zc = Sv.v[0];
        }
        ss = vfma(Sv, Sv, ss);
        const bool wrap = (k + 1 == k1);
        if (wrap) {
            const int64_t n = ctile * TTM_UL_ROWS + 2 * tid;
            if (WANT_LD) {
                if (act1) { D2 o = {ld.v[0], ld.v[1]}; *(D2*)(logdet + n) = o; }
                else if (act0) logdet[n] = ld.v[0];
            }
            if (sumsq) {
                if (act1) { D2 o = {ss.v[0], ss.v[1]}; *(D2*)(sumsq + n) = o; }
                else if (act0) sumsq[n] = ss.v[0];
            }
            ctile += gridDim.x;
            k = k0;
        } else {
            ++k;
        }
        TTM_RAW_BARRIER();                                               // A(s + 1)
    }
}

// Hot-record variant of k_forward_ul (include/ttm.h "H section"): the evaluating waves read ONE fixed-stride record
// per step (all scalar loads at known offsets, issued together) and run straight-line code: NG group records of
// degree class CLS, every column from the planned cache.  Same loader waves, rings and barrier protocol.
template <bool WANT_LD, int NG, int CLS, int NS>
__global__ TTM_HL_FWD_BOUNDS(WANT_LD) void k_forward_hl(const int* __restrict__ ucomp_, const double* __restrict__ U_, int64_t h_off,
                                                    int D, int k0, int k1,
                                                    const double* __restrict__ X, int64_t ldx, int64_t N,
                                                    double* __restrict__ Z, int64_t ldz, double* __restrict__ logdet,
                                                    const double* __restrict__ sigma, double* __restrict__ sumsq,
                                                    int tab_slot, int xlead, int tlead, int ways) {
    typedef VecD<NS> R;
    constexpr int CW = TTM_HL_FWD_CW(WANT_LD), CT = CW * 64;             // evaluating waves / threads
    constexpr int ROWS = CT * NS;         // rows per tile: NS / 2 blocks of 2 x CT rows, thread t owns the
    constexpr int NP = NS / 2;                   // adjacent pair (2t, 2t+1) of every block
    const int XSLOTS = xlead + 1, TSLOTS = tlead + 1;
    constexpr int DB = CLS == 1 ? 3 : (CLS == 2 ? 5 : 7), DA = CLS == 1 ? 1 : (CLS == 2 ? 5 : 7), GS = CLS == 1 ? 8 : (CLS == 2 ? 16 : 24);
    constexpr int HS = TTM_H_HDR + NG * GS;
    cdbl_p H = (cdbl_p)(U_ + h_off);
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int ncomp = k1 - k0;
    const int64_t ntiles = (N + ROWS - 1) / ROWS;
    if ((int64_t)blockIdx.x >= ntiles) return;
    const int64_t my_tiles = (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const int64_t S = my_tiles * ncomp;
    double* ring = g_smem;
    double* tabs = ring + (size_t)XSLOTS * ROWS;
    double* cache = tabs + (size_t)TSLOTS * tab_slot;

    if (wv == CW) {
        ul_column_loader<ROWS>([&](int kk) { return X + (int64_t)((cint_p)(H + (int64_t)kk * HS))[3] * ldx; }, k0, k1, S, N, ring,
                               XSLOTS, xlead, lane);
        return;
    }
    if (wv == CW + 1) {
        ul_table_loader([&](int kk, double* slot) {
            cint_p ri = (cint_p)(H + (int64_t)kk * HS);
            return ul_dma_block((const char*)(U_ + ri[12]), ri[2] * (TTM_U_TSTRIDE * 8), slot, lane);
        }, k0, k1, S, tabs, tab_slot, TSLOTS, tlead);
        return;
    }

    // ---- evaluating waves -------------------------------------------------------------------------------------------
    typedef CacheStore<R, true> Store;                                   // (x_j, exp(-x_j^2/4)) adjacent: 16-byte accesses
    Store cst;
    cst.base = cache + 2 * tid;
    cst.stride = CT;
    if (TTM_FWD_ETAB(NS)) {                                              // 2^(j/32) table behind the column cache
        double* etab = cache + (size_t)2 * ways * NS * CT;
        if (tid < TTM_EXPQ_TABLE_LEN) etab[tid] = g_expq_table[tid];
        cst.etab = etab;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (barrier A(0) follows)
    }
    const bool want_val = (Z != nullptr) || (sumsq != nullptr);
    bool act0[NP], act1[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) { act0[q] = false; act1[q] = false; }
    int64_t ctile = blockIdx.x;
    int k = k0;
    R ld(0.0), ss(0.0);
    int xs = 0, ts = 0;
    bool full = false;                                                   // (uniform) every row of the tile exists
    // sum_k log(dS_k / sigma_k) = sum_k log dS_k - sum_k log sigma_k: the second sum is one number per launch, taken once
    // here instead of a division per component evaluation (8 instructions of ~110)
    double lsig = 0.0;
    if (WANT_LD && sigma)
        for (int kk = k0; kk < k1; ++kk) lsig += fast_log(((cdbl_p)sigma)[kk - k0]);
    TTM_RAW_BARRIER();                                                   // A(0)
    for (int64_t s = 0; s < S; ++s) {
        cdbl_p rec = H + (int64_t)k * HS;
        if (k == k0) {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int64_t n = ctile * ROWS + q * (2 * CT) + 2 * tid;
                act0[q] = n < N; act1[q] = n + 1 < N;
            }
            full = (ctile + 1) * ROWS <= N;
            ld = R(0.0); ss = R(0.0);
            if (k0 > 0) {                                                // (rare: sweeps that start inside the map)
                XOffN<NS> cx;
                cx.X = (const char*)X; cx.ldb = ldx * 8;
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const int64_t n = ctile * ROWS + q * (2 * CT) + 2 * tid;
                    cx.off[2 * q] = (unsigned int)(act0[q] ? n : N - 1) * 8u;
                    cx.off[2 * q + 1] = (unsigned int)(act1[q] ? n + 1 : N - 1) * 8u;
                }
                PlanCache<XOffN<NS>, R, Store> x(cx, cst);
                x.warm((cint_p)ucomp_ + TTM_UC_STATE(D, k0));
            }
        }
        R xk;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const D2 xp = *(const D2*)(ring + (size_t)xs * ROWS + q * (2 * CT) + 2 * tid);
            xk.v[2 * q] = xp.x; xk.v[2 * q + 1] = xp.y;
        }
        const double* tab = tabs + (size_t)ts * tab_slot;
        xs = (xs + 1 == XSLOTS) ? 0 : xs + 1;
        ts = (ts + 1 == TSLOTS) ? 0 : ts + 1;
        R Sv, dS;
        h_component<NG, DB, DA, GS, WANT_LD, TTM_FWD_ETAB(NS)>(rec, tab, xk, cst, WANT_LD ? want_val : true, Sv, dS);
        if (WANT_LD) ld += fast_log(dS);
        if (Z) {
            double* zt = Z + (int64_t)(k - k0) * ldz + ctile * ROWS + 2 * tid;
            if (full) {                                                  // all but the last tile: no per-lane masks
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    D2 o = {Sv.v[2 * q], Sv.v[2 * q + 1]};
                    *(D2*)(zt + q * (2 * CT)) = o;
                }
            } else {
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    double* zc = zt + q * (2 * CT);
                    if (act1[q]) { D2 o = {Sv.v[2 * q], Sv.v[2 * q + 1]}; *(D2*)zc = o; }
                    else if (act0[q]) *zc = Sv.v[2 * q];
                }
            }
        }
        ss = vfma(Sv, Sv, ss);
        if (k + 1 == k1) {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int64_t n = ctile * ROWS + q * (2 * CT) + 2 * tid;
                if (WANT_LD) {
                    if (act1[q]) { D2 o = {ld.v[2 * q] - lsig, ld.v[2 * q + 1] - lsig}; *(D2*)(logdet + n) = o; }
                    else if (act0[q]) logdet[n] = ld.v[2 * q] - lsig;
                }
                if (sumsq) {
                    if (act1[q]) { D2 o = {ss.v[2 * q], ss.v[2 * q + 1]}; *(D2*)(sumsq + n) = o; }
                    else if (act0[q]) sumsq[n] = ss.v[2 * q];
                }
            }
            ctile += gridDim.x;
            k = k0;
        } else {
            ++k;
        }
        TTM_RAW_BARRIER();                                               // A(s + 1)
    }
}

// ---------------------------------------------------------------------------
// Table inverse with RESIDENT tables (the kernel the large-ensemble path runs).
//
// Round 1's kernel streamed the 12 KB table + bucket index of a component into LDS for every 512-row step: 945 MB of
// L2 -> LDS traffic per launch at C5 (1.5 x the algorithmic bytes), two loader waves per workgroup and one barrier
// per step.  Here a workgroup owns a contiguous chunk of rows and walks the components in BLOCKS of B: the tables of
// a block (xs row + 16-bit bucket index, 10 KB per component) are loaded into LDS once per workgroup and launch, then
// every tile of the chunk runs the B components against them.  Inside a block there are no barriers, no loader waves
// and no rings: a thread carries its NS rows through the B components, reads z_k with one 16-byte global load per row
// pair (issued one step ahead) and stores x_k the same way (one step behind).
//
// What a later component reads of the earlier ones (x_j and exp(-x_j^2/4) of the columns its nonmonotone groups use):
// BAND (every group reads column kc-1 or kc-2, columns consecutive - BASELINE config 5): the last two columns live in
// registers and the two register sets exchange roles from step to step; otherwise the planned column cache
// (termtable.py:_plan_column_cache) in per-thread 16-byte LDS words, 32 B per row and way.  One workgroup of 16 waves
// per CU either way.
// At a block boundary the state of a tile is re-loaded from the x columns the SAME thread stored in the previous block.
// With ETAB, exp(-x_k^2/4) comes from the located table interval instead of a full exp: x_k = y_lo + delta with
// 0 <= delta <= step, exp(-x_k^2/4) = E[i-1] exp(w), w = -delta (y_lo + x_k) / 4, |w| <= 0.1, E[i] = exp(-y_i^2/4)
// tabulated once per workgroup (the abscissae are the same for every component), exp(w) by its degree-7 Taylor
// polynomial (truncation 1.6e-16 relative for |y| <= 4; 3e-18 absolute everywhere).  The host enables it when the targets are clipped to the table
// (root_search_truncation) and step * max|y| / 2 <= 0.1.
// Search: the bucket function of k_table_index, bit for bit (table_bucket) -> a = number of entries in lower buckets; only
// the entries of the target's own bucket are compared (at most `per` = 2-3 with nb ~ T): exact without verification.
// LDS (doubles): [tables: B x tab_slot | {E_i, y_i}: 2 x Weven (ETAB) | column cache, 2 x ways x NS x blockDim]
// table slot: [lo, hi, bucket scale, bucket bias, int32 {entries per bucket at most, band code}, 0 | xs: W entries + 4
//              sentinels (+inf), rounded up to even | bucket index: nb + 1 uint16]
// Windowed tables (W < T): only entries [w0, w0 + W) of every table (and of E) are resident - the middle of the grid,
// where all but a handful per million of a standard-normal ensemble's rows land - so more components fit a block and
// the chunk is passed over fewer times.  The bucket index stays whole and holds entry numbers of the whole table; a
// row whose bucket starts outside [w0 + 1, w0 + W - 5) (its compares or its interval could leave the window) is an
// outlier: np.searchsorted over the table in memory, then the same interpolation.
// ---------------------------------------------------------------------------
// The bucket function of the table search, shared bit for bit by the index kernel and the lookup kernels (IEEE
// division, one fma): bucket(x) = clamp((int)fma(x, scale, bias), 0, nb - 1), scale = nb / (hi - lo), bias = -lo scale.
// It is monotone in x, so entries in a lower bucket than the target's are < target and entries in a higher one are
// > target: only the entries of the target's own bucket have to be compared.  A degenerate table (hi == lo, NaN) has
// scale 0: everything is bucket 0.
__device__ __forceinline__ void table_bucket_params(double lo, double hi, int nb, double& scale, double& bias) {
    scale = (double)nb / (hi - lo);
    if (!(scale > 0.0 && scale < 1.0e300)) scale = 0.0;
    bias = -lo * scale;
}
__device__ __forceinline__ int table_bucket(double x, double scale, double bias, int nb) {
    int q = (int)fma(x, scale, bias);
    q = max(q, 0);
    return min(q, nb - 1);
}

#define TTM_RT_HDR 6

// The planned column cache of k_inverse_rt: value i (x of way w at 2w, exp(-x^2/4) at 2w+1) of row pair q of a thread
// is ONE 16-byte LDS word {first row, second row} at cache[((i NP + q) CT + tid) 2]: consecutive lanes read consecutive
// words (ds_read_b128 / ds_write_b128 at full rate, no bank conflicts), half the LDS instructions of per-row columns.
template <int NP>
struct RtCache {
    double* base;                // cache + 2 tid
    int stride;                  // doubles between consecutive (value, pair) words: 2 CT
    __device__ __forceinline__ D2 get(int i, int q) const { return *(const D2*)(base + (size_t)(i * NP + q) * stride); }
    __device__ __forceinline__ void set(int i, int q, D2 v) const { *(D2*)(base + (size_t)(i * NP + q) * stride) = v; }
};

__device__ __forceinline__ D2 rt_expq(D2 x) {
    VecD<2> v;
    v.v[0] = x.x; v.v[1] = x.y;
    const VecD<2> e = exp_q_fast(v);
    D2 r = {e.v[0], e.v[1]};
    return r;
}

// entry state of the planned cache at a component from the columns already in X (block boundaries, conditional
// inverse): out of line, so that its exp() constants and addresses do not live in the registers of the step loop
template <int NP>
__device__ __attribute__((noinline)) void rt_warm(const int* state_, const char* X, int64_t ldb, unsigned int n0,
                                                  unsigned int last_pair, int CT, double* cache_base) {
    cint_p state = (cint_p)state_;
    RtCache<NP> cc{cache_base, 2 * CT};
    for (int w = 0; w < TTM_PLAN_WAYS; ++w) {
        const int v = state[w];
        if (v >= 0) {
            const char* col = X + (int64_t)(v & ~TTM_PLAN_E) * ldb;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                unsigned int n = n0 + (unsigned int)(q * 2 * CT);
                n = n < last_pair ? n : last_pair;
                const D2 x = *(const D2*)(col + (size_t)(n * 8u));
                cc.set(2 * w, q, x);
                const D2 zero = {0.0, 0.0};                  // (defined either way: the group evaluation reads both halves)
                cc.set(2 * w + 1, q, (v & TTM_PLAN_E) ? rt_expq(x) : zero);
            }
        }
    }
}

template <int NG, int CLS, int NS, bool ETAB, bool BAND>
__global__ __launch_bounds__(1024) void k_inverse_rt(const int* __restrict__ ucomp_, const int* __restrict__ ugrp_,
                                                     const double* __restrict__ U_, int64_t h_off,
                                                     int D, int k0, int k1,
                                                     const double* __restrict__ Z, int64_t ldz, double* X, int64_t ldx, int64_t N,
                                                     const double* __restrict__ tab_x, int T, double y0, double ystep, double ylast,
                                                     const double* __restrict__ tmin, const double* __restrict__ tmax,
                                                     const int* __restrict__ bkt, int nb, int truncate,
                                                     int tab_slot, int B, int ways, int64_t rows_per_wg, int w0, int W) {
    constexpr int NP = NS / 2;
    constexpr int DB = CLS == 1 ? 3 : (CLS == 2 ? 5 : 7), DA = CLS == 1 ? 1 : (CLS == 2 ? 5 : 7), GS = CLS == 1 ? 8 : (CLS == 2 ? 16 : 24);
    constexpr int HS = TTM_H_HDR + NG * GS;
    const int tid = threadIdx.x, CT = blockDim.x, ROWS = CT * NS;
    const int64_t c0 = (int64_t)blockIdx.x * rows_per_wg;
    if (c0 >= N) return;
    const int64_t c1 = c0 + rows_per_wg < N ? c0 + rows_per_wg : N;
    const int ntile = (int)((c1 - c0 + ROWS - 1) / ROWS);
    // resident part of every table: entries [w0, w0 + W) (the whole table when W == T), 4 sentinels behind it
    const int Weven = (W + 4 + 1) & ~1;
    double* tabs = g_smem;
    double* etab = tabs + (size_t)B * tab_slot;
    double* cache = etab + (ETAB ? 2 * Weven : 0);
    const double* etabw = etab - 2 * w0;                                 // pairs {E_i, y_i}, indexed by the entry's number in the whole table
    const RtCache<NP> cc{cache + 2 * tid, 2 * CT};
    // (row numbers are 32-bit - N < 2^28 - and every access is `uniform column base + 32-bit byte offset`)
    const unsigned int last_pair = (unsigned int)(((N + 1) & ~(int64_t)1) - 2);        // first row of the last readable pair
    const unsigned int row0 = (unsigned int)c0 + 2u * (unsigned int)tid, c1_32 = (unsigned int)c1;
    if (ETAB)
        for (int i = tid; i < W; i += CT) {
            // E of grid point w0 + i and, next to it, that point's abscissa exactly as the interpolation forms it from the
            // interval number (fma(i + 1, step, y0 - step)): one 16-byte read per row instead of a read, a convert and an fma
            etab[2 * i] = exp_q_fast(w0 + i == T - 1 ? ylast : (double)(w0 + i) * ystep + y0);
            etab[2 * i + 1] = fma((double)(w0 + i + 1), ystep, y0 - ystep);
        }
    // Taylor coefficients 1/7! .. 1/2! of the ETAB put, kept in VGPRs (as scalars they would push the kernel over the
    // SGPR budget and be spilled to VGPR lanes: v_readlane + hazard nops in the middle of every step)
    double kc[6];
    {
        const __attribute__((address_space(4))) double* kg = (const __attribute__((address_space(4))) double*)g_exp_coef;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            kc[j] = kg[6 + j];
            if (ETAB) asm volatile("" : "+v"(kc[j]));
        }
    }
    const int64_t ldzb = ldz * 8, ldxb = ldx * 8;
    const double y0m = y0 - ystep;
    const int kc_first = ((cint_p)ucomp_)[k0 * TTM_UC_LEN + TTM_UC_KC];    // BAND: column of component k is kc_first + k - k0
    D2 bx1[NP], be1[NP], bx2[NP], be2[NP];                               // BAND: x and exp(-x^2/4) of columns kc-1 and kc-2
#pragma unroll
    for (int q = 0; q < NP; ++q) { D2 zero = {0.0, 0.0}; bx1[q] = be1[q] = bx2[q] = be2[q] = zero; }

    for (int kb = k0; kb < k1; kb += B) {
        const int ke = kb + B < k1 ? kb + B : k1;
        const int nk = ke - kb;
        __syncthreads();                                                 // every wave is done with the previous block's tables
        if (tid < nk) {
            // header of the slot: search parameters (the index kernel's bucket function, bit for bit) and the largest
            // number of entries any bucket holds (filled in below)
            const int c = kb + tid;
            double* slot = tabs + (size_t)tid * tab_slot;
            const double lo = tmin[c - k0], hi = tmax[c - k0];
            double scale, bias;
            table_bucket_params(lo, hi, nb, scale, bias);
            slot[2] = scale; slot[3] = bias;                             // ([0], [1]: the resident search's target range, below)
            int code = 0;
            if (BAND) {                                                  // n_grp | lag of group 0 << 4 | lag of group 1 << 8
                const int* uc = ucomp_ + c * TTM_UC_LEN;
                const int n_grp = uc[TTM_UC_N_GRP];
                code = n_grp;
                for (int g = 0; g < n_grp && g < 2; ++g)
                    code |= (uc[TTM_UC_KC] - ugrp_[(uc[TTM_UC_GRP_OFF] + g) * TTM_UG_LEN + TTM_UG_VAR]) << (4 + 4 * g);
            }
            ((int*)slot)[8] = 0;                                         // entries per bucket, at most
            ((int*)slot)[9] = code;
            slot[5] = 0.0;                                               // int32 {degenerate, 0}
        }
        __syncthreads();
        // The tables of the block, sixteen components' loads in flight per thread: a loop over the components with one
        // load -> wait -> LDS store per pass costs a memory latency per component and kind - 80 latencies per workgroup
        // and launch at C5, a quarter of the kernel, while every wave waits at the barrier.
        for (int cg = 0; cg < nk; cg += 16) {
            for (int i = tid; i < Weven; i += CT) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = tab_x[(int64_t)(kb - k0 + min(cg + u, nk - 1)) * T + w0 + min(i, W - 1)];   // (clamped, unconditional)
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (cg + u < nk) tabs[(size_t)(cg + u) * tab_slot + TTM_RT_HDR + i] = i < W ? v[u] : INFINITY;
            }
            for (int i = tid; i <= nb; i += CT) {
                int v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = bkt[(int64_t)(kb - k0 + min(cg + u, nk - 1)) * (nb + 1) + i];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (cg + u < nk) ((unsigned short*)(tabs + (size_t)(cg + u) * tab_slot + TTM_RT_HDR + Weven))[i] = (unsigned short)v[u];
            }
        }
        __syncthreads();
        // entries per bucket, at most: one wave per component scans its bucket index in LDS
        for (int c = tid >> 6; c < nk; c += CT >> 6) {
            const unsigned short* bs = (const unsigned short*)(tabs + (size_t)c * tab_slot + TTM_RT_HDR + Weven);
            int per = 0;
            for (int i = tid & 63; i < nb; i += 64) {                    // (buckets with an entry in the window)
                const int b0 = bs[i], b1 = bs[i + 1];
                if (b1 > w0 && b0 < w0 + W) per = max(per, b1 - b0);
            }
            for (int o = 32; o > 0; o >>= 1) per = max(per, __shfl_xor(per, o));
            // Targets in [wl, wh] = [xs[w0 + per], xs[w0 + W - 2]] are searched in the resident window: a bucket that
            // holds entry w0 + per or a later one starts at a >= w0 + 1 (it has at most per entries), so the position p
            // lies in [w0 + 1, w0 + W - 2] and every entry read (a .. a + 3, p - 1, p) is resident or a sentinel.
            // Everything else - the tails of the table, NaN - is an outlier of the step (below).
            // A degenerate table (one bucket holds most of the window) sends every row there: bounds and bucket index are
            // overwritten so that the resident search stays inside the window whatever it is given.
            double* slot = tabs + (size_t)c * tab_slot;
            const int pm = max(per, 1);
            const bool deg = pm + 3 >= W;
            if ((tid & 63) == 0) {
                const double* xw = slot + TTM_RT_HDR;
                ((int*)slot)[8] = per;
                ((int*)slot)[10] = deg ? 1 : 0;
                slot[0] = deg ? xw[1] : xw[pm];
                slot[1] = deg ? xw[1] : xw[W - 2];
            }
            if (deg)
                for (int i = tid & 63; i <= nb; i += 64) const_cast<unsigned short*>(bs)[i] = (unsigned short)(w0 + 1);
        }
        __syncthreads();

        // Memory operations of a step, in issue order: [wait for z_s] -> store of x_{s-1} (deferred by one step) ->
        // load of z_{s+1} -> arithmetic.  The wait therefore only ever covers operations issued a whole step earlier
        // (the load of z_s and the store of x_{s-2}); a store issued at the END of its own step would sit between the
        // load and the next wait and stall every step for a full write latency (vmcnt counts in order).
        D2 zn[NP], rprev[NP];
        const char* xprev_col = nullptr;                                 // column of the deferred store (uniform)
        unsigned int xprev_off = 0;                                      // byte offset of this thread's first row in it
        bool xprev_full = true;
        {
            const char* zcol = (const char*)Z + (int64_t)(kb - k0) * ldzb;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                unsigned int n = row0 + (unsigned int)(q * 2 * CT);
                n = n < last_pair ? n : last_pair;
                zn[q] = *(const D2*)(zcol + (size_t)(n * 8u));
            }
        }
        for (int tile = 0; tile < ntile; ++tile) {
            const unsigned int tbase = row0 + (unsigned int)tile * (unsigned int)ROWS;     // this thread's first row of the tile
            const bool full = c0 + (int64_t)(tile + 1) * ROWS <= c1;     // (uniform) every row of the tile exists
            if (BAND && !(ntile == 1 && kb > k0)) {                      // the two columns in front of the block, if they exist
                // (a chunk of ONE tile keeps them in its registers from block to block: nothing to load)
                const int kcb = kc_first + (kb - k0);
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    unsigned int n = tbase + (unsigned int)(q * 2 * CT);
                    n = n < last_pair ? n : last_pair;
                    D2 zero = {0.0, 0.0};
                    bx1[q] = be1[q] = bx2[q] = be2[q] = zero;
                    if (kcb >= 1) {
                        bx1[q] = *(const D2*)((const char*)X + (int64_t)(kcb - 1) * ldxb + (size_t)(n * 8u));
                        be1[q] = rt_expq(bx1[q]);
                    }
                    if (kcb >= 2) {
                        bx2[q] = *(const D2*)((const char*)X + (int64_t)(kcb - 2) * ldxb + (size_t)(n * 8u));
                        be2[q] = rt_expq(bx2[q]);
                    }
                }
            } else if (kb > 0) {                                         // columns the earlier blocks (or the caller) left in X
                rt_warm<NP>(ucomp_ + TTM_UC_STATE(D, kb), (const char*)X, ldxb, tbase, last_pair, CT, cache + 2 * tid);
            }
            cdbl_p rec = (cdbl_p)(U_ + h_off) + (int64_t)kb * HS;
            const double* slot = tabs;
            const char* zcol = (const char*)Z + (int64_t)(kb - k0) * ldzb;   // column of THIS step's z
            // one step = one component for the rows of the tile.  Banded maps: the registers of the column at lag 2 take the
            // new column, so the two register sets swap roles from step to step - steps are issued in pairs with the roles
            // exchanged instead of moving 8 registers per row pair and step
            auto step = [&](int j, D2 (&L1x)[NP], D2 (&L1e)[NP], D2 (&L2x)[NP], D2 (&L2e)[NP]) {
                // ---- uniform data of the step: the whole record by scalar loads issued together --------------------
                cint_p ri = (cint_p)rec;
                const int put2 = ri[0], flg = ri[1], kcol = ri[3], n_grp = ri[13];
                const double nm0 = rec[7];
                int gslot[NG];
                double gB[NG][DB + 1], gA[NG][DA + 1];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    cdbl_p gr = rec + TTM_H_HDR + g * GS;
                    gslot[g] = ((cint_p)gr)[0];
#pragma unroll
                    for (int i = 0; i <= DB; ++i) gB[g][i] = gr[1 + i];
#pragma unroll
                    for (int i = 0; i <= DA; ++i) gA[g][i] = gr[2 + DB + i];
                }
                // ---- memory: z of this step has landed; x of the step before goes out; z of the step after comes in ----
                D2 zc[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) { zc[q] = zn[q]; asm volatile("" : "+v"(zc[q])); }
                if (xprev_col) {
                    if (xprev_full) {
#pragma unroll
                        for (int q = 0; q < NP; ++q)
                            *(D2*)(const_cast<char*>(xprev_col) + (size_t)(xprev_off + (unsigned int)(q * 2 * CT) * 8u)) = rprev[q];
                    } else {
#pragma unroll
                        for (int q = 0; q < NP; ++q) {
                            const unsigned int n = xprev_off / 8u + (unsigned int)(q * 2 * CT);
                            char* xp = const_cast<char*>(xprev_col) + (size_t)(n * 8u);
                            if (n + 1 < c1_32) *(D2*)xp = rprev[q];
                            else if (n < c1_32) *(double*)xp = rprev[q].x;
                        }
                    }
                }
                {
                    const bool wrap = j + 1 == nk;                       // next: first component of the next tile
                    const char* zc_next = wrap ? (const char*)Z + (int64_t)(kb - k0) * ldzb : zcol + ldzb;
                    const unsigned int tb = (wrap && tile + 1 < ntile) ? tbase + (unsigned int)ROWS : tbase;   // (past the last tile: a harmless re-read)
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        unsigned int n = tb + (unsigned int)(q * 2 * CT);
                        n = n < last_pair ? n : last_pair;
                        zn[q] = *(const D2*)(zc_next + (size_t)(n * 8u));
                    }
                }
                // ---- the table's search parameters (LDS broadcast reads) ---------------------------------------------
                double wl, wh, scale, bias;
                load_pair(slot, wl, wh);
                load_pair(slot + 2, scale, bias);
                const double* xsl = slot + TTM_RT_HDR - w0;              // indexed by the entry's number in the whole table
                const unsigned short* bkl = (const unsigned short*)(slot + TTM_RT_HDR + Weven);
                // ---- nonmonotone offset (h_offset's arithmetic, operand for operand) ------------------------------------
                double off[NS];
#pragma unroll
                for (int e = 0; e < NS; ++e) off[e] = nm0;
                auto group = [&](int g, const D2 (&xv)[NP], const D2 (&ev)[NP]) {
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const double x = h ? xv[q].y : xv[q].x, ee = h ? ev[q].y : ev[q].x;
                            double bq = gB[g][DB], aq = gA[g][DA];
#pragma unroll
                            for (int i = DB - 1; i >= 0; --i) bq = fma(bq, x, gB[g][i]);
#pragma unroll
                            for (int i = DA - 1; i >= 0; --i) aq = fma(aq, x, gA[g][i]);
                            off[2 * q + h] = fma(ee, bq, off[2 * q + h]) + aq;
                        }
                    }
                };
                if (BAND) {
                    const int code = __builtin_amdgcn_readfirstlane(((const int*)slot)[9]);
#pragma unroll
                    for (int g = 0; g < (NG < 2 ? NG : 2); ++g) {
                        if (g < (code & 15)) {
                            if (((code >> (4 + 4 * g)) & 15) == 1) group(g, L1x, L1e);
                            else group(g, L2x, L2e);
                        }
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        if (g < n_grp) {
                            D2 xv[NP], ev[NP];
#pragma unroll
                            for (int q = 0; q < NP; ++q) { xv[q] = cc.get(gslot[g], q); ev[q] = cc.get(gslot[g] + 1, q); }
                            group(g, xv, ev);
                        }
                    }
                }
                // ---- target, bucket, position ---------------------------------------------------------------------------
                // np.searchsorted(xs, target) (left) = a + #{entries of the target's own bucket that are < target}, a =
                // number of entries in lower buckets: the bucket function is monotone, so those are all < target and the
                // entries of higher buckets are all > target (k_table_index).  `per` entries are compared - the most any
                // bucket of this table holds (2-3 with nb ~ T) - what lies behind the bucket compares as not smaller, and
                // behind the table stand +inf sentinels: no verification, no second pass.
                const int per = __builtin_amdgcn_readfirstlane(((const int*)slot)[8]);
                double traw[NS], tg[NS];
                int pos[NS];
                // the resident search runs on the target clipped to [wl, wh]; a row whose target that changes (the tails of
                // the table, beyond the window, NaN) is an outlier: redone from the table in memory further down
                unsigned long long outl = __builtin_amdgcn_readfirstlane(((const int*)slot)[10]) ? ~0ull : 0ull;
#pragma unroll
                for (int e = 0; e < NS; ++e) {
                    const double z = (e & 1) ? zc[e >> 1].y : zc[e >> 1].x;
                    traw[e] = -off[e] + z;
                    tg[e] = fmin(fmax(traw[e], wl), wh);
                    outl |= __builtin_amdgcn_ballot_w64(traw[e] != tg[e]);
                    // (tg >= wl >= lo: the lower clamp of table_bucket cannot bind - a product that rounds below zero
                    // converts to 0)
                    pos[e] = (int)bkl[min((int)fma(tg[e], scale, bias), nb - 1)];
                }
                if (per <= 2) {
#pragma unroll
                    for (int e = 0; e < NS; ++e) {
                        const double* q4 = xsl + pos[e];
                        const double q0 = q4[0], q1 = q4[1];
                        pos[e] += (q0 < tg[e] ? 1 : 0) + (q1 < tg[e] ? 1 : 0);
                    }
                } else if (per <= 4) {
#pragma unroll
                    for (int e = 0; e < NS; ++e) {
                        const double* q4 = xsl + pos[e];
                        const double q0 = q4[0], q1 = q4[1], q2 = q4[2], q3 = q4[3];
                        pos[e] += (q0 < tg[e] ? 1 : 0) + (q1 < tg[e] ? 1 : 0) + (q2 < tg[e] ? 1 : 0) + (q3 < tg[e] ? 1 : 0);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < NS; ++e) {
                        int ae = pos[e];
                        for (int done = 0; done < per; done += 4) {      // (uniform trip count; a finished lane re-reads sentinels or larger entries)
                            const double* q4 = xsl + ae;
                            const double q0 = q4[0], q1 = q4[1], q2 = q4[2], q3 = q4[3];
                            const int c = (q0 < tg[e] ? 1 : 0) + (q1 < tg[e] ? 1 : 0) + (q2 < tg[e] ? 1 : 0) + (q3 < tg[e] ? 1 : 0);
                            ae += c;
                            if (__builtin_amdgcn_ballot_w64(c == 4) == 0) break;
                        }
                        pos[e] = ae;
                    }
                }
                // ---- interp1d slope form (TM:4062-4065) and, with ETAB, exp(-x^2/4) from the located interval -------------
                double r[NS], ev[NS];
                auto interp = [&](double y_lo, double x_lo, double x_hi, double e_lo, double tgt, double& rr, double& ee) {
                    // (y_lo: abscissa i - 1 of the grid, fma(i, step, y0 - step) - the grid point to an ulp)
                    // slope = step / (x_hi - x_lo): v_rcp_f64 (2^-23) with one Newton step (2^-46); its error moves x by
                    // 0.02 x 2^-46 = 3e-16 at most.  y_hi - y_lo is the grid step up to the rounding of the two abscissae
                    // (1e-13 of the step, also in the last interval, whose end point np.linspace forces: 2e-15 of x)
                    const double dx = fmax(x_hi - x_lo, 1e-300);             // (tie at a flat start: table_lookup)
                    double rc = approx_rcp(dx);
                    rc = fma(fma(-dx, rc, 1.0), rc, rc);
                    const double slope = ystep * rc;
                    const double delta = slope * (tgt - x_lo);
                    rr = delta + y_lo;
                    if (ETAB) {
                        // exp(w), |w| <= 0.1 (0.04 for |y| <= 4), by its degree-7 Taylor polynomial: truncation w^8 / 8! is
                        // 1.6e-16 relative at |y| = 4 and at most 2.5e-13 relative where exp(-y^2/4) itself is 1e-11
                        const double w = (delta * -0.25) * (y_lo + rr);
                        double p = kc[0];
#pragma unroll
                        for (int i2 = 1; i2 < 6; ++i2) p = fma(p, w, kc[i2]);
                        p = fma(p, w, 1.0);
                        p = fma(p, w, 1.0);
                        ee = e_lo * p;
                    }
                };
#pragma unroll
                for (int e = 0; e < NS; ++e) {
                    const int i = pos[e];                    // (in [w0 + 1, w0 + W - 2] for a target in [wl, wh])
                    double e_lo = 0.0, y_lo;
                    if (ETAB) load_pair(etabw + 2 * (i - 1), e_lo, y_lo);
                    else y_lo = fma((double)i, ystep, y0m);
                    interp(y_lo, xsl[i - 1], xsl[i], e_lo, tg[e], r[e], ev[e]);
                }
                if (outl != 0) {
                    // outliers (a handful per million rows of a standard-normal ensemble): clip as TM:4074-4076 does,
                    // np.searchsorted (left) over the whole row in memory, the same interpolation
                    const double* xg = tab_x + (int64_t)(kb - k0 + j) * T;
                    const bool deg = __builtin_amdgcn_readfirstlane(((const int*)slot)[10]) != 0;
                    const double lo = tmin[kb - k0 + j], hi = tmax[kb - k0 + j];
#pragma unroll
                    for (int e = 0; e < NS; ++e) {
                        if (deg || traw[e] != tg[e]) {
                            double t = traw[e];
                            if (ETAB || truncate) {          // a NaN target stays NaN (fmin / fmax drop it)
                                                             // (ETAB is only instantiated for clipped searches)
                                const double cl = fmin(fmax(t, lo), hi);
                                t = t != t ? t : cl;
                            }
                            int a = 0, b = T;
                            while (a < b) {
                                const int mid = (a + b) >> 1;
                                if (xg[mid] < t) a = mid + 1; else b = mid;
                            }
                            const int i = min(max(a, 1), T - 1);
                            interp(fma((double)i, ystep, y0m), xg[i - 1], xg[i], ETAB ? exp_q_fast((double)(i - 1) * ystep + y0) : 0.0, t, r[e],
                                   ev[e]);
                        }
                    }
                }
                // ---- keep x_k (and exp(-x_k^2/4)) for the components behind; the store is deferred to the next step -----
#pragma unroll
                for (int q = 0; q < NP; ++q) { D2 o = {r[2 * q], r[2 * q + 1]}; rprev[q] = o; }
                if (BAND) {
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        D2 eo = {ev[2 * q], ev[2 * q + 1]};
                        L2x[q] = rprev[q]; L2e[q] = ETAB ? eo : rt_expq(rprev[q]);     // the oldest column's registers take the new one
                    }
                } else if (put2 >= 0) {
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        cc.set(put2, q, rprev[q]);
                        if (flg & 1) {
                            D2 eo = {ev[2 * q], ev[2 * q + 1]};
                            cc.set(put2 + 1, q, ETAB ? eo : rt_expq(rprev[q]));
                        } else {
                            const D2 zero = {0.0, 0.0};      // (defined: read next to x by the groups of later components)
                            cc.set(put2 + 1, q, zero);
                        }
                    }
                }
                xprev_col = (const char*)X + (int64_t)kcol * ldxb;
                xprev_off = tbase * 8u;
                xprev_full = full;
                rec += HS; slot += tab_slot; zcol += ldzb;
            };
            int j = 0;
#ifndef TTM_RT_UNPAIRED
            for (; j + 1 < nk; j += 2) {
                step(j, bx1, be1, bx2, be2);
                step(j + 1, bx2, be2, bx1, be1);
            }
#else
            for (; j + 1 < nk; ++j) {                                    // (tuning builds: the register shift per step)
                step(j, bx1, be1, bx2, be2);
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const D2 tx = bx1[q], te = be1[q];
                    bx1[q] = bx2[q]; be1[q] = be2[q]; bx2[q] = tx; be2[q] = te;
                }
            }
#endif
            if (j < nk) {
                step(j, bx1, be1, bx2, be2);
                if (BAND) {                                              // odd number of steps: back to the canonical roles
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        const D2 tx = bx1[q], te = be1[q];
                        bx1[q] = bx2[q]; be1[q] = be2[q]; bx2[q] = tx; be2[q] = te;
                    }
                }
            }
        }
        if (xprev_col) {                                                 // the last step's x
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const unsigned int n = xprev_off / 8u + (unsigned int)(q * 2 * CT);
                char* xp = const_cast<char*>(xprev_col) + (size_t)(n * 8u);
                if (n + 1 < c1_32) *(D2*)xp = rprev[q];
                else if (n < c1_32) *(double*)xp = rprev[q].x;
            }
        }
    }
}

// basis matrices of one component
__global__ __launch_bounds__(256) void k_basis(DevProg P, int k, int which, const double* __restrict__ X, int64_t ldx,
                                               int64_t N, double* __restrict__ out, int64_t ldo) {
    double* slots;
    CacheStore<double> cst;
    const Prog g = make_prog_lds(P, cst, slots);
    const Comp c = comp_at(P, k, 0, nullptr, nullptr);
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        XSoA x{X, ldx, n};
        sample_basis(c, g, which, x, [&](int i, double v) { out[(int64_t)i * ldo + n] = v; });
    }
}

// ---------------------------------------------------------------------------
// K4: table inverse
// ---------------------------------------------------------------------------

// blockIdx.y = component k0 + y, blockIdx.x over table points
__global__ __launch_bounds__(256) void k_table_build(DevProg P, int k0, const double* __restrict__ coef,
                                                     const double* __restrict__ fold,
                                                     const double* __restrict__ pts, int T, double* __restrict__ out) {
    double* slots;
    CacheStore<double> cst;
    Prog g = make_prog_lds(P, cst, slots);
    g.mono = TTM_MONO_SEPARABLE;
    LdsSlots w{slots + threadIdx.x, (int)blockDim.x};
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = k0 + blockIdx.y;
    const Comp c = comp_at(P, k, 0, coef, fold);
    if (i < T) {
        const double t = pts[i];
        double v, dv;
        if (c.n_mnt == 0 && c.n_xgrp == 0) {
            const UniformW uw{c.fold + c.off_wb};
            g_eval<false>(c, g, t, uw, v, dv);
        } else {
            XFake x{c.kc, t};
            mon_weights<double>(c, g, x, w);
            g_eval<false>(c, g, t, w, v, dv);
        }
        out[(int64_t)blockIdx.y * T + i] = v;
    }
}

// tmin / tmax / sortedness / bucket index of one table per block (blockIdx.x = component)
__global__ __launch_bounds__(256) void k_table_index(const double* __restrict__ tab_x, int T, int nb,
                                                     double* __restrict__ tmin, double* __restrict__ tmax,
                                                     int* __restrict__ bkt, int* __restrict__ unsorted) {
    __shared__ double xs[2048];
    __shared__ int bq[2048];
    __shared__ int bad;
    const double* row = tab_x + (int64_t)blockIdx.x * T;
    if (threadIdx.x == 0) bad = 0;
    for (int i = threadIdx.x; i < T; i += blockDim.x) xs[i] = row[i];
    __syncthreads();
    int mybad = 0;
    for (int i = threadIdx.x + 1; i < T; i += blockDim.x) mybad |= !(xs[i - 1] <= xs[i]);   // also flags NaN
    if (mybad) atomicOr(&bad, 1);
    const double lo = xs[0], hi = xs[T - 1];
    double scale, bias;
    table_bucket_params(lo, hi, nb, scale, bias);
    for (int i = threadIdx.x; i < T; i += blockDim.x) bq[i] = table_bucket(xs[i], scale, bias, nb);
    __syncthreads();
    if (threadIdx.x == 0) { tmin[blockIdx.x] = lo; tmax[blockIdx.x] = hi; unsorted[blockIdx.x] = bad; }
    // bkt[q] = number of entries in buckets below q (the entries are sorted, so their bucket numbers are too)
    for (int q = threadIdx.x; q <= nb; q += blockDim.x) {
        int a = 0, b = T;
        if (q == nb) a = T;
        else if (q > 0) {
            while (a < b) {
                const int mid = (a + b) >> 1;
                if (bq[mid] < q) a = mid + 1; else b = mid;
            }
        }
        bkt[(int64_t)blockIdx.x * (nb + 1) + q] = a;
    }
}

// resident-table image (csrc/ttm_band_image.h) of one table per workgroup from its row, range and bucket index in memory
// (the fused kernel below writes it from what it holds in LDS; this one serves the two-launch path)
__global__ __launch_bounds__(256) void k_table_image(const double* __restrict__ tab_x, int T, int nb, const double* __restrict__ tmin,
                                                     const double* __restrict__ tmax, const int* __restrict__ bkt, double* __restrict__ img,
                                                     int w0, int W, int slot) {
    __shared__ double xs[2048];
    __shared__ int bks[1024];
    __shared__ int red;
    const double* row = tab_x + (int64_t)blockIdx.x * T;
    for (int i = threadIdx.x; i < T; i += blockDim.x) xs[i] = row[i];
    for (int q = threadIdx.x; q <= nb; q += blockDim.x) bks[q] = bkt[(int64_t)blockIdx.x * (nb + 1) + q];
    __syncthreads();
    band_image_write(xs, bks, &red, T, nb, tmin[blockIdx.x], tmax[blockIdx.x], w0, W, (W + 4 + 1) & ~1, slot, img + (int64_t)blockIdx.x * slot);
}

// k_table_build + k_table_index as ONE launch, one workgroup per component (blockIdx.x = component k0 + x): the table's T
// points by the same evaluator as k_table_build (the same bits), kept in LDS for the index phase of k_table_index behind a
// barrier - a new coefficient vector's inverse tables cost one launch instead of two (6.8 + 6.9 us and the gap).
// what one workgroup does for the inverse table of component k (row krel of the outputs): the first ebd threads evaluate (the
// evaluator's LDS image is that of a workgroup of ebd threads), every thread of the workgroup takes part in the index phases
__device__ __forceinline__ void table_body(const DevProg& P, int k, int krel, int ebd, const double* __restrict__ coef,
                                           const double* __restrict__ fold, const double* __restrict__ pts, int T, int nb,
                                           double* __restrict__ out, double* __restrict__ tmin, double* __restrict__ tmax,
                                           int* __restrict__ bkt, int* __restrict__ unsorted, int* unsorted_host, double* __restrict__ img,
                                           int img_w0, int img_W, int img_slot) {
    __shared__ double xs[2048];
    __shared__ int bq[2048];
    __shared__ int bks[1024];
    __shared__ int bad;
    double* slots;
    CacheStore<double> cst;
    Prog g = make_prog_lds_n(P, cst, slots, ebd);
    g.mono = TTM_MONO_SEPARABLE;
    const int tid = threadIdx.x, bd = blockDim.x;
    LdsSlots w{slots + tid, ebd};
    const Comp c = comp_at(P, k, 0, coef, fold);
    if (tid == 0) bad = 0;
    if (tid < ebd)
        for (int i = tid; i < T; i += ebd) {
            const double t = pts[i];
            double v, dv;
            if (c.n_mnt == 0 && c.n_xgrp == 0) {
                const UniformW uw{c.fold + c.off_wb};
                g_eval<false>(c, g, t, uw, v, dv);
            } else {
                XFake x{c.kc, t};
                mon_weights<double>(c, g, x, w);
                g_eval<false>(c, g, t, w, v, dv);
            }
            out[(int64_t)krel * T + i] = v;
            xs[i] = v;
        }
    __syncthreads();
    int mybad = 0;
    for (int i = tid + 1; i < T; i += bd) mybad |= !(xs[i - 1] <= xs[i]);   // also flags NaN
    if (mybad) atomicOr(&bad, 1);
    const double lo = xs[0], hi = xs[T - 1];
    double scale, bias;
    table_bucket_params(lo, hi, nb, scale, bias);
    for (int i = tid; i < T; i += bd) bq[i] = table_bucket(xs[i], scale, bias, nb);
    __syncthreads();
    if (tid == 0) {
        tmin[krel] = lo; tmax[krel] = hi; unsorted[krel] = bad;
        if (unsorted_host) __hip_atomic_store(unsorted_host + krel, bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (int q = tid; q <= nb; q += bd) {
        int a = 0, b = T;
        if (q == nb) a = T;
        else if (q > 0) {
            while (a < b) {
                const int mid = (a + b) >> 1;
                if (bq[mid] < q) a = mid + 1; else b = mid;
            }
        }
        bkt[(int64_t)krel * (nb + 1) + q] = a;
        if (img) bks[q] = a;                                              // (nb + 1 = 1024: the host checked)
    }
    // the resident-table image of the component for the banded-map lookup kernel (csrc/ttm_band_image.h)
    if (img) {
        __syncthreads();
        band_image_write(xs, bks, &bad, T, nb, lo, hi, img_w0, img_W, (img_W + 4 + 1) & ~1, img_slot, img + (int64_t)krel * img_slot);
    }
}

// k_table_build + k_table_index as ONE launch, one workgroup per component (blockIdx.x = component k0 + x): the table's T
// points by the same evaluator as k_table_build (the same bits), kept in LDS for the index phase of k_table_index behind a
// barrier - a new coefficient vector's inverse tables cost one launch instead of two (6.8 + 6.9 us and the gap).
__global__ __launch_bounds__(256) void k_table_build_index(DevProg P, int k0, const double* __restrict__ coef,
                                                           const double* __restrict__ fold, const double* __restrict__ pts, int T,
                                                           int nb, double* __restrict__ out, double* __restrict__ tmin,
                                                           double* __restrict__ tmax, int* __restrict__ bkt, int* __restrict__ unsorted,
                                                           int* unsorted_host, double* __restrict__ img, int img_w0, int img_W,
                                                           int img_slot) {
    table_body(P, k0 + (int)blockIdx.x, (int)blockIdx.x, (int)blockDim.x, coef, fold, pts, T, nb, out, tmin, tmax, bkt, unsorted, unsorted_host,
               img, img_w0, img_W, img_slot);
}

// A NEW COEFFICIENT VECTOR IN ONE LAUNCH: workgroups [0, D) do what k_uform does (fold, U section, records), workgroups [D, 2 D)
// build the inverse table of component x - D as k_table_build_index does.  The tables need only the FOLDED coefficients, which
// the table workgroup folds for itself into a scratch copy (fold2: the same layout, the same bits; nobody else touches its slice),
// so the two halves share no data and run side by side: the launch takes as long as the slower half (20 us at C5) instead of
// the sum (20 + 16 us and a gap).  The evaluator reads coefficients and folded sums through the scalar cache, which knows
// nothing of the vector stores that just wrote them: invalidated between the two.
__global__ __launch_bounds__(1024) void k_setup(DevProg P, UTabs T, const double* __restrict__ coef_src, double* coef,
                                                double* __restrict__ fold, double* __restrict__ fold2, double* __restrict__ U,
                                                int64_t err_off, int64_t h_off, int h_cls, int h_ng, int64_t p_off, int p_lag, int p_stride,
                                                double* err_host, const double* __restrict__ pts, int Tn, int nb, double* __restrict__ out,
                                                double* __restrict__ tmin, double* __restrict__ tmax, int* __restrict__ bkt,
                                                int* __restrict__ unsorted, int* unsorted_host, double* __restrict__ img, int img_w0,
                                                int img_W, int img_slot) {
    if ((int)blockIdx.x < P.D) {
        uform_body(P, T, (int)blockIdx.x, coef_src, coef, fold, U, err_off, h_off, h_cls, h_ng, p_off, p_lag, p_stride, err_host);
        return;
    }
    const int k = (int)blockIdx.x - P.D, tid = threadIdx.x, bd = blockDim.x;
    const int D1 = P.D + 1;
    const int* off = P.off;
    if (coef_src) {
        // (the uform workgroup of the component copies the same values to the same place)
        for (int i = off[2 * D1 + k] + tid; i < off[2 * D1 + k + 1]; i += bd) coef[i] = coef_src[i];
        __syncthreads();
    }
    fold_coeffs(P.itab + off[k], P.ftab + off[4 * D1 + k], P.dpar + off[D1 + k], coef + off[2 * D1 + k], fold2 + off[3 * D1 + k], tid, bd);
    __syncthreads();
    fold_st8(P.fdesc + k * TTM_FDESC_LEN, P.fints, fold2 + off[3 * D1 + k], tid, bd);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    table_body(P, k, k, 256, coef, fold2, pts, Tn, nb, out, tmin, tmax, bkt, unsorted, unsorted_host, img, img_w0, img_W, img_slot);
}

// One sample per thread, no workgroup-level staging: the 1001-point tables of all components stay in
// L2 / L1 (8 KB each, shared by every wave) and each sample gathers the handful of entries its search
// needs: one 16-byte gather of the bucket index narrows the range to a few entries, two or three
// gathers bisect it, one 16-byte gather fetches the bracketing pair.  Waves never synchronise, so the
// gather latency is hidden by occupancy instead of barriers and prefetch registers.  The thread walks
// the components in order; its column cache serves the just-solved x_j to the following components.
// PLAN: every component of [k0,k1) takes the fast path -> statically planned column cache, family FAM.
template <int NS, bool PLAN, int FAM>
__global__ __launch_bounds__(256) void k_inverse_table(DevProg P, int k0, int k1, const double* __restrict__ coef,
                                                       const double* __restrict__ fold,
                                                       const double* __restrict__ Z, int64_t ldz,
                                                       double* X, int64_t ldx, int64_t N,
                                                       const double* __restrict__ tab_x, const double* __restrict__ tab_y,
                                                       int64_t ldy, int T, int yreg, double y0, double ystep, double ylast,
                                                       const double* __restrict__ tmin, const double* __restrict__ tmax,
                                                       const int* __restrict__ bkt, int nb, int truncate) {
    typedef typename real_of<NS>::type R;
    double* unused;
    CacheStore<R> cst;
    const Prog g = make_prog_lds(P, cst, unused);
    const int bd = blockDim.x, tid = threadIdx.x;
    const int nbk = nb + 1;
    for (int64_t tile = (int64_t)blockIdx.x * NS * bd; tile < N; tile += (int64_t)gridDim.x * NS * bd) {
        XSoAN<NS> xa;
        xa.X = X; xa.ld = ldx;
        bool act[NS];
#pragma unroll
        for (int e = 0; e < NS; ++e) {
            const int64_t n = tile + (int64_t)e * bd + tid;
            act[e] = n < N;
            xa.n[e] = act[e] ? n : N - 1;
        }
        VarCache<XSoAN<NS>, R> x(xa, cst);
        PlanCache<XSoAN<NS>, R> xp(xa, cst);
        if (PLAN && k0 > 0) xp.warm((cint_p)P.fints + ((cint_p)P.fdesc)[k0 * TTM_FDESC_LEN + TTM_FD_PLAN_OFF]);
        R z_next;
#pragma unroll
        for (int e = 0; e < NS; ++e) set_elem(z_next, e, Z[xa.n[e]]);
        for (int k = k0; k < k1; ++k) {
            const R zk = z_next;
            if (k + 1 < k1) {
#pragma unroll
                for (int e = 0; e < NS; ++e) set_elem(z_next, e, Z[(int64_t)(k + 1 - k0) * ldz + xa.n[e]]);
            }
            cint_p fd = (cint_p)P.fdesc + k * TTM_FDESC_LEN;
            const int kc = fd[TTM_FD_KC];
            const double* xs = tab_x + (int64_t)(k - k0) * T;
            const double* ys = tab_y + (int64_t)(k - k0) * ldy;
            const int* bk = bkt + (int64_t)(k - k0) * nbk;
            const double lo = ((cdbl_p)tmin)[k - k0], hi = ((cdbl_p)tmax)[k - k0];
            const double scale = (double)nb / (hi - lo);
            const bool use_bkt = scale > 0.0 && scale < 1.0e300 && nb >= 4;
            R off;
            if (PLAN) {
                off = nonmon_sum_fast<FAM, R>(make_fast(fd, (cint_p)P.fints, (cdbl_p)fold, 0), g, xp);
            } else if (!fd[TTM_FD_COMPLEX]) {
                TaggedFetch<XSoAN<NS>, R> xf{x};
                off = nonmon_sum_fast<-1, R>(make_fast(fd, (cint_p)P.fints, (cdbl_p)fold, 0), g, xf);
            } else {
                const Comp c = comp_at(P, k, 0, coef, fold);
                off = nonmon_sum<R>(c, g, x);
            }
            R r;
#pragma unroll
            for (int e = 0; e < NS; ++e) {
                double target = -elem(off, e) + elem(zk, e);
                if (truncate) {                      // TM:4074-4076 (comparisons keep NaN untouched)
                    if (target < lo) target = lo;
                    if (target > hi) target = hi;
                }
                // np.searchsorted(xs, target) (left): bisect inside the buckets around the target only
                // (a 64-byte window + count variant was measured slower than these 2-3 dependent gathers)
                int a = 0, b = T;
                if (use_bkt) {
                    int q = (int)((target - lo) * scale);
                    q = q < 1 ? 1 : (q > nb - 2 ? nb - 2 : q);
                    a = bk[q - 1];                   // bucket edges q-1 .. q+2 bracket the target
                    b = bk[q + 2];
                }
                while (a < b) {
                    const int mid = (a + b) >> 1;
                    if (xs[mid] < target) a = mid + 1; else b = mid;
                }
                const int i = a < 1 ? 1 : (a > T - 1 ? T - 1 : a);
                const double x_lo = xs[i - 1], x_hi = xs[i];
                double y_lo, y_hi;
                if (yreg) {                          // np.linspace: y_i = i*step + y0, last point exact
                    y_lo = (double)(i - 1) * ystep + y0;
                    y_hi = (i == T - 1) ? ylast : (double)i * ystep + y0;
                } else {
                    y_lo = ys[i - 1]; y_hi = ys[i];
                }
                const double slope = fast_div(y_hi - y_lo, fmax(x_hi - x_lo, 1e-300));          // interp1d slope form (TM:4062-4065)
                const double re = slope * (target - x_lo) + y_lo;
                set_elem(r, e, re);
                if (act[e]) X[(int64_t)kc * ldx + xa.n[e]] = re;
            }
            if (PLAN) xp.put(fd[TTM_FD_KC_SLOT], r); else x.put(kc, r);
        }
    }
}

// ---------------------------------------------------------------------------
// K5: bisection inverse
// ---------------------------------------------------------------------------

template <int MONO, bool NEWTON>
__global__ __launch_bounds__(256) void k_inverse_bisect(DevProg P, int k0, int k1, const double* __restrict__ coef,
                                                        const double* __restrict__ fold,
                                                        const double* __restrict__ Z, int64_t ldz,
                                                        double* X, int64_t ldx, int64_t N,
                                                        int* __restrict__ iters, const int* __restrict__ cap) {
    double* slots;
    CacheStore<double> cst;
    const Prog g = make_prog_lds(P, cst, slots);
    LdsSlots w{slots + threadIdx.x, (int)blockDim.x};
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
                const double r = sample_root<MONO, NEWTON>(c, g, x, w, off, zk, capk, it);
                X[(int64_t)c.kc * ldx + n] = r;
                x.put(c.kc, r);
            }
            // wave-level max, one atomic per wave
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) it = max(it, __shfl_down(it, off, 64));
            if ((threadIdx.x & 63) == 0 && it > 0) atomicMax(&iters[k - k0], it);
        }
    }
}

// ---------------------------------------------------------------------------
// K6/K7: objective + gradient partial sums ; K8: Gram partial sums
// ---------------------------------------------------------------------------


// LDS: erf table | per-thread columns [scratch (nscr) | acc (nacc)]
__global__ __launch_bounds__(256) void k_objective(DevProg P, int k, const double* __restrict__ coef_k,
                                                   const double* __restrict__ fold_k,
                                                   const double* __restrict__ X, int64_t ldx, int64_t N,
                                                   int nscr, int nacc, double* __restrict__ partial,
                                                   unsigned int* __restrict__ counter, double* __restrict__ out,
                                                   double* flag, double mark) {
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
    double* accbase = slots + (size_t)nscr * bd;                         // nw rows of nacc running sums
    for (int i = tid; i < nw * nacc; i += bd) accbase[i] = 0.0;
    __syncthreads();
    WaveAcc acc{accbase + wv * nacc, true, lane == 63};
    const bool shared_bv = g.mono == TTM_MONO_INTEGRATED && dense_B(c);      // (the host sized the scratch by the same rule)
    LdsSlots w{slots + tid, bd};
    LdsSlots Bv{slots + (shared_bv ? (size_t)0 : (size_t)nb1 * bd) + tid, bd};
    LdsSlots I{slots + (size_t)(shared_bv ? 1 : 2) * nb1 * bd + tid, bd};
    // (uniform trip count: the lanes of a wave add together; a lane beyond the ensemble repeats the last sample and adds zero)
    for (int64_t n0 = (int64_t)blockIdx.x * bd; n0 < N; n0 += (int64_t)gridDim.x * bd) {
        const int64_t n = n0 + tid;
        acc.active = n < N;
        const XSoA xa{X, ldx, acc.active ? n : N - 1};
        VarCache<XSoA, double> x(xa, cst);
        if (g.mono == TTM_MONO_SEPARABLE) sample_objective_sep(c, g, x, w, acc);
        else sample_objective_int(c, g, x, w, Bv, I, acc);
    }
    __syncthreads();
    // block sums: the waves' rows in wave order
    for (int i = tid; i < nacc; i += bd) {
        double v = 0.0;
        for (int wq = 0; wq < nw; ++wq) v += accbase[wq * nacc + i];
        partial[(int64_t)blockIdx.x * nacc + i] = v;
    }
    if (out) {
        // single-launch variant: the workgroup that draws the last ticket adds the partials up, in the very order
        // k_reduce_partials uses (bit-identical sums), and writes the result - `out` may be pinned host memory
        if (last_workgroup(counter)) {
            double* fin = slots;                  // (the per-thread columns are free: every thread is past its partial sums;
                                                  // a static array here would sit on top of a dynamic image sized to the budget)
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

// coefficients handed over BY VALUE (kernel arguments: no host-to-device copy, no extra launch): block 0 writes
// them to the workspace and folds them
struct HostCoef { double c[TTM_HOSTCOEF_MAX]; };

__global__ __launch_bounds__(64) void k_fold_host(DevProg P, int k, HostCoef hc, int n, double* __restrict__ coef_dev,
                                                  double* __restrict__ fold) {
    const int D1 = P.D + 1;
    const int* off = P.off;
    for (int i = threadIdx.x; i < n; i += blockDim.x) coef_dev[i] = hc.c[i];
    __threadfence_block();
    __syncthreads();
    fold_coeffs(P.itab + off[k], P.ftab + off[4 * D1 + k], P.dpar + off[D1 + k], coef_dev, fold, threadIdx.x, blockDim.x);
    __syncthreads();
    fold_st8(P.fdesc + k * TTM_FDESC_LEN, P.fints, fold, threadIdx.x, blockDim.x);
}

// The evaluation of a host optimiser loop with NO ticket, NO fence and NO completion mark (grids of up to 128 workgroups: the
// filter's ensembles, where an evaluation is a chain of device-scope round trips - drain the stores, draw a ticket, two batches
// of loads, drain the results, the mark: 8.9 us of which 3 are arithmetic).  The rows of partial sums are SELF-VALIDATING: every
// slot holds a signalling-NaN pattern no arithmetic produces (TTM_SENT_BITS) until its workgroup stores the sum - one 8-byte
// agent-scope store, whole or absent.  Workgroup 0 stores its own row and then polls the others' slots until none holds the
// pattern (every wait bounded), adds the rows in a fixed order (wave i mod 4 takes sum i: lanes = rows, DPP-free shuffle sum),
// hands the slots back with the pattern in them for the next evaluation, and writes the results to `out` - page-locked host
// memory whose slots the HOST has filled with the same pattern and polls the same way (csrc/ttm_optim.cpp: poll_values), so
// no mark and no drain in front of it.  A wait that runs out writes TTM_SENT_FAIL instead (the host then fails the loop).
// partial: gridDim.x rows of 1 + M slots, all TTM_SENT_BITS at entry (ttm_sentinel_fill before the first evaluation), and at exit.
#define TTM_SENT_BITS 0x7FF4DEADBEEF0001ull
#define TTM_SENT_FAIL 0x7FF4DEADBEEF0002ull
#define TTM_SENT_WGS 128
#define TTM_SENT_SPINS (1 << 16)

// (all threads of workgroup 0 call, behind the coherent_store of its own row; all: LDS, gridDim.x x nacc doubles; fin: nacc)
__device__ __forceinline__ void sentinel_finish(double* __restrict__ partial, int nacc, double* all, double* fin, double* __restrict__ out,
                                                bool give_up) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int total = (int)gridDim.x * nacc;
    const double sent = __longlong_as_double((long long)TTM_SENT_BITS);
    bool done = false;
    for (int spin = 0; spin < (give_up ? 0 : TTM_SENT_SPINS) && !done; ++spin) {
        int ok = 1;
        for (int i = tid; i < total; i += (int)blockDim.x) {
            const double v = coherent_load(partial + i);
            all[i] = v;
            ok &= (unsigned long long)__double_as_longlong(v) != TTM_SENT_BITS ? 1 : 0;
        }
        done = __syncthreads_and(ok) != 0;
    }
    for (int i = tid; i < total; i += (int)blockDim.x) coherent_store(partial + i, sent);           // (the slots of the next evaluation)
    if (!done) {
        if (tid < nacc) __hip_atomic_store(out + tid, __longlong_as_double((long long)TTM_SENT_FAIL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    // the order of add_rows / finish_sums (csrc/ttm_dev.h), so that both finishes make the same bits: wave w adds rows w, w + nw,
    // ... alternately into two accumulators, then the waves' sums in wave order
    const int nw = (int)(blockDim.x >> 6);
    __syncthreads();                                  // (`all` is complete; fin doubles as the waves' scratch rows below)
    double mine = 0.0;
    if (lane < nacc) {
        double a0 = 0.0, a1 = 0.0;
        int r = wv;
        for (; r + nw < (int)gridDim.x; r += 2 * nw) { a0 += all[r * nacc + lane]; a1 += all[(r + nw) * nacc + lane]; }
        if (r < (int)gridDim.x) a0 += all[r * nacc + lane];
        mine = a0 + a1;
    }
    __syncthreads();                                  // (everybody has read `all`: its head is reused for the waves' sums)
    if (lane < nacc) all[wv * nacc + lane] = mine;
    __syncthreads();
    if (tid < nacc) {
        double v = 0.0;
        for (int w = 0; w < nw; ++w) v += all[w * nacc + tid];
        fin[tid] = v;
        __hip_atomic_store(out + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(256) void k_fill_bits(unsigned long long* __restrict__ p, int n, unsigned long long bits) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = bits;
}

// Separable objective from a cached derivative basis (TM:2978-3018 with the der_Psi_mon the reference precalculates,
// TM:789-821): dPsi is m rows of N doubles (ttm_basis(which = 2)), constant while a component is optimised, so one
// evaluation is a streaming pass: dS = dPsi.c + delta rowsum(dPsi); acc[0] += log dS, acc[1+i] += dPsi_i / dS.
// Coefficients travel as kernel arguments; block partials, then the last workgroup finishes (fixed order).
#define TTM_SEPC_MAXM 16
struct SepCoef { double c[TTM_SEPC_MAXM]; };

template <int M>
__global__ __launch_bounds__(256) void k_objective_sep_cached(const double* __restrict__ dPsi, int64_t ldp, int64_t N,
                                                              SepCoef hc, double delta, double* __restrict__ partial,
                                                              unsigned int* __restrict__ counter, double* __restrict__ out,
                                                              double* flag, double mark, int sentinel) {
    // M is a template parameter: the M column loads of a row are issued together (a run-time `i < m` guard per load
    // made every load wait for the one before it: 77 us per launch at N = 1e6, m = 4 - 0.4 TB/s), two rows per
    // pass of the loop are independent chains
    __shared__ double red[4][M + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double acc[M + 1];
#pragma unroll
    for (int i = 0; i <= M; ++i) acc[i] = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    auto row = [&](const double (&d)[M]) {
        double dS = 0.0, rowsum = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            dS = fma(hc.c[i], d[i], dS);
            rowsum += d[i];
        }
        dS += rowsum * delta;
        acc[0] += fast_log(dS);
        const double inv = fast_rcp(dS);
#pragma unroll
        for (int i = 0; i < M; ++i) acc[1 + i] += d[i] * inv;
    };
    int64_t n = (int64_t)blockIdx.x * blockDim.x + tid;
    for (; n + stride < N; n += 2 * stride) {
        double d0[M], d1[M];
#pragma unroll
        for (int i = 0; i < M; ++i) { d0[i] = dPsi[(int64_t)i * ldp + n]; d1[i] = dPsi[(int64_t)i * ldp + n + stride]; }
        row(d0);
        row(d1);
    }
    if (n < N) {
        double d0[M];
#pragma unroll
        for (int i = 0; i < M; ++i) d0[i] = dPsi[(int64_t)i * ldp + n];
        row(d0);
    }
#pragma unroll
    for (int i = 0; i <= M; ++i) {
        const double v = wave_sum(acc[i]);
        if (lane == 0) red[wv][i] = v;
    }
    __syncthreads();
    const int nacc = 1 + M;
    if (tid < nacc) coherent_store(partial + (int64_t)blockIdx.x * nacc + tid, (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
    if (sentinel) {                                   // self-validating rows: workgroup 0 polls them (sentinel_finish)
        __shared__ double all_s[TTM_SENT_WGS * (M + 1)];
        __shared__ double fin_s[M + 1];
        if (blockIdx.x == 0) sentinel_finish(partial, nacc, all_s, fin_s, out, sentinel == 2);
        return;
    }
    drain_stores();
    // counter != nullptr: the workgroup that draws the last ticket finishes (grids of up to 128 workgroups; the partial sums
    // travel as agent-scope stores / loads, no fence - csrc/ttm_dev.h: finish_sums); nullptr: a second launch does
    if (counter) {
        __shared__ double fin[M + 1];
        __shared__ double scr[4 * (M + 1)];
        if (finish_sums(partial, nacc, counter, scr, fin)) publish(fin, nacc, out, flag, mark);
    }
}

// THE EVALUATION SERVER of a host optimiser loop (grids of up to 128 workgroups; the filter's 37-evaluation chains): ONE
// launch for the whole loop.  A launch per evaluation costs 6.2 us of launch + completion round trip even for an empty kernel;
// a request posted by the host into a mailbox in DEVICE memory (fine-grained VRAM, written by the host through the PCIe BAR: posted
// writes, pushed out by sfence) and polled there by resident workgroups costs 2.6-3.7 us (tools/micro/mailbox.cpp, 1 / 98
// polling workgroups).  [Round 3 polled a mailbox in HOST memory: every poll a non-posted PCIe read - 27-94 us per request.]
//   box:  [0] = generation << 32 | request number r = 1, 2, ... (0xffffffff: leave), [1 ..] the M trial coefficients - written by
//         the host behind each other (coefficients, sfence, word, sfence).  The generation is the loop's (a process-wide counter):
//         a word of an OLDER generation is what the mailbox's last user left (keep waiting), of a NEWER one a later loop's - this
//         server was forgotten: leave;
//   every workgroup waits for request r (bounded: TTM_SRV_TICKS of the 100 MHz wall clock - a host that went away lets the grid
//   drain), adds its rows and stores its partial sums into region r mod 2 of the armed rows; workgroup 0 finishes the request as
//   sentinel_finish does (the same order of the sums: the same bits as a launch per evaluation) and re-arms the region; it drains
//   its stores BEHIND the results, so a region is clean before the request after next can write it.
#define TTM_SRV_QUIT (~0ull)
#define TTM_SRV_TICKS 20000000ll                      /* 0.2 s */

template <int M>
__global__ __launch_bounds__(256) void k_objective_sep_server(const double* __restrict__ dPsi, int64_t ldp, int64_t N, double delta,
                                                              double* __restrict__ partial, double* __restrict__ out,
                                                              const unsigned long long* box, unsigned int gen, int give_up) {
    __shared__ double red[4][M + 1];
    __shared__ double all_s[TTM_SENT_WGS * (M + 1)];
    __shared__ double fin_s[M + 1];
    __shared__ double s_c[M];
    __shared__ unsigned long long s_seq;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int nacc = 1 + M;
    const int region = (int)gridDim.x * nacc;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (unsigned long long r = 1;; ++r) {
        if (tid == 0) {
            unsigned long long v = 0;
            const long long t0 = wall_clock64();
            for (int spin = 0;; ++spin) {
                v = __hip_atomic_load(box, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const unsigned int vg = (unsigned int)(v >> 32), vr = (unsigned int)v;
                if (vg == gen && vr >= (unsigned int)r) { v = vr == 0xffffffffu ? TTM_SRV_QUIT : v; break; }   // (the request, or "leave")
                if (vg > gen) { v = TTM_SRV_QUIT; break; }
                if ((spin & 255) == 255 && wall_clock64() - t0 > TTM_SRV_TICKS) { v = TTM_SRV_QUIT; break; }
            }
            s_seq = v;
        }
        __syncthreads();
        if (s_seq == TTM_SRV_QUIT) return;
        if (tid < M) s_c[tid] = __hip_atomic_load((const double*)(box + 1) + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __syncthreads();
        double cc[M], acc[M + 1];
#pragma unroll
        for (int i = 0; i < M; ++i) cc[i] = s_c[i];
#pragma unroll
        for (int i = 0; i <= M; ++i) acc[i] = 0.0;
        auto row = [&](const double (&d)[M]) {
            double dS = 0.0, rowsum = 0.0;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                dS = fma(cc[i], d[i], dS);
                rowsum += d[i];
            }
            dS += rowsum * delta;
            acc[0] += fast_log(dS);
            const double inv = fast_rcp(dS);
#pragma unroll
            for (int i = 0; i < M; ++i) acc[1 + i] += d[i] * inv;
        };
        int64_t n = (int64_t)blockIdx.x * blockDim.x + tid;
        for (; n + stride < N; n += 2 * stride) {
            double d0[M], d1[M];
#pragma unroll
            for (int i = 0; i < M; ++i) { d0[i] = dPsi[(int64_t)i * ldp + n]; d1[i] = dPsi[(int64_t)i * ldp + n + stride]; }
            row(d0);
            row(d1);
        }
        if (n < N) {
            double d0[M];
#pragma unroll
            for (int i = 0; i < M; ++i) d0[i] = dPsi[(int64_t)i * ldp + n];
            row(d0);
        }
#pragma unroll
        for (int i = 0; i <= M; ++i) {
            const double v = wave_sum(acc[i]);
            if (lane == 0) red[wv][i] = v;
        }
        __syncthreads();
        double* rows = partial + (int)(r & 1ull) * region;
        if (tid < nacc) coherent_store(rows + (int64_t)blockIdx.x * nacc + tid, (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
        if (blockIdx.x == 0) {
            sentinel_finish(rows, nacc, all_s, fin_s, out, give_up != 0);
            drain_stores();                            // (results out and the region re-armed before this workgroup takes the next request)
        }
        __syncthreads();
    }
}

// The same reduction with the derivative basis RECOMPUTED from the x_k column: every monotone term of the component is
// a plain special term of x_k (LET / RET / RBF / iRBF; kinds[i], pars[5 i ..] = its constants, in coefficient order), so
// a row costs 8 bytes of HBM instead of 8 M and ~25 instructions per term (st_eval - the very code ttm_basis runs, hence
// the same bits as the cached basis).  No N x M matrix is built or kept.
template <int M>
__global__ __launch_bounds__(256) void k_objective_sep_direct(const double* __restrict__ xk, int64_t N, const int* __restrict__ kinds,
                                                              const double* __restrict__ pars, SepCoef hc, double delta,
                                                              double* __restrict__ partial, unsigned int* __restrict__ counter,
                                                              double* __restrict__ out, double* flag, double mark, int sentinel) {
    __shared__ double et[TTM_ERF_TABLE_LEN];
    __shared__ double red[4][M + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < TTM_ERF_TABLE_LEN; i += blockDim.x) et[i] = g_erf_table[i];
    __syncthreads();
    Prog g;
    g.qx = nullptr; g.qw = nullptr; g.erf_tab = et; g.Q = 0; g.family = 0; g.mono = TTM_MONO_SEPARABLE; g.rect = 0; g.delta = delta;
    cint_p kd = (cint_p)kinds;
    cdbl_p pr = (cdbl_p)pars;
    double acc[M + 1];
#pragma unroll
    for (int i = 0; i <= M; ++i) acc[i] = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    auto basis = [&](double x, double (&d)[M]) {
#pragma unroll
        for (int i = 0; i < M; ++i) {
            double v, dv;
            st_eval<false, true>(g, kd[i], x, pr + 5 * i, v, dv);
            d[i] = dv;
        }
    };
    auto row = [&](const double (&d)[M]) {
        double dS = 0.0, rowsum = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            dS = fma(hc.c[i], d[i], dS);
            rowsum += d[i];
        }
        dS += rowsum * delta;
        acc[0] += fast_log(dS);
        const double inv = fast_rcp(dS);
#pragma unroll
        for (int i = 0; i < M; ++i) acc[1 + i] += d[i] * inv;
    };
    int64_t n = (int64_t)blockIdx.x * blockDim.x + tid;
    for (; n + stride < N; n += 2 * stride) {
        const double x0 = xk[n], x1 = xk[n + stride];
        double d0[M], d1[M];
        basis(x0, d0);
        basis(x1, d1);
        row(d0);
        row(d1);
    }
    if (n < N) {
        double d0[M];
        basis(xk[n], d0);
        row(d0);
    }
#pragma unroll
    for (int i = 0; i <= M; ++i) {
        const double v = wave_sum(acc[i]);
        if (lane == 0) red[wv][i] = v;
    }
    __syncthreads();
    const int nacc = 1 + M;
    if (tid < nacc) coherent_store(partial + (int64_t)blockIdx.x * nacc + tid, (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
    if (sentinel) {                                   // (as k_objective_sep_cached: the same finish, the same bits)
        __shared__ double all_s[TTM_SENT_WGS * (M + 1)];
        __shared__ double fin_s[M + 1];
        if (blockIdx.x == 0) sentinel_finish(partial, nacc, all_s, fin_s, out, sentinel == 2);
        return;
    }
    drain_stores();
    if (counter) {
        __shared__ double fin[M + 1];
        __shared__ double scr[4 * (M + 1)];
        if (finish_sums(partial, nacc, counter, scr, fin)) publish(fin, nacc, out, flag, mark);
    }
}

// stream-ordered completion mark in (pinned host) memory: the host polls it instead of calling hipStreamSynchronize
__global__ void k_signal(double* flag, double value) { *flag = value; }

// out[i] = sum_b partial[b*nout + i] for ALL outputs in ONE workgroup (a wave per output, the summation order of
// k_reduce_partials), then the completion mark: the finishing launch of a reduction whose host waits on the mark
__global__ __launch_bounds__(1024) void k_reduce_partials_mark(const double* __restrict__ partial, int nblocks, int nout,
                                                              double* __restrict__ out, double* flag, double mark) {
    __shared__ double fin[TTM_FIN_MAX];
    const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    for (int i = threadIdx.x >> 6; i < nout; i += nw) {
        double v = 0.0;
        for (int b = lane; b < nblocks; b += 64) v += partial[(int64_t)b * nout + i];
        v = wave_sum(v);
        if (lane == 0) fin[i] = v;
    }
    publish(fin, nout, out, flag, mark);
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

// LDS: erf table | rows (m) per-thread columns
__global__ __launch_bounds__(256) void k_gram(DevProg P, int k, const double* __restrict__ X, int64_t ldx, int64_t N,
                                              int m, double* __restrict__ partial) {
    double* rows;
    CacheStore<double> cst;
    const Prog g = make_prog_lds(P, cst, rows);
    const int bd = blockDim.x, tid = threadIdx.x;
    const Comp c = comp_at(P, k, 0, nullptr, nullptr);
    const int npair = m * m;
    double acc[TTM_GRAM_MAXPAIR];
#pragma unroll
    for (int q = 0; q < TTM_GRAM_MAXPAIR; ++q) acc[q] = 0.0;
    for (int64_t n0 = (int64_t)blockIdx.x * bd; n0 < N; n0 += (int64_t)gridDim.x * bd) {
        const int64_t n = n0 + tid;
        if (n < N) {
            XSoA x{X, ldx, n};
            sample_basis(c, g, 0, x, [&](int i, double v) { rows[i * bd + tid] = v; });
            sample_basis(c, g, 1, x, [&](int i, double v) { rows[(c.n_nm + i) * bd + tid] = v; });
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
                double a = acc[q];
                for (int t = 0; t < bd; ++t) {
                    const int tt = (t + tid) & (bd - 1);    // skewed start: conflict-free LDS columns
                    a = fma(ri[tt], rj[tt], a);
                }
                acc[q] = a;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < TTM_GRAM_MAXPAIR; ++q) {
        const int pr = tid + q * bd;
        if (pr < npair) partial[(int64_t)blockIdx.x * npair + pr] = acc[q];
    }
}

// The Gram matrix G = Psi' Psi is the one dense contraction of the path: for m <= 16 basis functions it goes through the
// fp64 matrix cores (v_mfma_f64_16x16x4f64: a 16 x 16 tile of G from 4 samples per instruction, A = B' = the basis values
// themselves).  fp64 MFMA runs at the vector rate - what it saves is LDS traffic: a lane reads ONE basis value per 4-sample
// step (32 KB per 256-sample tile) where the pairwise kernel above reads two per multiply-add (496 KB per tile at m = 11,
// which made it LDS-bound).  Layout of a step: lane l holds Psi_i(sample) with i = l % 16 and sample = s + 16 (l / 16) of
// the wave's 64 samples (row stride bd + 1: the 32 lanes of an LDS pass fall on distinct banks).
typedef double ttm_v4f64 __attribute__((ext_vector_type(4)));
// WIDE = false: m <= 16, one tile.  WIDE = true: 16 < m <= 32 - three tiles (G_lo,lo | G_lo,hi | G_hi,hi; the fourth is
// the transpose of the second), a lane holds Psi_i and Psi_{16 + i} of its sample.
template <bool WIDE>
__device__ __forceinline__ void gram_mfma_body(const DevProg& P, int k, const double* __restrict__ X, int64_t ldx, int64_t N,
                                               int m, double* __restrict__ partial) {
    double* rows;
    CacheStore<double> cst;
    Prog g = make_prog_lds(P, cst, rows);
    rows = g_smem + TTM_ERF_TABLE_LEN;                     // (no column cache in this kernel: its LDS goes to more workgroups per CU)
    const int bd = blockDim.x, tid = threadIdx.x, rs = bd + 1, nw = bd >> 6;
    const Comp c = comp_at(P, k, 0, nullptr, nullptr);
    const int lane = tid & 63, wv = tid >> 6;
    const int bi = lane & 15, kq = lane >> 4;
    constexpr int NT = WIDE ? 3 : 1;
    ttm_v4f64 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { ttm_v4f64 z = {0.0, 0.0, 0.0, 0.0}; acc[t] = z; }
    for (int64_t n0 = (int64_t)blockIdx.x * bd; n0 < N; n0 += (int64_t)gridDim.x * bd) {
        const int64_t n = n0 + tid;
        if (n < N) {
            XSoA x{X, ldx, n};
            sample_basis(c, g, 0, x, [&](int i, double v) { rows[i * rs + tid] = v; });
            sample_basis(c, g, 1, x, [&](int i, double v) { rows[(c.n_nm + i) * rs + tid] = v; });
        } else {
            for (int i = 0; i < m; ++i) rows[i * rs + tid] = 0.0;
        }
        __syncthreads();
        const double* src = rows + bi * rs + wv * 64 + 16 * kq;
#pragma unroll 4
        for (int s2 = 0; s2 < 16; ++s2) {
            const double lo = bi < m ? src[s2] : 0.0;
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(lo, lo, acc[0], 0, 0, 0);
            if (WIDE) {
                const double hi = 16 + bi < m ? src[16 * rs + s2] : 0.0;
                acc[NT > 1 ? 1 : 0] = __builtin_amdgcn_mfma_f64_16x16x4f64(lo, hi, acc[NT > 1 ? 1 : 0], 0, 0, 0);
                acc[NT > 2 ? 2 : 0] = __builtin_amdgcn_mfma_f64_16x16x4f64(hi, hi, acc[NT > 2 ? 2 : 0], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // the waves' tiles: summed through LDS in wave order; element r of lane l of a tile is row l / 16 + 4 r, column l % 16
    // of that tile (measured; rows are the A operand's basis index)
    double* red = rows;
    const int mm = m * m;
    for (int t = 0; t < NT; ++t) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wv * 4 + r) * 64 + lane] = acc[t][r];
        __syncthreads();
        for (int e = tid; e < 256; e += bd) {
            const int r = e >> 6, l = e & 63;
            double v = 0.0;
            for (int w = 0; w < nw; ++w) v += red[(w * 4 + r) * 64 + l];
            int gi = (l >> 4) + 4 * r, gj = l & 15;
            if (t == 1) gj += 16;
            if (t == 2) { gi += 16; gj += 16; }
            if (gi < m && gj < m) {
                partial[(int64_t)blockIdx.x * mm + gi * m + gj] = v;
                if (t == 1) partial[(int64_t)blockIdx.x * mm + gj * m + gi] = v;
            }
        }
    }
}

template <bool WIDE>
__global__ __launch_bounds__(256) void k_gram_mfma(DevProg P, int k, const double* __restrict__ X, int64_t ldx, int64_t N,
                                                   int m, double* __restrict__ partial) {
    gram_mfma_body<WIDE>(P, k, X, ldx, N, m, partial);
}

// Gram matrices of SEVERAL components in one launch (blockIdx.y = component of the batch; the optimiser batches of the filter:
// three launches of 10 us one after the other, each followed by its reduction) and one reduction launch for all of them
#define TTM_GRAM_BATCH 8
struct GramBatch { int k[TTM_GRAM_BATCH]; int m[TTM_GRAM_BATCH]; int poff[TTM_GRAM_BATCH]; int ooff[TTM_GRAM_BATCH]; };

__global__ __launch_bounds__(256) void k_gram_mfma_many(DevProg P, GramBatch gb, const double* __restrict__ X, int64_t ldx, int64_t N,
                                                        double* __restrict__ partial) {
    const int y = blockIdx.y;
    gram_mfma_body<false>(P, gb.k[y], X, ldx, N, gb.m[y], partial + gb.poff[y]);
}

// out[ooff[y] + i] = sum_b partial[poff[y] + b m_y^2 + i]: the order of k_reduce_partials, segment by segment
__global__ __launch_bounds__(256) void k_reduce_partials_many(const double* __restrict__ partial, GramBatch gb, int nblocks,
                                                              double* __restrict__ out) {
    const int y = blockIdx.y, lane = threadIdx.x & 63;
    const int nout = gb.m[y] * gb.m[y];
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nout) return;
    const double* src = partial + gb.poff[y];
    double v = 0.0;
    for (int b = lane; b < nblocks; b += 64) v += src[(int64_t)b * nout + i];
    v = wave_sum(v);
    if (lane == 0) out[gb.ooff[y] + i] = v;
}

// ---------------------------------------------------------------------------
// Column utilities of the device-resident ensemble filter (entf.Filter: example_06.py:252-328 without a host copy of
// the ensemble): everything is column-major, one thread per row.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lorenz63(double* __restrict__ E, int64_t ld, int64_t N, double dt, int nt) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double x = E[n], y = E[ld + n], z = E[2 * ld + n];
    for (int i = 0; i < nt; ++i) lorenz63_rk4_step(x, y, z, dt);
    E[n] = x; E[ld + n] = y; E[2 * ld + n] = z;
}

// out = in + sd * noise, noise = the given column or (noise == nullptr) standard normal deviates (seed, stream, row)
__global__ __launch_bounds__(256) void k_perturb(const double* __restrict__ in, const double* __restrict__ noise, double sd,
                                                 uint64_t seed, uint32_t stream_id, int64_t row0, int64_t N, double* __restrict__ out) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    out[n] = in[n] + sd * (noise ? noise[n] : normal_deviate(seed, stream_id, (uint64_t)(row0 + n)));
}

// out[j][n] = in[src[j]][n] * scale[j] + shift[j]  (column gather + affine map; scale / shift nullable)
struct ColMap { int src[16]; double scale[16]; double shift[16]; };
__global__ __launch_bounds__(256) void k_map_columns(const double* __restrict__ in, int64_t ldi, ColMap cmap, int ncols, int64_t N,
                                                     double* __restrict__ out, int64_t ldo) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    for (int j = 0; j < ncols; ++j) {
        const double v = cmap.src[j] >= 0 ? in[(int64_t)cmap.src[j] * ldi + n] : 0.0;
        out[(int64_t)j * ldo + n] = v * cmap.scale[j] + cmap.shift[j];
    }
}

// ---------------------------------------------------------------------------
// host side: launch planning
// ---------------------------------------------------------------------------

static const int kLdsBudget = 64 * 1024;      // bytes per workgroup

// What the launch planning needs to know about the device (queried once per process) and the tuning knobs of the
// environment (read once, at the first launch - never on the launch path again).
struct DeviceInfo {
    int cus = 256;                 // compute units
    size_t lds_per_cu = 160 * 1024;   // bytes of LDS a workgroup may be granted (gfx950: 160 KB per CU)
};
static const DeviceInfo& device_info() {
    static DeviceInfo di = [] {
        DeviceInfo d;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) {
            if (prop.multiProcessorCount > 0) d.cus = prop.multiProcessorCount;
            if (prop.maxSharedMemoryPerMultiProcessor >= 64 * 1024) d.lds_per_cu = prop.maxSharedMemoryPerMultiProcessor;
        }
        (void)hipGetLastError();
        return d;
    }();
    return di;
}
// Options: what a test or a tuning run may override (ttm_set_option).  The defaults come from the environment
// (TTM_<NAME IN UPPER CASE>), which is read ONCE, when the library is first used - never on the launch path.
// -1 = "let the launch planning decide".
#define TTM_OPTIONS(X)                                                                                                   \
    X(no_plan, 0)        /* 1: generic kernels instead of the planned-cache ones                                     */ \
    X(no_uform, 0)       /* 1: direct kernels instead of the U-form ones                                             */ \
    X(u_no_hot, 0)       /* 1: U-form kernels without hot records                                                    */ \
    X(u_loader, -1)      /* 0 / 1: loader-wave forward kernels off / on whatever the ensemble size                   */ \
    X(forward_ns, -1)    /* samples per thread of the generic forward kernels (1, 2, 4)                              */ \
    X(inverse_ns, -1)    /* samples per thread of the generic table inverse (1, 2)                                   */ \
    X(u_ns, -1)          /* samples per thread of k_forward_u (1, 2, 4)                                              */ \
    X(hl_ns, -1)         /* samples per evaluating thread of k_forward_hl (2, 4)                                     */ \
    X(u_xlead, -1)       /* ring depths of the loader-wave forward kernels                                           */ \
    X(u_tlead, -1)                                                                                                       \
    X(u_wgs, -1)         /* workgroups per CU of the U-form forward kernels                                          */ \
    X(rt_off, 0)         /* 1: table inverse through the generic kernel instead of k_inverse_rt                      */ \
    X(rt_threads, -1)    /* threads per workgroup of k_inverse_rt (multiple of 64, <= 1024)                          */ \
    X(rt_ns, -1)         /* rows per thread of k_inverse_rt (2, 4)                                                   */ \
    X(rt_block, -1)      /* components per block of k_inverse_rt                                                     */ \
    X(rt_etab, -1)       /* 0: exp(-x^2/4) of the put from the series instead of the interval table                  */ \
    X(rt_band, -1)       /* 0: banded maps through the LDS column cache instead of the register shift                */ \
    X(rt_window, -1)     /* resident entries per table of k_inverse_rt: 0 whole tables, > 0 that many, -1 planned    */ \
    X(gram_mfma, -1)     /* 0: Gram matrices by the pairwise kernel instead of the matrix cores                      */ \
    X(band_fwd, -1)      /* 0: banded maps through k_forward_hl instead of the push-form kernel (csrc/ttm_band.hip)   */ \
    X(band_inv, -1)      /* 0: banded maps through k_inverse_rt instead of the push-form kernel                      */ \
    X(band_cus, -1)      /* > 0: the band kernels plan their row chunks for this many CUs (tests: several tiles per chunk) */ \
    X(band_ring, -1)     /* 0: banded table inverse through k_band_inverse (tables assembled per block) although images are at hand */ \
    X(int_dense, -1)     /* 0: integrated maps with dense B sets through the generic kernels instead of csrc/ttm_int.hip */ \
    X(int_xprog, -1)     /* 0: integrated components without their X programs (csrc/ttm_xprog.h: forward map, objective / gradient sums); \
                            2: the root searches through them as well (measured: the weights are 1 % of a bisection - no gain, 5 % slower at C2a) */ \
    X(int_wgs, -1)       /* > 0: workgroups per CU of the dense integrated kernels (default: one workgroup per tile of samples) */ \
    X(int_chunks, -1)    /* > 0: component chunks of the dense integrated forward kernel (default: planned from the ensemble size) */ \
    X(fold_fused, -1)    /* 0: ttm_fold as three launches (k_fold, k_uform, k_band_records) instead of one                */ \
    X(table_fused, -1)   /* 0: inverse tables as two launches (k_table_build, k_table_index) instead of one               */ \
    X(setup_fused, -1)   /* 0: ttm_setup_staged declines (the caller then launches ttm_fold_staged and the table kernel)           */ \
    X(select_coop, -1)   /* 0: order statistics by 17 launches (k_select_hist / k_select_pick) whatever the column length;         \
                            2: tests - the one-launch select with every wait given up at once (workgroup 0 selects by itself)    */ \
    X(colstats_one, -1)  /* 0: column moments by four launches (k_colsum / k_colfinish) whatever the shape                        */ \
    X(sep_sentinel, -1)  /* 0: the evaluations of ttm_optimize_separable with ticket and completion mark whatever the grid;       \
                            2: tests - the finishing workgroup gives up at once (the failure pattern reaches the host)            */ \
    X(sep_server, -1)    /* 0: the loops of ttm_optimize_separable launch per evaluation instead of ONE evaluation server per loop      */ \
    X(roundtrip_fused, -1) /* 0: ttm_roundtrip declines (the caller makes the forward and the inverse call); 1: the fused kernel for \
                              every shape it can run, also those it is slower for (reach of three columns, density terms)          */
struct Tuning {
#define X(name, dflt) int name = dflt;
    TTM_OPTIONS(X)
#undef X
};
static Tuning tuning_from_env() {
    Tuning u;
    auto geti = [](const char* name, int dflt) {
        char env[64] = "TTM_";
        size_t n = 4;
        for (const char* c = name; *c && n < sizeof(env) - 1; ++c) env[n++] = (char)((*c >= 'a' && *c <= 'z') ? *c - 32 : *c);
        env[n] = 0;
        const char* e = getenv(env);
        return e ? atoi(e) : dflt;
    };
#define X(name, dflt) u.name = geti(#name, dflt);
    TTM_OPTIONS(X)
#undef X
    return u;
}
static Tuning& tuning() {
    static Tuning t = tuning_from_env();
    return t;
}

static int grid_for(int64_t N, int per_block) {
    int64_t tiles = (N + per_block - 1) / per_block;
    const int64_t cap = (int64_t)device_info().cus * 8;       // up to 8 resident workgroups per CU
    if (tiles > cap) tiles = cap;
    if (tiles < 1) tiles = 1;
    return (int)tiles;
}

// grid of the dense integrated kernels: these spend ~1e3 vector instructions per sample and component, so a launch is a few
// dozen tile-times per CU - one workgroup per tile lets the dispatcher balance the CUs (a capped persistent grid ends on
// the slowest CU's whole extra tile)
// (the forward map's tile is short - one evaluation per sample: six workgroups per CU walking the tiles take 6 % less than one
// workgroup per tile at C2a, whose 7 814 workgroups each pay their launch and their first load; nothing at C5-int.  The root
// searches keep one workgroup per tile.)
static int int_grid_for(int64_t N, int per_block, int default_wgs = 0) {
    int64_t tiles = (N + per_block - 1) / per_block;
    const int wgs = tuning().int_wgs > 0 ? tuning().int_wgs : default_wgs;
    const int64_t cap = wgs > 0 ? (int64_t)device_info().cus * wgs : ((int64_t)1 << 20);
    if (tiles > cap) tiles = cap;
    if (tiles < 1) tiles = 1;
    return (int)tiles;
}

static DevProg dev_prog(const ttm_program* p) {
    DevProg P;
    P.itab = p->itab; P.ftab = p->ftab; P.fdesc = p->fdesc; P.fints = p->fints; P.dpar = p->dpar; P.qx = p->quad_x; P.qw = p->quad_w; P.off = p->d_offsets;
    P.D = p->D; P.Q = p->Q; P.family = p->family; P.mono = p->monotonicity; P.rect = p->rectifier; P.delta = p->delta;
    return P;
}

static int validate(const ttm_program* p, int k0, int k1) {
    if (!p || !p->itab || !p->ftab || !p->fdesc || !p->fints || !p->dpar || !p->d_offsets || !p->h_comp_off || !p->h_dpar_off || !p->h_coef_off ||
        !p->h_nslots || !p->h_n_nm || !p->h_fold_off || !p->h_ftab_off || !p->h_nb1 || !p->h_complex)
        return set_err(TTM_E_ARG, "ttm_program has null tables%s");
    if (k0 < 0 || k1 > p->D || k0 >= k1) return set_err(TTM_E_ARG, "component range [%s%lld,%lld) invalid", "", k0, k1);
    if (p->Q < 0 || p->Q > 4096) return set_err(TTM_E_ARG, "quadrature order %s%lld out of range", "", p->Q);
    if (p->monotonicity == TTM_MONO_INTEGRATED && (p->Q < 1 || !p->quad_x || !p->quad_w))
        return set_err(TTM_E_ARG, "integrated rectifier needs quadrature nodes%s");
    return TTM_OK;
}

// do all components of [ka,kb) take the fast path (planned-cache kernels)?
static bool all_fast(const ttm_program* p, int ka, int kb) {
    if (tuning().no_plan) return false;                                   // (option: generic kernels)
    for (int k = ka; k < kb; ++k)
        if (p->h_complex[k] & 1) return false;
    return true;
}

// per-thread scratch slots the map kernels need for components [ka,kb)
static int map_slots(const ttm_program* p, int ka, int kb) {
    int ns = 0;
    for (int k = ka; k < kb; ++k) ns = p->h_nslots[k] > ns ? p->h_nslots[k] : ns;
    return ns;
}

static size_t lds_bytes(int nslots, int bd, int extra_doubles, int ns = 1) {
    return ((size_t)TTM_ERF_TABLE_LEN + (size_t)(TTM_CACHE_SLOTS + nslots) * ns * bd + extra_doubles) * 8;
}

// LDS image of the planned-cache kernels: erf table + 2 slots per way
static size_t lds_bytes_plan(const ttm_program* p, int bd, int ns) {
    int ways = p->plan_ways;
    if (ways < 1 || ways > TTM_PLAN_WAYS) ways = TTM_PLAN_WAYS;
    return ((size_t)TTM_ERF_TABLE_LEN + (size_t)2 * ways * ns * bd) * 8;
}

// largest block size whose LDS image fits; 0 if none
static int pick_block(int nslots, int extra_doubles, int ns = 1) {
    for (int bd = 256; bd >= 64; bd >>= 1)
        if (lds_bytes(nslots, bd, extra_doubles, ns) <= (size_t)kLdsBudget) return bd;
    return 0;
}

// the one-launch moments (k_colstats_one) when they apply: 0 launched, 1 not applicable
template <bool COLS>
static int colstats_one(const double* X, int64_t ld, int64_t N, int32_t d, double* mean, double* sd, double* work, hipStream_t s) {
    if (d > TTM_CS1_DMAX || N > (int64_t)TTM_CS1_WGS * 256 * TTM_CS1_ROWS || tuning().colstats_one == 0) return 1;
    typedef void (*kern_t)(const double*, int64_t, int64_t, double*, unsigned int*, double*, double*);
    static const kern_t kerns[TTM_CS1_DMAX] = {k_colstats_one<1, COLS>, k_colstats_one<2, COLS>, k_colstats_one<3, COLS>, k_colstats_one<4, COLS>,
                                               k_colstats_one<5, COLS>, k_colstats_one<6, COLS>, k_colstats_one<7, COLS>, k_colstats_one<8, COLS>};
    unsigned int* counter = (unsigned int*)work;                     // work: [ticket (16 bytes) | rows of partial results]
    if (hipMemsetAsync(counter, 0, 4, s) != hipSuccess) return set_err(TTM_E_HIP, "ttm_colstats: hipMemsetAsync failed%s");
    const int nb = (int)((N + 256 * TTM_CS1_ROWS - 1) / (256 * TTM_CS1_ROWS));
    hipLaunchKernelGGL(kerns[d - 1], dim3(nb), dim3(256), 0, s, X, ld, N, work + 2, counter, mean, sd);
    return check_launch("k_colstats_one") == TTM_OK ? 0 : TTM_E_HIP;
}

static int select_passes(const double* col, int64_t N, const int64_t* ranks, int32_t nr, double* out, SelState* st, unsigned int* hist,
                         ttm_comm* comm, hipStream_t s) {
    if (!comm && tuning().select_coop != 0 && N <= (int64_t)TTM_SELC_WGS * 256 * TTM_SELC_ROWS) {
        // one launch: keys in registers, a grid barrier per pass (k_select_coop); the barrier words and the three histogram
        // regions sit behind the multi-launch state in `work`
        unsigned int* bar = hist + TTM_SEL_MAX * 256;                     // 32 words: arrivals, candidate-list lengths
        unsigned int* ghist = bar + 32;                                   // three regions; the first one is cleared with `bar`
        unsigned long long* cand = (unsigned long long*)(ghist + 3 * TTM_SEL_MAX * 256);
        const int nb = (int)((N + 256 * TTM_SELC_ROWS - 1) / (256 * TTM_SELC_ROWS));
        if (hipMemsetAsync(bar, 0, (32 + TTM_SEL_MAX * 256) * 4, s) != hipSuccess)
            return set_err(TTM_E_HIP, "ttm_order_statistics: hipMemsetAsync failed%s");
        hipLaunchKernelGGL(k_select_coop, dim3(nb), dim3(256), 0, s, col, N, (const long long*)ranks, (int)nr, ghist, bar, cand, out,
                           tuning().select_coop == 2 ? 1 : 0);
        return check_launch("k_select_coop");
    }
    hipLaunchKernelGGL(k_select_init, dim3(1), dim3(256), 0, s, (const long long*)ranks, (int)nr, st, hist);
    const int nb = N > 0 ? grid_for(N, 256 * 8) : 0;
    for (int shift = 56; shift >= 0; shift -= 8) {
        // (a rank whose shard is empty has nothing to count but still joins the all-reduce of the pass)
        if (nb > 0) hipLaunchKernelGGL(k_select_hist, dim3(nb), dim3(256), 0, s, col, N, (int)nr, shift, (const SelState*)st, hist);
        if (comm) {
            const int rc = ttm_allreduce_i32(comm, (int32_t*)hist, (int64_t)nr * 256, TTM_OP_SUM, (void*)s);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(k_select_pick, dim3(nr), dim3(256), 0, s, (int)nr, shift, st, hist, out);
    }
    return check_launch("k_select");
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------

extern "C" {

const char* ttm_last_error_string(void) { return g_err; }

int ttm_set_error_string(const char* text) {
    snprintf(g_err, sizeof(g_err), "%s", text ? text : "");
    return TTM_OK;
}

int ttm_version(void) { return TTM_VERSION; }

const char* ttm_last_kernel(void) { return g_last_kernel; }

int ttm_set_option(const char* name, int32_t value) {
    if (!name) return set_err(TTM_E_ARG, "ttm_set_option: null name%s");
    Tuning& t = tuning();
#define X(field, dflt) if (!strcmp(name, #field)) { t.field = (int)value; return TTM_OK; }
    TTM_OPTIONS(X)
#undef X
    return set_err(TTM_E_ARG, "ttm_set_option: unknown option '%s'", name);
}

int ttm_reset_options(void) {
    tuning() = tuning_from_env();
    return TTM_OK;
}

int64_t ttm_program_sizeof(void) { return (int64_t)sizeof(ttm_program); }

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

int64_t ttm_colstats_work_size(int64_t N, int32_t d) {
    (void)N;        // the four-launch path: TTM_STAT_BLOCKS rows of d sums; the one-launch path: a ticket + 128 rows of 1 + 2 d
    const int64_t a = (int64_t)TTM_STAT_BLOCKS * d, b = 2 + (int64_t)TTM_CS1_WGS * (1 + 2 * TTM_CS1_DMAX);
    return a > b ? a : b;
}

int ttm_colstats(const double* Xrow, int64_t N, int32_t d, double* mean, double* sd, double* work, void* stream) {
    if (!Xrow || !mean || !sd || !work || N < 1 || d < 1) return set_err(TTM_E_ARG, "ttm_colstats: bad arguments%s");
    hipStream_t s = (hipStream_t)stream;
    const int one = colstats_one<false>(Xrow, d, N, d, mean, sd, work, s);
    if (one <= 0) return one;
    int nb = (int)((N + 3) / 4 < TTM_STAT_BLOCKS ? (N + 3) / 4 : TTM_STAT_BLOCKS);
    dim3 grid(nb, (d + 63) / 64);
    hipLaunchKernelGGL(k_colsum, grid, dim3(256), 0, s, Xrow, N, (int)d, (const double*)nullptr, work);
    hipLaunchKernelGGL(k_colfinish, dim3(d), dim3(64), 0, s, work, nb, (int)d, N, mean, 0);
    hipLaunchKernelGGL(k_colsum, grid, dim3(256), 0, s, Xrow, N, (int)d, (const double*)mean, work);
    hipLaunchKernelGGL(k_colfinish, dim3(d), dim3(64), 0, s, work, nb, (int)d, N, sd, 1);
    return check_launch("k_colsum/k_colfinish");
}

int ttm_stream_synchronize(void* stream) {
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? TTM_OK : set_err(TTM_E_HIP, "hipStreamSynchronize failed%s");
}

int ttm_colstats_cols(const double* Xcols, int64_t ld, int64_t N, int32_t d, double* mean, double* sd, double* work, void* stream) {
    if (!Xcols || !mean || !sd || !work || N < 1 || d < 1 || ld < N) return set_err(TTM_E_ARG, "ttm_colstats_cols: bad arguments%s");
    const int one = colstats_one<true>(Xcols, ld, N, d, mean, sd, work, (hipStream_t)stream);
    return one == 1 ? TTM_E_UNSUPPORTED : one;
}

int ttm_standardize_cols(const double* Xcols, int64_t ld, int64_t N, int32_t d, const double* mean, const double* sd, double* Xs,
                         int64_t ldx, void* stream) {
    if (!Xcols || !mean || !sd || !Xs || N < 1 || d < 1 || ld < N || ldx < N)
        return set_err(TTM_E_ARG, "ttm_standardize_cols: bad arguments%s");
    dim3 grid((unsigned)grid_for(N, 256 * 4), (unsigned)d);
    hipLaunchKernelGGL(k_standardize_cols, grid, dim3(256), 0, (hipStream_t)stream, Xcols, ld, N, (int)d, mean, sd, Xs, ldx);
    return check_launch("k_standardize_cols");
}

int ttm_import(const double* Xrow, int64_t N, int32_t d, const double* mean, const double* sd, double* Xsoa,
               int64_t ldx, void* stream) {
    if (!Xrow || !Xsoa || N < 1 || d < 1 || ldx < N) return set_err(TTM_E_ARG, "ttm_import: bad arguments%s");
    if ((mean == nullptr) != (sd == nullptr)) return set_err(TTM_E_ARG, "ttm_import: mean and std must both be given%s");
    if (d <= 64) {
        hipLaunchKernelGGL(k_import_flat, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, (hipStream_t)stream, Xrow, N, (int)d, mean, sd, Xsoa, ldx);
        return check_launch("k_import_flat");
    }
    dim3 grid((unsigned)((N + 63) / 64), (d + 63) / 64);
    hipLaunchKernelGGL(k_import, grid, dim3(256), 0, (hipStream_t)stream, Xrow, N, (int)d, mean, sd, Xsoa, ldx);
    return check_launch("k_import");
}

int ttm_export(const double* Xsoa, int64_t ldx, int64_t N, int32_t j0, int32_t dout, const double* mean,
               const double* sd, double* Xrow, void* stream) {
    if (!Xrow || !Xsoa || N < 1 || dout < 1 || j0 < 0 || ldx < N) return set_err(TTM_E_ARG, "ttm_export: bad arguments%s");
    if ((mean == nullptr) != (sd == nullptr)) return set_err(TTM_E_ARG, "ttm_export: mean and std must both be given%s");
    if (dout <= 64) {
        hipLaunchKernelGGL(k_export_flat, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, (hipStream_t)stream, Xsoa, ldx, N, (int)j0, (int)dout, mean, sd, Xrow);
        return check_launch("k_export_flat");
    }
    dim3 grid((unsigned)((N + 63) / 64), (dout + 63) / 64);
    hipLaunchKernelGGL(k_export, grid, dim3(256), 0, (hipStream_t)stream, Xsoa, ldx, N, (int)j0, (int)dout, mean, sd, Xrow);
    return check_launch("k_export");
}

int64_t ttm_select_work_size(int32_t nr) {
    (void)nr;       // multi-launch state + histogram | 16 barrier words | three histogram regions of the one-launch select
    return (int64_t)sizeof(SelState) + (int64_t)TTM_SEL_MAX * 256 * 4 + 128 + 3 * (int64_t)TTM_SEL_MAX * 256 * 4 +
           (int64_t)TTM_SEL_MAX * 32 * 8;
}

int ttm_order_statistics(const double* col, int64_t N, const int64_t* ranks, int32_t nr, double* out, void* work,
                         void* stream) {
    if (!col || !ranks || !out || !work || N < 1 || nr < 1 || nr > TTM_SEL_MAX)
        return set_err(TTM_E_ARG, "ttm_order_statistics: bad arguments%s");
    hipStream_t s = (hipStream_t)stream;
    SelState* st = (SelState*)work;
    unsigned int* hist = (unsigned int*)((char*)work + sizeof(SelState));
    return select_passes(col, N, ranks, nr, out, st, hist, nullptr, s);
}

// The same radix select over the shards of a column that is spread over the ranks of a communicator: every pass
// all-reduces the nr x 256 bin counts (one ttm_allreduce_i32 of <= 16 KB) before the bin of the requested GLOBAL rank
// is picked - identically on every rank - so after eight passes every rank holds the exact order statistic of the
// whole column without any of its elements having moved.  N: local elements (may be 0 < N on every rank).
int ttm_order_statistics_dist(const double* col, int64_t N, const int64_t* ranks, int32_t nr, double* out, void* work,
                              ttm_comm* comm, void* stream) {
    // N = 0 is a valid (empty) shard when there is a communicator.  The bin counts travel as int32 sums: the caller keeps
    // the GLOBAL ensemble below 2^31 samples (transport_map._order_statistics checks it)
    if ((!col && N > 0) || !ranks || !out || !work || N < 0 || (N < 1 && !comm) || N >= ((int64_t)1 << 31) || nr < 1 || nr > TTM_SEL_MAX)
        return set_err(TTM_E_ARG, "ttm_order_statistics_dist: bad arguments%s");
    SelState* st = (SelState*)work;
    unsigned int* hist = (unsigned int*)((char*)work + sizeof(SelState));
    return select_passes(col, N, ranks, nr, out, st, hist, comm, (hipStream_t)stream);
}

static int64_t fold_base_size(const ttm_program* p) { return ((int64_t)p->h_fold_off[p->D] + 8 + 1) & ~(int64_t)1; }   // + read-ahead padding, even

// dynamic LDS above 64 KB has to be allowed per kernel; remember what was granted (the call is not free)
static void allow_big_lds(const void* kern, size_t bytes) {
    static thread_local const void* seen[64];
    static thread_local size_t granted[64];
    static thread_local int n = 0;
    for (int i = 0; i < n; ++i)
        if (seen[i] == kern) {
            if (granted[i] >= bytes) return;
            granted[i] = bytes;
            (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            return;
        }
    (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (n < 64) { seen[n] = kern; granted[n] = bytes; ++n; }
}

static int plan_ways_of(const ttm_program* p) {
    return (p->plan_ways < 1 || p->plan_ways > TTM_PLAN_WAYS) ? TTM_PLAN_WAYS : p->plan_ways;
}

static bool u_on(const ttm_program* p) {
    return p->u_enabled && p->ucomp && p->ugrp && p->umono && p->ugeo && p->h_ucomp && p->h_ugrp && !tuning().no_uform;
}

int64_t ttm_fold_size(const ttm_program* p) {
    if (!p || !p->h_fold_off) return -1;
    return fold_base_size(p) + (p->u_enabled ? p->u_size : 0);
}

int64_t ttm_uform_offset(const ttm_program* p) { return (p && p->h_fold_off && p->u_enabled) ? fold_base_size(p) : -1; }

int ttm_fold(const ttm_program* p, const double* coef, double* fold, void* stream) {
    int rc = validate(p, 0, p ? p->D : 0);
    if (rc) return rc;
    if (!coef || !fold) return set_err(TTM_E_ARG, "ttm_fold: bad arguments%s");
    if (p->u_enabled && p->ucomp && p->ugrp && p->umono && p->ugeo) {
        for (int k = 0; k < p->D; ++k)
            if (p->h_ucomp && p->h_ucomp[k * TTM_UC_LEN + TTM_UC_NI] > TTM_U_NI_MAX)
                return set_err(TTM_E_LIMIT, "ttm_fold: spline of component %s%lld exceeds TTM_U_NI_MAX columns", "", k);
        const UTabs T{p->ucomp, p->ugrp, p->umono, p->ugeo};
        // ONE launch, one workgroup per component: fold -> monomial groups, special-term spline, fit check -> hot record ->
        // the component's share of the push records of a banded map (option fold_fused = 0: the three launches of round 3)
        const bool fused = tuning().fold_fused != 0;
        const bool records = p->u_p_lag > 0 && p->u_h_cls > 0 && p->u_p_stride == ttm_band::record_stride(p->u_h_cls, p->u_p_lag);
        if (!fused) hipLaunchKernelGGL(k_fold, dim3(p->D), dim3(64), 0, (hipStream_t)stream, dev_prog(p), 0, 0, coef, fold);
        hipLaunchKernelGGL(k_uform, dim3(p->D), dim3(1024), 0, (hipStream_t)stream, dev_prog(p), T, (const double*)nullptr,
                           fused ? (double*)coef : (double*)nullptr, fold,
                           fold + fold_base_size(p), (int64_t)p->u_err_off, (int64_t)p->u_h_off, (int)p->u_h_cls, (int)p->u_h_ng,
                           (int64_t)p->u_p_off, (fused && records) ? (int)p->u_p_lag : 0, (int)p->u_p_stride, (double*)nullptr);
        if (!fused && p->u_p_lag > 0 && p->u_h_cls > 0) ttm_band::build_records(p, fold + fold_base_size(p), stream);       // push records of banded maps
        return check_launch(fused ? "k_uform<fold, records>" : "k_fold");
    }
    hipLaunchKernelGGL(k_fold, dim3(p->D), dim3(64), 0, (hipStream_t)stream, dev_prog(p), 0, 0, coef, fold);
    return check_launch("k_fold");
}

int ttm_fold_staged(const ttm_program* p, const double* h_coef, double* coef, double* fold, double* h_err, void* stream) {
    int rc = validate(p, 0, p ? p->D : 0);
    if (rc) return rc;
    if (!h_coef || !coef || !fold) return set_err(TTM_E_ARG, "ttm_fold_staged: bad arguments%s");
    if (!(p->u_enabled && p->ucomp && p->ugrp && p->umono && p->ugeo) || tuning().fold_fused == 0)
        return set_err(TTM_E_UNSUPPORTED, "ttm_fold_staged: the map has no U-form (copy the coefficients and call ttm_fold)%s");
    for (int k = 0; k < p->D; ++k)
        if (p->h_ucomp && p->h_ucomp[k * TTM_UC_LEN + TTM_UC_NI] > TTM_U_NI_MAX)
            return set_err(TTM_E_LIMIT, "ttm_fold: spline of component %s%lld exceeds TTM_U_NI_MAX columns", "", k);
    const UTabs T{p->ucomp, p->ugrp, p->umono, p->ugeo};
    const bool records = p->u_p_lag > 0 && p->u_h_cls > 0 && p->u_p_stride == ttm_band::record_stride(p->u_h_cls, p->u_p_lag);
    hipLaunchKernelGGL(k_uform, dim3(p->D), dim3(1024), 0, (hipStream_t)stream, dev_prog(p), T, h_coef, coef, fold,
                       fold + fold_base_size(p), (int64_t)p->u_err_off, (int64_t)p->u_h_off, (int)p->u_h_cls, (int)p->u_h_ng,
                       (int64_t)p->u_p_off, records ? (int)p->u_p_lag : 0, (int)p->u_p_stride, h_err);
    return check_launch("k_uform<staged, fold, records>");
}

int ttm_forward(const ttm_program* p, const double* coef, const double* fold, const double* Xsoa, int64_t ldx, int64_t N,
                int32_t k0, int32_t k1, double* Zsoa, int64_t ldz, double* logdet, const double* sigma, double* sumsq,
                void* stream) {
    int rc = validate(p, k0, k1);
    if (rc) return rc;
    if (!coef || !fold || !Xsoa || N < 1 || ldx < N || (Zsoa && ldz < N) || (!Zsoa && !logdet && !sumsq))
        return set_err(TTM_E_ARG, "ttm_forward: bad arguments%s");
    const int nsl = map_slots(p, k0, k1);
    // several samples per thread: the scalar (table-interpreter) work is paid once per NS*64 samples
    int NS = N >= 4 * 256 * 256 ? 2 : 1;
    const bool sep = p->monotonicity == TTM_MONO_SEPARABLE;
    if (!sep) {
        // integrated components with dense B sets evaluate their quadrature nodes in short vectors when a lane holds ONE
        // sample (ttm_eval.h, integrate_rect_dense): that amortises the scalar work better than two samples per lane
        bool dense = true;
        for (int k = k0; k < k1; ++k) dense = dense && (p->h_complex[k] & 2);
        if (dense) NS = 1;
    }
    if (tuning().forward_ns > 0) NS = tuning().forward_ns;
    if (NS != 1 && NS != 2 && NS != 4) NS = 1;
    while (NS > 1 && !pick_block(nsl, 0, NS)) NS >>= 1;
    const int bd = pick_block(nsl, 0, NS);
    if (!bd) return set_err(TTM_E_LIMIT, "ttm_forward: %s%lld scratch slots per sample do not fit the LDS budget", "", nsl);
    if (!sep && tuning().int_dense != 0 && ttm_int::usable(p, k0, k1)) {
        // every component an integrated one with a dense B set: monomial-form kernels (csrc/ttm_int.hip), one sample per lane
        const int ibd = pick_block(nsl, 0, 1);
        const char* name = nullptr;
        // a small ensemble is split over the components as well, until the launch has ~16 workgroups per CU to balance with
        const int tiles = ibd ? int_grid_for(N, ibd, 6) : 1;
        int nchunk = (int)(((int64_t)device_info().cus * 16 + tiles - 1) / tiles);
        if (tuning().int_chunks > 0) nchunk = tuning().int_chunks;
        if (nchunk > k1 - k0) nchunk = k1 - k0;
        if (nchunk < 1) nchunk = 1;
        const int chunk = (k1 - k0 + nchunk - 1) / nchunk;
        if (ibd && ttm_int::forward(p, dev_prog(p), k0, k1, coef, fold, Xsoa, ldx, N, Zsoa, ldz, logdet, sigma, sumsq, tiles, chunk, ibd,
                                    lds_bytes(nsl, ibd, 0, 1), tuning().int_xprog != 0, stream, &name) == TTM_OK)
            return check_launch(name);
    }
    if (sep && u_on(p) && all_fast(p, k0, k1) && N < ((int64_t)1 << 29)) {
        // U-form: monomial groups + special-term splines staged per component in LDS
        int tab_cap = 0;
        for (int k = k0; k < k1; ++k) {
            const int c = TTM_U_TSTRIDE * p->h_ucomp[k * TTM_UC_LEN + TTM_UC_NI];
            tab_cap = c > tab_cap ? c : tab_cap;
        }
        int ways = p->plan_ways;
        if (ways < 1 || ways > TTM_PLAN_WAYS) ways = TTM_PLAN_WAYS;
        // banded maps, large ensembles: push-form kernel with every spline resident in LDS (csrc/ttm_band.hip)
        if (tuning().band_fwd != 0 && !tuning().u_no_hot && (N >= 64 * 1024 || tuning().band_fwd == 1) && ttm_band::usable(p, k0, k1)) {
            const char* name = nullptr;
            if (ttm_band::forward(p, fold + fold_base_size(p), k0, k1, Xsoa, ldx, N, Zsoa, ldz, logdet, sigma, sumsq,
                                  tuning().band_cus > 0 ? tuning().band_cus : device_info().cus, device_info().lds_per_cu, tuning().rt_block,
                                  stream, &name) == 0)
                return check_launch(name);
        }
        // large ensembles with aligned columns: loader-wave kernel
        {
            int nimax = 0, nchmax = 0;
            for (int k = k0; k < k1; ++k) {
                const int ni = p->h_ucomp[k * TTM_UC_LEN + TTM_UC_NI];
                nimax = ni > nimax ? ni : nimax;
            }
            nchmax = (nimax * TTM_U_TSTRIDE * 8 + 1023) >> 10;
            const int tab_slot = TTM_U_TSTRIDE * nimax;                   // doubles (nI is even: 16-byte multiple)
            int xlead = 3, tlead = 2;
            const Tuning& tn = tuning();
            if (tn.u_xlead > 0) xlead = tn.u_xlead;
            if (tn.u_tlead > 0) tlead = tn.u_tlead;
            xlead = xlead < 1 ? 1 : (xlead > 4 ? 4 : xlead);
            tlead = tlead < 1 ? 1 : (tlead > 2 ? 2 : tlead);
            // samples per evaluating thread of the hot kernels: four (a wave issues at most one fp64 instruction every
            // ~8 cycles and a dependent one only after ~30, tools/micro/fp64_peak.hip: the Horner chains of four
            // samples interleave to that rate; with two the chains wait on themselves)
            int hNS = 4;
            if (tn.hl_ns > 0) hNS = tn.hl_ns == 4 ? 4 : 2;
            const bool hot = p->u_h_cls >= 1 && p->u_h_cls <= 3 && (p->u_h_ng == 2 || p->u_h_ng == 4) && !tn.u_no_hot &&
                             p->u_p_lag <= 2;                          // (lag-3 maps: hot records without cache hits, include/ttm.h)
            const int hcw = TTM_HL_FWD_CW(logdet != nullptr);            // evaluating waves per workgroup of the hot kernel
            const int rows = hot ? hcw * 64 * hNS : TTM_UL_ROWS;
            auto lds_for = [&](int xl, int tl) { return ((size_t)(xl + 1) * rows + (size_t)(tl + 1) * tab_slot + (size_t)2 * ways * rows + TTM_EXPQ_TABLE_LEN) * 8; };
            if (hot && tn.u_xlead <= 0 && tn.u_tlead <= 0) {
                // shallower rings when they buy a workgroup per CU (more independent phases per CU outweigh the look-ahead:
                // 0.155 -> 0.152 ms at C5 with five workgroups and one-step rings)
                static const int cand[4][2] = {{3, 2}, {2, 2}, {2, 1}, {1, 1}};
                size_t best = 0;
                for (int c = 0; c < 4; ++c) {
                    size_t w = device_info().lds_per_cu / lds_for(cand[c][0], cand[c][1]);
                    if (w > (size_t)(32 / (hcw + 2))) w = 32 / (hcw + 2);
                    if (w > best) { best = w; xlead = cand[c][0]; tlead = cand[c][1]; }
                }
            }
            const size_t lds_ul = lds_for(xlead, tlead);
            const bool aligned = ((uintptr_t)Xsoa % 16 == 0) && (ldx % 2 == 0) && ldx >= ((N + 1) & ~(int64_t)1) &&
                                 (!Zsoa || ((uintptr_t)Zsoa % 16 == 0 && ldz % 2 == 0)) &&
                                 (!logdet || (uintptr_t)logdet % 16 == 0) && (!sumsq || (uintptr_t)sumsq % 16 == 0) &&
                                 ((uintptr_t)(fold + fold_base_size(p)) % 16 == 0);
            bool use_ul = aligned && N >= 64 * 1024 && nchmax <= 15 && lds_ul <= device_info().lds_per_cu / 2;
            if (tn.u_loader >= 0) use_ul = aligned && nchmax <= 15 && lds_ul <= device_info().lds_per_cu && tn.u_loader != 0;
            if (use_ul && hot) {
                typedef void (*hkern_t)(const int*, const double*, int64_t, int, int, int, const double*, int64_t, int64_t, double*,
                                        int64_t, double*, const double*, double*, int, int, int, int);
                hkern_t hk;
#define TTM_HK(L, NGV, NSV) (p->u_h_cls == 1 ? k_forward_hl<L, NGV, 1, NSV> : p->u_h_cls == 2 ? k_forward_hl<L, NGV, 2, NSV> : k_forward_hl<L, NGV, 3, NSV>)
                if (hNS == 2) {
                    if (p->u_h_ng == 2) hk = logdet ? TTM_HK(true, 2, 2) : TTM_HK(false, 2, 2);
                    else hk = logdet ? TTM_HK(true, 4, 2) : TTM_HK(false, 4, 2);
                } else {
                    if (p->u_h_ng == 2) hk = logdet ? TTM_HK(true, 2, 4) : TTM_HK(false, 2, 4);
                    else hk = logdet ? TTM_HK(true, 4, 4) : TTM_HK(false, 4, 4);
                }
#undef TTM_HK
                int wgs = (int)(device_info().lds_per_cu / lds_ul);
                if (wgs > 32 / (hcw + 2)) wgs = 32 / (hcw + 2);
                if (wgs < 1) wgs = 1;
                if (tn.u_wgs > 0) wgs = tn.u_wgs;
                const int64_t tiles = (N + rows - 1) / rows;
                const int64_t grid = tiles < (int64_t)device_info().cus * wgs ? tiles : (int64_t)device_info().cus * wgs;
                allow_big_lds((const void*)hk, lds_ul);
                hipLaunchKernelGGL(hk, dim3((unsigned)grid), dim3((hcw + 2) * 64), lds_ul, (hipStream_t)stream, p->ucomp,
                                   fold + fold_base_size(p), (int64_t)p->u_h_off, (int)p->D, (int)k0, (int)k1, Xsoa, ldx, N, Zsoa, ldz,
                                   logdet, sigma, sumsq, tab_slot, xlead, tlead, ways);
                return check_launch("k_forward_hl");
            }
            if (use_ul && !hot) {
                typedef void (*lkern_t)(const int*, const int*, const double*, int, int, int, const double*, int64_t, int64_t,
                                        double*, int64_t, double*, const double*, double*, int, int, int);
                int mb = 0, ma = 0;
                for (int k = k0; k < k1; ++k) {
                    const int* uc = p->h_ucomp + k * TTM_UC_LEN;
                    for (int g = 0; g < uc[TTM_UC_N_GRP]; ++g) {
                        const int fl = p->h_ugrp[(uc[TTM_UC_GRP_OFF] + g) * TTM_UG_LEN + TTM_UG_FLAGS];
                        if ((fl & TTM_PLAN_HF) && TTM_UG_DEGB(fl) > mb) mb = TTM_UG_DEGB(fl);
                        if ((fl & TTM_UGF_POLY) && TTM_UG_DEGA(fl) > ma) ma = TTM_UG_DEGA(fl);
                    }
                }
                const int cls = (mb <= 3 && ma <= 1) ? 0 : ((mb <= 5 && ma <= 5) ? 1 : ((mb <= 7 && ma <= 7) ? 2 : 3));
                lkern_t lk = logdet ? (cls == 0 ? k_forward_ul<true, 3, 1> : cls == 1 ? k_forward_ul<true, 5, 5> : cls == 2 ? k_forward_ul<true, 7, 7> : k_forward_ul<true, 10, 10>)
                                    : (cls == 0 ? k_forward_ul<false, 3, 1> : cls == 1 ? k_forward_ul<false, 5, 5> : cls == 2 ? k_forward_ul<false, 7, 7> : k_forward_ul<false, 10, 10>);
                int wgs = (int)(device_info().lds_per_cu / lds_ul);
                if (wgs > 32 / (TTM_UL_CW + 2)) wgs = 32 / (TTM_UL_CW + 2);                                    // 6 waves per workgroup, 32 per CU
                if (wgs < 1) wgs = 1;
                if (tn.u_wgs > 0) wgs = tn.u_wgs;
                const int64_t tiles = (N + TTM_UL_ROWS - 1) / TTM_UL_ROWS;
                const int64_t grid = tiles < (int64_t)device_info().cus * wgs ? tiles : (int64_t)device_info().cus * wgs;
                allow_big_lds((const void*)lk, lds_ul);
                hipLaunchKernelGGL(lk, dim3((unsigned)grid), dim3(TTM_UL_THREADS), lds_ul, (hipStream_t)stream, p->ucomp, p->ugrp,
                                   fold + fold_base_size(p), (int)p->D, (int)k0, (int)k1, Xsoa, ldx, N, Zsoa, ldz, logdet, sigma, sumsq,
                                   tab_slot, xlead, tlead);
                return check_launch("k_forward_ul");
            }
        }
        int uNS = N >= 2 * 256 * 256 ? 2 : 1;
        if (tuning().u_ns > 0) uNS = tuning().u_ns;
        if (uNS != 1 && uNS != 2 && uNS != 4) uNS = 2;
        const int ubd = 256;
        const size_t lds = ((size_t)2 * tab_cap + (size_t)2 * ways * uNS * ubd) * 8;
        if (lds <= (size_t)kLdsBudget) {
            typedef void (*ukern_t)(const int*, const int*, const double*, int, int, int, const double*, int64_t, int64_t, double*,
                                    int64_t, double*, const double*, double*, int);
            // Horner degrees of the nonmonotone groups in this launch -> smallest fixed-degree instantiation
            int mb = 0, ma = 0;
            for (int k = k0; k < k1; ++k) {
                const int* uc = p->h_ucomp + k * TTM_UC_LEN;
                for (int g = 0; g < uc[TTM_UC_N_GRP]; ++g) {
                    const int fl = p->h_ugrp[(uc[TTM_UC_GRP_OFF] + g) * TTM_UG_LEN + TTM_UG_FLAGS];
                    if ((fl & TTM_PLAN_HF) && TTM_UG_DEGB(fl) > mb) mb = TTM_UG_DEGB(fl);
                    if ((fl & TTM_UGF_POLY) && TTM_UG_DEGA(fl) > ma) ma = TTM_UG_DEGA(fl);
                }
            }
            const int cls = (mb <= 3 && ma <= 1) ? 0 : ((mb <= 5 && ma <= 5) ? 1 : ((mb <= 7 && ma <= 7) ? 2 : 3));
#define TTM_UK(L, NSV) (cls == 0 ? k_forward_u<L, NSV, 3, 1> : cls == 1 ? k_forward_u<L, NSV, 5, 5> : cls == 2 ? k_forward_u<L, NSV, 7, 7> : k_forward_u<L, NSV, 10, 10>)
            ukern_t uk = logdet ? (uNS == 4 ? TTM_UK(true, 4) : uNS == 2 ? TTM_UK(true, 2) : TTM_UK(true, 1))
                                : (uNS == 4 ? TTM_UK(false, 4) : uNS == 2 ? TTM_UK(false, 2) : TTM_UK(false, 1));
#undef TTM_UK
            int wgs_per_cu = (int)(device_info().lds_per_cu / (lds ? lds : 1));
            if (wgs_per_cu > 8) wgs_per_cu = 8;
            if (wgs_per_cu < 1) wgs_per_cu = 1;
            if (tuning().u_wgs > 0) wgs_per_cu = tuning().u_wgs;
            int64_t tiles = (N + (int64_t)uNS * ubd - 1) / ((int64_t)uNS * ubd);
            int64_t grid = tiles < (int64_t)device_info().cus * wgs_per_cu ? tiles : (int64_t)device_info().cus * wgs_per_cu;
            hipLaunchKernelGGL(uk, dim3((unsigned)grid), dim3(ubd), lds, (hipStream_t)stream, p->ucomp, p->ugrp,
                               fold + fold_base_size(p), (int)p->D, (int)k0, (int)k1, Xsoa, ldx, N, Zsoa, ldz, logdet, sigma, sumsq, tab_cap);
            return check_launch("k_forward_u");
        }
    }
    if (all_fast(p, k0, k1)) {
        typedef void (*pkern_t)(DevProg, int, int, const double*, const double*, int64_t, int64_t, double*, int64_t, double*,
                                const double*, double*);
        pkern_t pk;
        const bool he = p->family == TTM_FAM_HERMITE_E;
#define TTM_FWDP_PICK2(M, L, NSV) (he ? k_forward_plan<M, L, NSV, TTM_FAM_HERMITE_E> : k_forward_plan<M, L, NSV, -1>)
#define TTM_FWDP_PICK(NSV)                                                                                              \
    (sep ? (logdet ? TTM_FWDP_PICK2(TTM_MONO_SEPARABLE, true, NSV) : TTM_FWDP_PICK2(TTM_MONO_SEPARABLE, false, NSV))     \
         : (logdet ? TTM_FWDP_PICK2(TTM_MONO_INTEGRATED, true, NSV) : TTM_FWDP_PICK2(TTM_MONO_INTEGRATED, false, NSV)))
        if (NS >= 2) { NS = 2; pk = TTM_FWDP_PICK(2); } else pk = TTM_FWDP_PICK(1);
#undef TTM_FWDP_PICK
#undef TTM_FWDP_PICK2
        const int pbd = 256;
        hipLaunchKernelGGL(pk, dim3(grid_for(N, NS * pbd)), dim3(pbd), lds_bytes_plan(p, pbd, NS), (hipStream_t)stream, dev_prog(p),
                           (int)k0, (int)k1, fold, Xsoa, ldx, N, Zsoa, ldz, logdet, sigma, sumsq);
        return check_launch("k_forward_plan");
    }
    typedef void (*kern_t)(DevProg, int, int, const double*, const double*, const double*, int64_t, int64_t, double*, int64_t,
                           double*, const double*, double*);
    kern_t kern;
#define TTM_FWD_PICK(NSV)                                                                                              \
    (sep ? (logdet ? k_forward<TTM_MONO_SEPARABLE, true, NSV> : k_forward<TTM_MONO_SEPARABLE, false, NSV>)              \
         : (logdet ? k_forward<TTM_MONO_INTEGRATED, true, NSV> : k_forward<TTM_MONO_INTEGRATED, false, NSV>))
    if (NS == 4) kern = TTM_FWD_PICK(4);
    else if (NS == 2) kern = TTM_FWD_PICK(2);
    else kern = TTM_FWD_PICK(1);
#undef TTM_FWD_PICK
    hipLaunchKernelGGL(kern, dim3(grid_for(N, NS * bd)), dim3(bd), lds_bytes(nsl, bd, 0, NS), (hipStream_t)stream, dev_prog(p),
                       (int)k0, (int)k1, coef, fold, Xsoa, ldx, N, Zsoa, ldz, logdet, sigma, sumsq);
    return check_launch("k_forward");
}

int ttm_basis(const ttm_program* p, int32_t k, int32_t which, const double* Xsoa, int64_t ldx, int64_t N, double* out,
              int64_t ldo, void* stream) {
    int rc = validate(p, k, k + 1);
    if (rc) return rc;
    if (!Xsoa || !out || N < 1 || ldx < N || ldo < N || which < 0 || which > 2) return set_err(TTM_E_ARG, "ttm_basis: bad arguments%s");
    hipLaunchKernelGGL(k_basis, dim3(grid_for(N, 256)), dim3(256), lds_bytes(0, 256, 0), (hipStream_t)stream, dev_prog(p), (int)k,
                       (int)which, Xsoa, ldx, N, out, ldo);
    return check_launch("k_basis");
}

int ttm_inverse_table_build(const ttm_program* p, const double* coef, const double* fold, int32_t k0, int32_t k1,
                            const double* pts, int32_t T, double* out, void* stream) {
    int rc = validate(p, k0, k1);
    if (rc) return rc;
    if (!coef || !fold || !pts || !out || T < 2) return set_err(TTM_E_ARG, "ttm_inverse_table_build: bad arguments%s");
    const int ns = map_slots(p, k0, k1);
    const int bd = pick_block(ns, 0);
    if (!bd) return set_err(TTM_E_LIMIT, "ttm_inverse_table_build: %s%lld scratch slots do not fit the LDS budget", "", ns);
    hipLaunchKernelGGL(k_table_build, dim3((T + bd - 1) / bd, k1 - k0), dim3(bd), lds_bytes(ns, bd, 0), (hipStream_t)stream,
                       dev_prog(p), (int)k0, coef, fold, pts, (int)T, out);
    return check_launch("k_table_build");
}

int ttm_inverse_table_index(const double* tab_x, int32_t ncomp, int32_t T, int32_t nb, double* tmin, double* tmax,
                            int32_t* bkt, int32_t* unsorted, void* stream) {
    if (!tab_x || !tmin || !tmax || !bkt || !unsorted || ncomp < 1 || T < 2 || T > 2048 || nb < 1 || nb > 4096)
        return set_err(TTM_E_ARG, "ttm_inverse_table_index: bad arguments%s");
    hipLaunchKernelGGL(k_table_index, dim3(ncomp), dim3(256), 0, (hipStream_t)stream, tab_x, (int)T, (int)nb, tmin, tmax, bkt,
                       unsorted);
    return check_launch("k_table_index");
}

int64_t ttm_inverse_table_image_doubles(const ttm_program* p, int32_t k0, int32_t k1, int32_t T, int32_t nb) {
    if (validate(p, k0, k1) || T < 2 || T > 2048 || nb + 1 != 1024 || !u_on(p) || tuning().band_inv == 0) return 0;
    int w0, W, slot;
    const Tuning& tn = tuning();
    if (!ttm_band::image_plan(p, k0, k1, (int)T, (int)nb, device_info().lds_per_cu, tn.rt_window, tn.rt_block, &w0, &W, &slot)) return 0;
    return slot;
}

int ttm_inverse_table_build_index(const ttm_program* p, const double* coef, const double* fold, int32_t k0, int32_t k1, const double* pts,
                                  int32_t T, int32_t nb, double* out, double* tmin, double* tmax, int32_t* bkt, int32_t* unsorted,
                                  int32_t* h_unsorted, double* img, void* stream) {
    int rc = validate(p, k0, k1);
    if (rc) return rc;
    if (!coef || !fold || !pts || !out || !tmin || !tmax || !bkt || !unsorted || T < 2 || T > 2048 || nb < 1 || nb > 4096)
        return set_err(TTM_E_ARG, "ttm_inverse_table_build_index: bad arguments%s");
    if (p->monotonicity != TTM_MONO_SEPARABLE) return set_err(TTM_E_UNSUPPORTED, "table inverse needs separable monotonicity%s");
    int iw0 = 0, iW = 0, islot = 0;
    if (img) {
        const Tuning& tn = tuning();
        if ((uintptr_t)img % 16 != 0 || ttm_inverse_table_image_doubles(p, k0, k1, T, nb) == 0 ||
            !ttm_band::image_plan(p, k0, k1, (int)T, (int)nb, device_info().lds_per_cu, tn.rt_window, tn.rt_block, &iw0, &iW, &islot))
            return set_err(TTM_E_ARG, "ttm_inverse_table_build_index: no resident-table images for this map and table geometry "
                                      "(ttm_inverse_table_image_doubles returns 0)%s");
    }
    if (tuning().table_fused == 0) {
        rc = ttm_inverse_table_build(p, coef, fold, k0, k1, pts, T, out, stream);
        if (!rc) rc = ttm_inverse_table_index(out, k1 - k0, T, nb, tmin, tmax, bkt, unsorted, stream);
        if (!rc && h_unsorted && hipMemcpyAsync(h_unsorted, unsorted, sizeof(int32_t) * (size_t)(k1 - k0), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess)
            return set_err(TTM_E_HIP, "ttm_inverse_table_build_index: copy of the flags failed%s");
        if (!rc && img) {
            hipLaunchKernelGGL(k_table_image, dim3(k1 - k0), dim3(256), 0, (hipStream_t)stream, out, (int)T, (int)nb, tmin, tmax, (const int*)bkt, img,
                               iw0, iW, islot);
            rc = check_launch("k_table_image");
        }
        return rc;
    }
    const int ns = map_slots(p, k0, k1);
    const size_t stat = 2048 * 8 + 2048 * 4 + 1024 * 4 + 64;    // (the kernel's static arrays)
    int bd = 0;
    for (int b = 256; b >= 64; b >>= 1)
        if (lds_bytes(ns, b, 0) + stat <= (size_t)kLdsBudget) { bd = b; break; }
    if (!bd) return set_err(TTM_E_LIMIT, "ttm_inverse_table_build_index: %s%lld scratch slots do not fit the LDS budget", "", ns);
    hipLaunchKernelGGL(k_table_build_index, dim3(k1 - k0), dim3(bd), lds_bytes(ns, bd, 0), (hipStream_t)stream, dev_prog(p), (int)k0, coef,
                       fold, pts, (int)T, (int)nb, out, tmin, tmax, (int*)bkt, (int*)unsorted, (int*)h_unsorted, img, iw0, iW, islot);
    return check_launch("k_table_build_index");
}

int ttm_setup_staged(const ttm_program* p, const double* h_coef, double* coef, double* fold, double* fold2, double* h_err,
                     const double* pts, int32_t T, int32_t nb, double* out, double* tmin, double* tmax, int32_t* bkt, int32_t* unsorted,
                     int32_t* h_unsorted, double* img, void* stream) {
    int rc = validate(p, 0, p ? p->D : 0);
    if (rc) return rc;
    if (!h_coef || !coef || !fold || !fold2 || fold2 == fold || !pts || !out || !tmin || !tmax || !bkt || !unsorted || T < 2 || T > 2048 ||
        nb < 1 || nb > 4096)
        return set_err(TTM_E_ARG, "ttm_setup_staged: bad arguments%s");
    const Tuning& tn = tuning();
    if (!(p->u_enabled && p->ucomp && p->ugrp && p->umono && p->ugeo) || tn.fold_fused == 0 || tn.table_fused == 0 || tn.setup_fused == 0 ||
        p->monotonicity != TTM_MONO_SEPARABLE)
        return set_err(TTM_E_UNSUPPORTED, "ttm_setup_staged: not for this map or these options (ttm_fold_staged + ttm_inverse_table_build_index)%s");
    for (int k = 0; k < p->D; ++k)
        if (p->h_ucomp && p->h_ucomp[k * TTM_UC_LEN + TTM_UC_NI] > TTM_U_NI_MAX)
            return set_err(TTM_E_LIMIT, "ttm_fold: spline of component %s%lld exceeds TTM_U_NI_MAX columns", "", k);
    int iw0 = 0, iW = 0, islot = 0;
    if (img) {
        if ((uintptr_t)img % 16 != 0 || ttm_inverse_table_image_doubles(p, 0, p->D, T, nb) == 0 ||
            !ttm_band::image_plan(p, 0, p->D, (int)T, (int)nb, device_info().lds_per_cu, tn.rt_window, tn.rt_block, &iw0, &iW, &islot))
            return set_err(TTM_E_ARG, "ttm_setup_staged: no resident-table images for this map and table geometry%s");
    }
    const int ns = map_slots(p, 0, p->D);
    const size_t dyn = lds_bytes(ns, 256, 0);
    const size_t stat = (size_t)TTM_U_NI_MAX * TTM_CHEB_N * 8 + 2 * 16 * 8 + 2048 * 8 + 2048 * 4 + 1024 * 4 + 64;     // (the kernel's static arrays)
    if (dyn + stat > device_info().lds_per_cu / 2)
        return set_err(TTM_E_UNSUPPORTED, "ttm_setup_staged: the evaluator's scratch does not fit next to the U-form arrays%s");
    static thread_local size_t granted = 0;
    if (dyn > granted) {
        (void)hipFuncSetAttribute((const void*)k_setup, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        granted = dyn;
    }
    const UTabs Tt{p->ucomp, p->ugrp, p->umono, p->ugeo};
    const bool records = p->u_p_lag > 0 && p->u_h_cls > 0 && p->u_p_stride == ttm_band::record_stride(p->u_h_cls, p->u_p_lag);
    hipLaunchKernelGGL(k_setup, dim3(2 * p->D), dim3(1024), dyn, (hipStream_t)stream, dev_prog(p), Tt, h_coef, coef, fold, fold2,
                       fold + fold_base_size(p), (int64_t)p->u_err_off, (int64_t)p->u_h_off, (int)p->u_h_cls, (int)p->u_h_ng,
                       (int64_t)p->u_p_off, records ? (int)p->u_p_lag : 0, (int)p->u_p_stride, h_err, pts, (int)T, (int)nb, out, tmin, tmax,
                       (int*)bkt, (int*)unsorted, (int*)h_unsorted, img, iw0, iW, islot);
    return check_launch("k_setup");
}

int ttm_roundtrip(const ttm_program* p, const double* coef, const double* fold, const double* Xsoa, int64_t ldx, int64_t N, double* Zsoa,
                  int64_t ldz, double* Xr, int64_t ldr, double* logdet, const double* sigma, double* sumsq, const double* tab_x, int32_t T,
                  const double* h_y_affine, const double* tmin, const double* tmax, const int32_t* bkt, int32_t nb, void* stream) {
    int rc = validate(p, 0, p ? p->D : 0);
    if (rc) return rc;
    if (!coef || !fold || !Xsoa || !Xr || !tab_x || !tmin || !tmax || !bkt || !h_y_affine || N < 1 || ldx < N || ldr < N || (Zsoa && ldz < N) ||
        T < 8 || T > 65536 || nb < 4 || nb > 65536)
        return set_err(TTM_E_ARG, "ttm_roundtrip: bad arguments%s");
    // the conditions of the two calls it stands for (ttm_forward -> k_band_few, ttm_inverse_table -> k_band_few_inverse)
    const Tuning& tn = tuning();
    if (p->monotonicity != TTM_MONO_SEPARABLE || !u_on(p) || p->u_h_cls < 1 || p->u_h_cls > 4 || !all_fast(p, 0, p->D) || tn.rt_off || tn.u_no_hot ||
        tn.band_fwd == 0 || tn.band_inv == 0 || tn.roundtrip_fused == 0 || !(N >= 64 * 1024 || (tn.band_fwd == 1 && tn.u_loader == 1)) ||
        !ttm_band::usable(p, 0, p->D))
        return TTM_E_UNSUPPORTED;
    const DeviceInfo& di = device_info();
    const char* name = nullptr;
    if (ttm_band::roundtrip(p, fold + fold_base_size(p), 0, p->D, Xsoa, ldx, N, Zsoa, ldz, Xr, ldr, logdet, sigma, sumsq, tab_x, (int)T, h_y_affine,
                            tmin, tmax, bkt, (int)nb, tn.band_cus > 0 ? tn.band_cus : di.cus, di.lds_per_cu, tn.roundtrip_fused == 1, stream,
                            &name) != 0)
        return TTM_E_UNSUPPORTED;
    return check_launch(name);
}

int ttm_inverse_table(const ttm_program* p, const double* coef, const double* fold, int32_t k0, int32_t k1, const double* Zsoa,
                      int64_t ldz, double* Xsoa, int64_t ldx, int64_t N, const double* tab_x, const double* tab_y, int64_t ldy,
                      int32_t T, const double* h_y_affine, const double* tmin, const double* tmax, const int32_t* bkt, int32_t nb,
                      int32_t truncate, const double* img, int64_t img_doubles, void* stream) {
    int rc = validate(p, k0, k1);
    if (rc) return rc;
    if (!coef || !fold || !Zsoa || !Xsoa || !tab_x || !tab_y || !tmin || !tmax || !bkt || N < 1 || ldx < N || ldz < N || T < 2 ||
        T < 8 || T > 65536 || nb < 4 || nb > 65536 || (ldy != 0 && ldy < T) || (h_y_affine && ldy != 0))
        return set_err(TTM_E_ARG, "ttm_inverse_table: bad arguments%s");
    if (p->monotonicity != TTM_MONO_SEPARABLE) return set_err(TTM_E_UNSUPPORTED, "table inverse needs separable monotonicity%s");
    // large ensembles of maps with hot records: resident-table kernel (components in blocks, tables resident in LDS)
    if (u_on(p) && p->u_h_cls >= 1 && p->u_h_cls <= 4 && (p->u_h_ng == 2 || p->u_h_ng == 4 || p->u_p_lag > 2) && all_fast(p, k0, k1) &&
        h_y_affine && ldy == 0 && T <= 4096 && nb <= 65535 && N < ((int64_t)1 << 28) && !tuning().rt_off && !tuning().u_no_hot &&
        (N >= 64 * 1024 || tuning().u_loader == 1)) {                       // (small ensembles: the table load per workgroup does not pay;
                                                                            // option u_loader = 1 forces it)
        const DeviceInfo& di = device_info();
        const Tuning& tn = tuning();
        // banded maps: push-form kernel (csrc/ttm_band.hip); clipped searches only (exp(-x^2/4) from the located interval)
        if (tn.band_inv != 0 && truncate && ttm_band::usable(p, k0, k1)) {
            const char* name = nullptr;
            if (ttm_band::inverse(p, fold + fold_base_size(p), k0, k1, Zsoa, ldz, Xsoa, ldx, N, tab_x, (int)T, h_y_affine, tmin, tmax, bkt, (int)nb,
                                  tn.band_ring != 0 ? img : nullptr, (int)img_doubles, tn.band_cus > 0 ? tn.band_cus : di.cus, di.lds_per_cu, tn.rt_window, tn.rt_block, stream, &name) == 0)
                return check_launch(name);
        }
      // k_inverse_rt sweeps the hot records themselves: not for lag-3 maps (include/ttm.h), and with fewer than four
      // components its table load per workgroup does not pay
      if (p->u_p_lag <= 2 && p->u_h_cls <= 3 && (k1 - k0 >= 4 || tn.u_loader == 1)) {      // (order class 4: the few-component kernels only)
        const int ways = plan_ways_of(p);
        const int ncomp = k1 - k0;
        int NS = tn.rt_ns == 4 ? 4 : 2;
        int W = T, w0 = 0;                                               // resident window of every table (entries [w0, w0 + W))
        int Weven = (W + 4 + 1) & ~1;
        int tab_slot = TTM_RT_HDR + Weven + (((nb + 1 + 3) / 4 + 1) & ~1);   // doubles: header + xs window + uint16 bucket index (even)
        const double ymax = fabs(h_y_affine[0]) > fabs(h_y_affine[2]) ? fabs(h_y_affine[0]) : fabs(h_y_affine[2]);
        bool etab = truncate && h_y_affine[1] > 0.0 && h_y_affine[1] * ymax * 0.5 <= 0.1;
        if (tn.rt_etab == 0) etab = false;
        const bool aligned = ((uintptr_t)Zsoa % 16 == 0) && (ldz % 2 == 0) && ldz >= ((N + 1) & ~(int64_t)1) &&
                             ((uintptr_t)Xsoa % 16 == 0) && (ldx % 2 == 0) && ldx >= ((N + 1) & ~(int64_t)1);
        // banded map: every nonmonotone group of component k reads column kc-1 or kc-2, columns consecutive -> the last
        // two columns are carried in registers (no LDS cache, four rows per thread)
        bool band = p->u_h_ng == 2 && tn.rt_band != 0;
        for (int k = k0; k < k1 && band; ++k) {
            const int* uc = p->h_ucomp + k * TTM_UC_LEN;
            if (uc[TTM_UC_KC] != p->h_ucomp[k0 * TTM_UC_LEN + TTM_UC_KC] + (k - k0) || uc[TTM_UC_N_GRP] > 2) band = false;
            for (int g = 0; g < uc[TTM_UC_N_GRP] && band; ++g) {
                const int lag = uc[TTM_UC_KC] - p->h_ugrp[(uc[TTM_UC_GRP_OFF] + g) * TTM_UG_LEN + TTM_UG_VAR];
                if (lag != 1 && lag != 2) band = false;
            }
        }
        const int wgs = 1;                                               // the tables (+ the column cache) fill the LDS of a CU
        // banded maps: four rows per thread when the chunk of a workgroup then is ONE tile - the register columns survive
        // the block boundaries and nothing is re-read (0.1755 against 0.1781 ms with two at C5; on the final kernel two rows
        // per thread are 0.7 % faster, but a tile that re-reads its columns at a block boundary takes exp(-x^2/4) from the
        // series there instead of the interval: the last bits then depend on where the boundaries fall, i.e. on the window)
        if (band && tn.rt_ns <= 0 && (N + di.cus - 1) / di.cus <= 4 * 1024) NS = 4;
        int CT = tn.rt_threads >= 64 && tn.rt_threads <= 1024 ? (tn.rt_threads & ~63) : 1024;
        const size_t budget = di.lds_per_cu / wgs;
        int Bc = 0, nblk = 0;
        size_t lds = 0;
        const int CT0 = CT;
        auto plan_blocks = [&]() {
            Bc = 0;
            for (CT = CT0; CT >= 256; CT -= 256) {                       // fewer rows in flight if the tables would not fit
                const size_t fixed = ((size_t)(etab ? 2 * Weven : 0) + (band ? 0 : (size_t)2 * ways * NS * CT)) * 8;
                if (fixed + (size_t)tab_slot * 8 > budget) continue;
                Bc = (int)((budget - fixed) / ((size_t)tab_slot * 8));
                if (Bc > ncomp) Bc = ncomp;
                if (tn.rt_block > 0 && tn.rt_block < Bc) Bc = tn.rt_block;
                nblk = (ncomp + Bc - 1) / Bc;
                Bc = (ncomp + nblk - 1) / nblk;                          // even out the blocks
                lds = fixed + (size_t)Bc * tab_slot * 8;
                if (Bc >= 4 || Bc == ncomp) break;
                Bc = 0;
            }
        };
        plan_blocks();
        // Windowed tables: with only the middle of every table resident (|y| <= 6.5 of the +-10 the grid spans; rows
        // beyond it are searched in memory, k_inverse_rt "outliers") more components fit a block.  Taken when that
        // saves a pass over the chunk - every block boundary costs the table load and a drained pipeline (~8 us at C5).
        if (tn.rt_window != 0 && aligned && T >= 64 && (tn.rt_window > 0 || (Bc > 0 && nblk > 1))) {
            const int Bfull = Bc, nfull = nblk, CTfull = CT;
            const size_t ldsfull = lds;
            W = tn.rt_window > 0 ? (tn.rt_window < 16 ? 16 : tn.rt_window) : (int)(0.65 * T);
            if (W >= T) W = T - 1;
            w0 = (T - W) / 2;
            Weven = (W + 4 + 1) & ~1;
            tab_slot = TTM_RT_HDR + Weven + (((nb + 1 + 3) / 4 + 1) & ~1);
            plan_blocks();
            if (Bc == 0 || (tn.rt_window < 0 && !(Bfull > 0 && nblk < nfull))) {            // no gain: whole tables
                W = T; w0 = 0;
                Weven = (W + 4 + 1) & ~1;
                tab_slot = TTM_RT_HDR + Weven + (((nb + 1 + 3) / 4 + 1) & ~1);
                Bc = Bfull; nblk = nfull; CT = CTfull; lds = ldsfull;
            }
        }
        if (aligned && Bc > 0) {
            typedef void (*rkern_t)(const int*, const int*, const double*, int64_t, int, int, int, const double*, int64_t, double*, int64_t,
                                    int64_t, const double*, int, double, double, double, const double*, const double*, const int*, int,
                                    int, int, int, int, int64_t, int, int);
            rkern_t rk;
#define TTM_RK4(NGV, CLSV, NSV, E) (band ? k_inverse_rt<NGV, CLSV, NSV, E, true> : k_inverse_rt<NGV, CLSV, NSV, E, false>)
#define TTM_RK3(NGV, CLSV, NSV) (etab ? TTM_RK4(NGV, CLSV, NSV, true) : TTM_RK4(NGV, CLSV, NSV, false))
#define TTM_RK(NGV, NSV) (p->u_h_cls == 1 ? TTM_RK3(NGV, 1, NSV) : p->u_h_cls == 2 ? TTM_RK3(NGV, 2, NSV) : TTM_RK3(NGV, 3, NSV))
            if (NS == 2) { if (p->u_h_ng == 2) rk = TTM_RK(2, 2); else rk = TTM_RK(4, 2); }
            else { if (p->u_h_ng == 2) rk = TTM_RK(2, 4); else rk = TTM_RK(4, 4); }
#undef TTM_RK
#undef TTM_RK3
#undef TTM_RK4
            // every workgroup gets the same (even) number of rows
            const int nwg = di.cus * wgs;
            int64_t rows = (N + nwg - 1) / nwg;
            rows = (rows + 1) & ~(int64_t)1;
            const int64_t grid = (N + rows - 1) / rows;
            allow_big_lds((const void*)rk, lds);
            hipLaunchKernelGGL(rk, dim3((unsigned)grid), dim3(CT), lds, (hipStream_t)stream, p->ucomp, p->ugrp, fold + fold_base_size(p),
                               (int64_t)p->u_h_off, (int)p->D, (int)k0, (int)k1, Zsoa, ldz, Xsoa, ldx, N, tab_x, (int)T, h_y_affine[0],
                               h_y_affine[1], h_y_affine[2], tmin, tmax, bkt, (int)nb, (int)truncate, tab_slot, Bc, ways, rows, w0, W);
            return check_launch(band ? "k_inverse_rt<band>" : "k_inverse_rt");
        }
      }
    }
    const int bd = 256;
    int NS = N >= 4 * 256 * 256 ? 2 : 1;       // two samples per thread for large ensembles (scalar work halves)
    if (tuning().inverse_ns > 0) NS = tuning().inverse_ns == 2 ? 2 : 1;
    const bool planned = all_fast(p, k0, k1);
    auto kern = NS == 2 ? k_inverse_table<2, false, -1> : k_inverse_table<1, false, -1>;
    if (planned) {
        if (p->family == TTM_FAM_HERMITE_E) kern = NS == 2 ? k_inverse_table<2, true, TTM_FAM_HERMITE_E> : k_inverse_table<1, true, TTM_FAM_HERMITE_E>;
        else kern = NS == 2 ? k_inverse_table<2, true, -1> : k_inverse_table<1, true, -1>;
    }
    hipLaunchKernelGGL(kern, dim3(grid_for(N, NS * bd)), dim3(bd), planned ? lds_bytes_plan(p, bd, NS) : lds_bytes(0, bd, 0, NS),
                       (hipStream_t)stream, dev_prog(p), (int)k0, (int)k1, coef, fold, Zsoa, ldz, Xsoa, ldx, N, tab_x, tab_y, ldy, (int)T, h_y_affine ? 1 : 0,
                       h_y_affine ? h_y_affine[0] : 0.0, h_y_affine ? h_y_affine[1] : 0.0, h_y_affine ? h_y_affine[2] : 0.0, tmin,
                       tmax, bkt, (int)nb, (int)truncate);
    return check_launch("k_inverse_table");
}

int ttm_inverse_bisect(const ttm_program* p, const double* coef, const double* fold, int32_t k0, int32_t k1, const double* Zsoa,
                       int64_t ldz, double* Xsoa, int64_t ldx, int64_t N, int32_t* iters, const int32_t* cap, void* stream) {
    int rc = validate(p, k0, k1);
    if (rc) return rc;
    if (!coef || !fold || !Zsoa || !Xsoa || !iters || N < 1 || ldx < N || ldz < N) return set_err(TTM_E_ARG, "ttm_inverse_bisect: bad arguments%s");
    const int ns = map_slots(p, k0, k1);
    const int bd = pick_block(ns, 0);
    if (!bd) return set_err(TTM_E_LIMIT, "ttm_inverse_bisect: %s%lld scratch slots do not fit the LDS budget", "", ns);
    if (tuning().int_dense != 0 && ttm_int::usable(p, k0, k1)) {
        const char* name = nullptr;
        if (ttm_int::root(p, dev_prog(p), k0, k1, coef, fold, Zsoa, ldz, Xsoa, ldx, N, iters, cap, 0, int_grid_for(N, bd), bd, lds_bytes(ns, bd, 0),
                          tuning().int_xprog == 2, stream, &name) == TTM_OK)
            return check_launch(name);
    }
    auto kern = p->monotonicity == TTM_MONO_SEPARABLE ? k_inverse_bisect<TTM_MONO_SEPARABLE, false> : k_inverse_bisect<TTM_MONO_INTEGRATED, false>;
    hipLaunchKernelGGL(kern, dim3(grid_for(N, bd)), dim3(bd), lds_bytes(ns, bd, 0), (hipStream_t)stream, dev_prog(p), (int)k0, (int)k1,
                       coef, fold, Zsoa, ldz, Xsoa, ldx, N, iters, cap);
    return check_launch("k_inverse_bisect");
}

int ttm_inverse_newton(const ttm_program* p, const double* coef, const double* fold, int32_t k0, int32_t k1, const double* Zsoa,
                       int64_t ldz, double* Xsoa, int64_t ldx, int64_t N, int32_t* iters, void* stream) {
    int rc = validate(p, k0, k1);
    if (rc) return rc;
    if (!coef || !fold || !Zsoa || !Xsoa || !iters || N < 1 || ldx < N || ldz < N) return set_err(TTM_E_ARG, "ttm_inverse_newton: bad arguments%s");
    const int ns = map_slots(p, k0, k1);
    const int bd = pick_block(ns, 0);
    if (!bd) return set_err(TTM_E_LIMIT, "ttm_inverse_newton: %s%lld scratch slots do not fit the LDS budget", "", ns);
    if (tuning().int_dense != 0 && ttm_int::usable(p, k0, k1)) {
        const char* name = nullptr;
        if (ttm_int::root(p, dev_prog(p), k0, k1, coef, fold, Zsoa, ldz, Xsoa, ldx, N, iters, nullptr, 1, int_grid_for(N, bd), bd, lds_bytes(ns, bd, 0),
                          tuning().int_xprog == 2, stream, &name) == TTM_OK)
            return check_launch(name);
    }
    auto kern = p->monotonicity == TTM_MONO_SEPARABLE ? k_inverse_bisect<TTM_MONO_SEPARABLE, true> : k_inverse_bisect<TTM_MONO_INTEGRATED, true>;
    hipLaunchKernelGGL(kern, dim3(grid_for(N, bd)), dim3(bd), lds_bytes(ns, bd, 0), (hipStream_t)stream, dev_prog(p), (int)k0, (int)k1,
                       coef, fold, Zsoa, ldz, Xsoa, ldx, N, iters, (const int*)nullptr);
    return check_launch("k_inverse_newton");
}

// workspace: [folded coefficients of the component (<= 4096) | per-block partials]
#define TTM_OBJ_FOLD_MAX 4096
// (rows of per-workgroup partial sums: nout doubles, or the TTM_X_SUM_MAX sums of an X-program evaluation, csrc/ttm_xprog.h)
int64_t ttm_reduce_work_size(int32_t nout) { return (int64_t)TTM_OBJ_FOLD_MAX + (int64_t)TTM_RED_BLOCKS * (nout > TTM_X_SUM_MAX ? nout : TTM_X_SUM_MAX); }

// objective + gradient sums of an integrated component through its X program (csrc/ttm_xprog.h, k_int_objective): ONE launch,
// coefficients as kernel arguments (h_coef_k) or from device memory (d_coef_k).  TTM_E_UNSUPPORTED: the component has none.
static int objective_xprog(const ttm_program* p, int k, const double* h_coef_k, const double* d_coef_k, int ncoef, const double* Xsoa,
                           int64_t ldx, int64_t N, double* partial, unsigned int* counter, double* out, double* flag, double mark,
                           void* stream, const char** name) {
    if (tuning().int_dense == 0 || tuning().int_xprog == 0 || ncoef > TTM_HOSTCOEF_MAX || !counter || !ttm_int::has_xprog(p, k)) return TTM_E_UNSUPPORTED;
    int bd = 256;
    while (bd > 128 && ttm_int::objective_x_lds(p, k, bd) > (size_t)kLdsBudget) bd >>= 1;       // (>= 128: a result per thread at the finish)
    if (ttm_int::objective_x_lds(p, k, bd) > (size_t)kLdsBudget) return TTM_E_UNSUPPORTED;
    int nb = grid_for(N, bd);                        // (partial: TTM_RED_BLOCKS rows of <= TTM_X_SUM_MAX sums, ttm_reduce_work_size)
    if (nb > TTM_RED_BLOCKS - 8) nb = TTM_RED_BLOCKS - 8;         // (eight more rows: the group sums of the two-stage finish)
    return ttm_int::objective_x(p, dev_prog(p), k, h_coef_k, d_coef_k, ncoef, Xsoa, ldx, N, partial, counter, out, flag, mark, nb, bd, stream, name);
}

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
    // scratch columns per thread: separable dB | integrated w, B values, integrals - a dense B set keeps its B values in the
    // weights' columns (they are dead by then): two sets instead of three
    const int nscr = (sep ? 1 : ((p->h_complex[k] & 2) ? 2 : 3)) * p->h_nb1[k];
    const int nfold = p->h_fold_off[k + 1] - p->h_fold_off[k];
    if (nfold > TTM_OBJ_FOLD_MAX) return set_err(TTM_E_LIMIT, "component %s%lld has too many folded coefficients", "", k);
    const int bd = pick_block(nscr, 4 * nacc);                  // per-thread scratch columns + one row of sums per wave
    if (!bd) return set_err(TTM_E_LIMIT, "component %s%lld does not fit the LDS budget", "", k);
    int nb = grid_for(N, bd);
    if (nb > TTM_RED_BLOCKS) nb = TTM_RED_BLOCKS;
    const DevProg P = dev_prog(p);
    double* fold_k = work;
    double* partial = work + TTM_OBJ_FOLD_MAX;
    hipLaunchKernelGGL(k_fold, dim3(1), dim3(64), 0, (hipStream_t)stream, P, (int)k, (int)k, coef_k, fold_k);
    const char* oname = "k_objective";
    if (!(!sep && tuning().int_dense != 0 && ttm_int::usable(p, k, k + 1) &&
          ttm_int::objective(p, P, k, coef_k, fold_k, Xsoa, ldx, N, nscr, nacc, partial, nullptr, nullptr, nullptr, 0.0, nb, bd,
                             lds_bytes(nscr, bd, 4 * nacc), stream, &oname) == TTM_OK))
    hipLaunchKernelGGL(k_objective, dim3(nb), dim3(bd), lds_bytes(nscr, bd, 4 * nacc), (hipStream_t)stream, P, (int)k, coef_k,
                       (const double*)fold_k, Xsoa, ldx, N, nscr, nacc, partial, (unsigned int*)nullptr, (double*)nullptr,
                       (double*)nullptr, 0.0);
    hipLaunchKernelGGL(k_reduce_partials, dim3((nacc + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const double*)partial, nb, nacc, out);
    return check_launch(oname);
}

int ttm_objective_host(const ttm_program* p, int32_t k, const double* h_coef_k, const double* Xsoa, int64_t ldx, int64_t N,
                       double* work, uint32_t* counter, double* out, void* stream) {
    return ttm_objective_host_marked(p, k, h_coef_k, Xsoa, ldx, N, work, counter, out, nullptr, 0.0, stream);
}

int ttm_objective_host_marked(const ttm_program* p, int32_t k, const double* h_coef_k, const double* Xsoa, int64_t ldx, int64_t N,
                              double* work, uint32_t* counter, double* out, double* flag, double mark, void* stream) {
    int rc = validate(p, k, k + 1);
    if (rc) return rc;
    if (!h_coef_k || !Xsoa || !work || !counter || !out || N < 1 || ldx < N) return set_err(TTM_E_ARG, "ttm_objective_host: bad arguments%s");
    if (p->monotonicity == TTM_MONO_INTEGRATED && p->rectifier != TTM_RECT_EXPONENTIAL && p->rectifier != TTM_RECT_SOFTPLUS &&
        p->rectifier != TTM_RECT_EXPNEG)
        return set_err(TTM_E_UNSUPPORTED, "rectifier has no evaluate_dfdc in the reference (TM:5119-5163)%s");
    const int sep = p->monotonicity == TTM_MONO_SEPARABLE;
    const int n_nm = p->h_n_nm[k];
    const int ncoef = p->h_coef_off[k + 1] - p->h_coef_off[k];
    const int n_mon = ncoef - n_nm;
    if (ncoef > TTM_HOSTCOEF_MAX) return set_err(TTM_E_LIMIT, "ttm_objective_host: component %s%lld has more than 128 coefficients", "", k);
    const int nacc = sep ? 1 + n_mon : 1 + n_nm + n_mon;
    // scratch columns per thread: separable dB | integrated w, B values, integrals - a dense B set keeps its B values in the
    // weights' columns (they are dead by then): two sets instead of three
    const int nscr = (sep ? 1 : ((p->h_complex[k] & 2) ? 2 : 3)) * p->h_nb1[k];
    const int nfold = p->h_fold_off[k + 1] - p->h_fold_off[k];
    if (nfold > TTM_OBJ_FOLD_MAX - TTM_HOSTCOEF_MAX) return set_err(TTM_E_LIMIT, "component %s%lld has too many folded coefficients", "", k);
    const int bd = pick_block(nscr, 4 * nacc);                  // per-thread scratch columns + one row of sums per wave
    if (!bd) return set_err(TTM_E_LIMIT, "component %s%lld does not fit the LDS budget", "", k);
    int nb = grid_for(N, bd);
    if (nb > TTM_RED_BLOCKS) nb = TTM_RED_BLOCKS;
    const DevProg P = dev_prog(p);
    double* fold_k = work;
    double* coef_dev = work + TTM_OBJ_FOLD_MAX - TTM_HOSTCOEF_MAX;
    double* partial = work + TTM_OBJ_FOLD_MAX;
    if (!sep) {
        const char* xname = nullptr;
        if (objective_xprog(p, k, h_coef_k, nullptr, ncoef, Xsoa, ldx, N, partial, (unsigned int*)counter, out, flag, mark, stream, &xname) == TTM_OK)
            return check_launch(xname);
    }
    HostCoef hc;
    for (int i = 0; i < TTM_HOSTCOEF_MAX; ++i) hc.c[i] = i < ncoef ? h_coef_k[i] : 0.0;
    hipLaunchKernelGGL(k_fold_host, dim3(1), dim3(64), 0, (hipStream_t)stream, P, (int)k, hc, ncoef, coef_dev, fold_k);
    const bool ticket = nb <= 64;                    // (see ttm_objective_sep_cached)
    const char* oname = "k_objective";
    if (!(!sep && tuning().int_dense != 0 && ttm_int::usable(p, k, k + 1) &&
          ttm_int::objective(p, P, k, coef_dev, fold_k, Xsoa, ldx, N, nscr, nacc, partial, ticket ? (unsigned int*)counter : nullptr,
                             ticket ? out : nullptr, flag, mark, nb, bd, lds_bytes(nscr, bd, 4 * nacc), stream, &oname) == TTM_OK))
    hipLaunchKernelGGL(k_objective, dim3(nb), dim3(bd), lds_bytes(nscr, bd, 4 * nacc), (hipStream_t)stream, P, (int)k,
                       (const double*)coef_dev, (const double*)fold_k, Xsoa, ldx, N, nscr, nacc, partial,
                       ticket ? (unsigned int*)counter : (unsigned int*)nullptr, ticket ? out : (double*)nullptr, flag, mark);
    if (!ticket) {
        if (flag)
            hipLaunchKernelGGL(k_reduce_partials_mark, dim3(1), dim3(64 * (nacc < 16 ? nacc : 16)), 0, (hipStream_t)stream, (const double*)partial, nb, nacc, out, flag, mark);
        else
            hipLaunchKernelGGL(k_reduce_partials, dim3((nacc + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const double*)partial, nb, nacc, out);
    }
    return check_launch(oname);
}

int ttm_objective_sep_cached(const double* dPsi, int64_t ldp, int64_t N, int32_t m, const double* h_coef_mon, double delta,
                             double* work, uint32_t* counter, double* out, void* stream) {
    return ttm_objective_sep_cached_marked(dPsi, ldp, N, m, h_coef_mon, delta, work, counter, out, nullptr, 0.0, stream);
}

// one evaluation of the reduced separable objective: cached basis (dPsi) or recomputed from the x_k column (xk, kinds, pars).
// sentinel: self-validating rows and results (<= 128 workgroups; out = page-locked host slots armed by the caller); else up to
// 128 workgroups (N <= 131 072: the filter's ensembles) ONE launch, the workgroup that draws the last ticket adds the rows of
// partial sums itself (csrc/ttm_dev.h: finish_sums, no fence), and larger grids finish in a second launch: the in-kernel
// two-stage finish was measured at + 8 us per evaluation at N = 1e6 (21.6 against 13.4 us; rounds of agent-scope loads)
static int launch_sep_objective(const double* dPsi, int64_t ldp, const double* xk, const int32_t* kinds, const double* pars, int64_t N,
                                int32_t m, const double* h_coef_mon, double delta, double* work, uint32_t* counter, double* out,
                                double* flag, double mark, bool sentinel, void* stream) {
    SepCoef hc;
    for (int i = 0; i < TTM_SEPC_MAXM; ++i) hc.c[i] = i < m ? h_coef_mon[i] : 0.0;
    int nb = grid_for(N, 256 * 4);
    if (nb > TTM_RED_BLOCKS - 8) nb = TTM_RED_BLOCKS - 8;         // (eight more rows of partial sums: the group sums of finish_sums)
    // (self-validating sums for grids of up to 128 workgroups only: at N = 10^6 - 977 rows of partial sums polled and added from
    // memory by one workgroup, the same bits as the second launch - an evaluation took 17.6 us against 13.0 us with the second
    // launch, optimize() of C5 0.0205 against 0.016 s: measured and dropped, OPTLOG round 5 item 14)
    if (sentinel && nb > TTM_SENT_WGS) return TTM_E_UNSUPPORTED;
    const bool ticket = nb <= 128;
    unsigned int* cnt = ticket && !sentinel ? (unsigned int*)counter : (unsigned int*)nullptr;
    double* partial = work + TTM_OBJ_FOLD_MAX;
    if (dPsi) {
        typedef void (*skern_t)(const double*, int64_t, int64_t, SepCoef, double, double*, unsigned int*, double*, double*, double, int);
        static const skern_t kerns[TTM_SEPC_MAXM] = {
            k_objective_sep_cached<1>, k_objective_sep_cached<2>, k_objective_sep_cached<3>, k_objective_sep_cached<4>,
            k_objective_sep_cached<5>, k_objective_sep_cached<6>, k_objective_sep_cached<7>, k_objective_sep_cached<8>,
            k_objective_sep_cached<9>, k_objective_sep_cached<10>, k_objective_sep_cached<11>, k_objective_sep_cached<12>,
            k_objective_sep_cached<13>, k_objective_sep_cached<14>, k_objective_sep_cached<15>, k_objective_sep_cached<16>};
        hipLaunchKernelGGL(kerns[m - 1], dim3(nb), dim3(256), 0, (hipStream_t)stream, dPsi, ldp, N, hc, delta, partial, cnt, out, flag, mark,
                           sentinel ? (tuning().sep_sentinel == 2 ? 2 : 1) : 0);
    } else {
        typedef void (*dkern_t)(const double*, int64_t, const int*, const double*, SepCoef, double, double*, unsigned int*, double*, double*,
                                double, int);
        static const dkern_t kerns[TTM_SEPC_MAXM] = {
            k_objective_sep_direct<1>, k_objective_sep_direct<2>, k_objective_sep_direct<3>, k_objective_sep_direct<4>,
            k_objective_sep_direct<5>, k_objective_sep_direct<6>, k_objective_sep_direct<7>, k_objective_sep_direct<8>,
            k_objective_sep_direct<9>, k_objective_sep_direct<10>, k_objective_sep_direct<11>, k_objective_sep_direct<12>,
            k_objective_sep_direct<13>, k_objective_sep_direct<14>, k_objective_sep_direct<15>, k_objective_sep_direct<16>};
        hipLaunchKernelGGL(kerns[m - 1], dim3(nb), dim3(256), 0, (hipStream_t)stream, xk, N, (const int*)kinds, pars, hc, delta, partial, cnt, out,
                           flag, mark, sentinel ? (tuning().sep_sentinel == 2 ? 2 : 1) : 0);
    }
    if (!ticket && !sentinel) {
        if (flag)
            hipLaunchKernelGGL(k_reduce_partials_mark, dim3(1), dim3(64 * (1 + (int)m < 16 ? 1 + (int)m : 16)), 0, (hipStream_t)stream,
                               (const double*)partial, nb, 1 + (int)m, out, flag, mark);
        else
            hipLaunchKernelGGL(k_reduce_partials, dim3((1 + m + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const double*)partial, nb, 1 + (int)m, out);
    }
    return check_launch(dPsi ? "k_objective_sep_cached" : "k_objective_sep_direct");
}

int ttm_objective_sep_cached_marked(const double* dPsi, int64_t ldp, int64_t N, int32_t m, const double* h_coef_mon, double delta,
                                    double* work, uint32_t* counter, double* out, double* flag, double mark, void* stream) {
    if (!dPsi || !h_coef_mon || !work || !counter || !out || N < 1 || ldp < N || m < 1)
        return set_err(TTM_E_ARG, "ttm_objective_sep_cached: bad arguments%s");
    if (m > TTM_SEPC_MAXM) return set_err(TTM_E_LIMIT, "ttm_objective_sep_cached: more than %s%lld monotone terms", "", TTM_SEPC_MAXM);
    return launch_sep_objective(dPsi, ldp, nullptr, nullptr, nullptr, N, m, h_coef_mon, delta, work, counter, out, flag, mark, false, stream);
}

int ttm_sentinel_fill(double* work, int32_t m, int64_t N, void* stream) {
    if (!work || m < 1 || m > TTM_SEPC_MAXM || N < 1) return set_err(TTM_E_ARG, "ttm_sentinel_fill: bad arguments%s");
    const int nb = grid_for(N, 256 * 4);
    if (nb > TTM_SENT_WGS || tuning().sep_sentinel == 0) return TTM_E_UNSUPPORTED;
    const int n = 2 * nb * (1 + m);                   // (two regions: the evaluation server alternates between them)
    hipLaunchKernelGGL(k_fill_bits, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)(work + TTM_OBJ_FOLD_MAX), n,
                       (unsigned long long)TTM_SENT_BITS);
    return check_launch("k_fill_bits");
}

// The evaluation server of ttm_optimize_separable (k_objective_sep_server): one launch for a whole optimiser loop; box: a mailbox
// from ttm_mailbox_acquire.  TTM_E_UNSUPPORTED: a grid of more than 128 workgroups, option sep_server = 0.
int ttm_objective_sep_server_start(const double* dPsi, int64_t ldp, int64_t N, int32_t m, double delta, double* work, double* out_host,
                                   const void* box, uint32_t gen, void* stream) {
    if (!dPsi || !work || !out_host || !box || N < 1 || ldp < N || m < 1 || m > TTM_SEPC_MAXM)
        return set_err(TTM_E_ARG, "ttm_objective_sep_server_start: bad arguments%s");
    const int nb = grid_for(N, 256 * 4);
    if (nb > TTM_SENT_WGS || tuning().sep_sentinel == 0 || tuning().sep_server == 0) return TTM_E_UNSUPPORTED;
    typedef void (*vkern_t)(const double*, int64_t, int64_t, double, double*, double*, const unsigned long long*, unsigned int, int);
    static const vkern_t kerns[TTM_SEPC_MAXM] = {
        k_objective_sep_server<1>, k_objective_sep_server<2>, k_objective_sep_server<3>, k_objective_sep_server<4>,
        k_objective_sep_server<5>, k_objective_sep_server<6>, k_objective_sep_server<7>, k_objective_sep_server<8>,
        k_objective_sep_server<9>, k_objective_sep_server<10>, k_objective_sep_server<11>, k_objective_sep_server<12>,
        k_objective_sep_server<13>, k_objective_sep_server<14>, k_objective_sep_server<15>, k_objective_sep_server<16>};
    hipLaunchKernelGGL(kerns[m - 1], dim3(nb), dim3(256), 0, (hipStream_t)stream, dPsi, ldp, N, delta, work + TTM_OBJ_FOLD_MAX, out_host,
                       (const unsigned long long*)box, (unsigned int)gen, tuning().sep_sentinel == 2 ? 1 : 0);
    return check_launch("k_objective_sep_server");
}

// Mailboxes of the evaluation servers: 256-byte slots of ONE fine-grained device allocation per process (host-writable through the
// PCIe BAR; hipExtMallocWithFlags), handed out and taken back under a lock.  NULL: none to be had (no such memory on this
// platform, or all 64 in use) - the caller launches per evaluation.
static std::mutex g_box_lock;
static unsigned char* g_box_pool = nullptr;
static unsigned long long g_box_used = 0;
static bool g_box_tried = false;
static sigjmp_buf g_box_jmp;
static void box_fault(int) { siglongjmp(g_box_jmp, 1); }
void* ttm_mailbox_acquire(void) {
    std::lock_guard<std::mutex> g(g_box_lock);
    if (!g_box_tried) {
        g_box_tried = true;
        void* p = nullptr;
        if (hipExtMallocWithFlags(&p, 64 * 256, hipDeviceMallocFinegrained) == hipSuccess && p) {
            bool ok = hipMemset(p, 0, 64 * 256) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
            if (ok) {
                // the one thing the servers need from the platform: a HOST store into this memory (a system without a large PCIe
                // BAR maps none of it).  Probed once, with the fault caught: no mailboxes then, a launch per evaluation.
                struct sigaction sa, old_segv, old_bus;
                memset(&sa, 0, sizeof(sa));
                sa.sa_handler = box_fault;
                sigemptyset(&sa.sa_mask);
                sigaction(SIGSEGV, &sa, &old_segv);
                sigaction(SIGBUS, &sa, &old_bus);
                ok = false;
                if (sigsetjmp(g_box_jmp, 1) == 0) {
                    *(volatile unsigned long long*)p = 0ull;
                    ok = true;
                }
                sigaction(SIGSEGV, &old_segv, nullptr);
                sigaction(SIGBUS, &old_bus, nullptr);
            }
            if (ok) g_box_pool = (unsigned char*)p;
            else (void)hipFree(p);
        }
        (void)hipGetLastError();
    }
    if (!g_box_pool) return nullptr;
    for (int i = 0; i < 64; ++i)
        if (!(g_box_used >> i & 1ull)) { g_box_used |= 1ull << i; return g_box_pool + 256 * i; }
    return nullptr;
}
void ttm_mailbox_release(void* box) {
    if (!box) return;
    std::lock_guard<std::mutex> g(g_box_lock);
    const long i = ((unsigned char*)box - g_box_pool) / 256;
    if (g_box_pool && i >= 0 && i < 64) g_box_used &= ~(1ull << i);
}

int ttm_objective_sep_cached_sent(const double* dPsi, int64_t ldp, int64_t N, int32_t m, const double* h_coef_mon, double delta,
                                  double* work, double* out_host, void* stream) {
    if (!dPsi || !h_coef_mon || !work || !out_host || N < 1 || ldp < N || m < 1 || m > TTM_SEPC_MAXM)
        return set_err(TTM_E_ARG, "ttm_objective_sep_cached_sent: bad arguments%s");
    return launch_sep_objective(dPsi, ldp, nullptr, nullptr, nullptr, N, m, h_coef_mon, delta, work, nullptr, out_host, nullptr, 0.0, true, stream);
}

int ttm_objective_sep_direct_sent(const double* xk, int64_t N, int32_t m, const int32_t* kinds, const double* pars,
                                  const double* h_coef_mon, double delta, double* work, double* out_host, void* stream) {
    if (!xk || !kinds || !pars || !h_coef_mon || !work || !out_host || N < 1 || m < 1 || m > TTM_SEPC_MAXM)
        return set_err(TTM_E_ARG, "ttm_objective_sep_direct_sent: bad arguments%s");
    return launch_sep_objective(nullptr, 0, xk, kinds, pars, N, m, h_coef_mon, delta, work, nullptr, out_host, nullptr, 0.0, true, stream);
}

int ttm_objective_sep_direct_marked(const double* xk, int64_t N, int32_t m, const int32_t* kinds, const double* pars,
                                    const double* h_coef_mon, double delta, double* work, uint32_t* counter, double* out,
                                    double* flag, double mark, void* stream) {
    if (!xk || !kinds || !pars || !h_coef_mon || !work || !counter || !out || N < 1 || m < 1)
        return set_err(TTM_E_ARG, "ttm_objective_sep_direct_marked: bad arguments%s");
    if (m > TTM_SEPC_MAXM) return set_err(TTM_E_LIMIT, "ttm_objective_sep_direct_marked: more than %s%lld monotone terms", "", TTM_SEPC_MAXM);
    return launch_sep_objective(nullptr, 0, xk, kinds, pars, N, m, h_coef_mon, delta, work, counter, out, flag, mark, false, stream);
}

int ttm_gram(const ttm_program* p, int32_t k, const double* Xsoa, int64_t ldx, int64_t N, double* work, double* out,
             void* stream) {
    int rc = validate(p, k, k + 1);
    if (rc) return rc;
    if (!Xsoa || !work || !out || N < 1 || ldx < N) return set_err(TTM_E_ARG, "ttm_gram: bad arguments%s");
    const int m = p->h_coef_off[k + 1] - p->h_coef_off[k];
    int bd = pick_block(m, 0);
    if (bd && m * m > TTM_GRAM_MAXPAIR * bd) bd = 0;
    if (!bd) return set_err(TTM_E_LIMIT, "ttm_gram: %s%lld basis functions exceed the kernel limits", "", m);
    int nb = grid_for(N, bd);
    if (nb > TTM_RED_BLOCKS) nb = TTM_RED_BLOCKS;
    double* partial = work + TTM_OBJ_FOLD_MAX;
    // rows of bd + 1 doubles; the final sum of the waves' tiles reuses them (1024 doubles)
    const size_t glds = ((size_t)TTM_ERF_TABLE_LEN + (m * (bd + 1) >= 1024 ? (size_t)m * (bd + 1) : 1024)) * 8;
    if (m <= 32 && bd >= 64 && tuning().gram_mfma != 0 && glds <= (size_t)kLdsBudget) {
        hipLaunchKernelGGL(m <= 16 ? k_gram_mfma<false> : k_gram_mfma<true>, dim3(nb), dim3(bd), glds, (hipStream_t)stream,
                           dev_prog(p), (int)k, Xsoa, ldx, N, m, partial);
        hipLaunchKernelGGL(k_reduce_partials, dim3((m * m + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const double*)partial, nb, m * m, out);
        return check_launch("k_gram_mfma");
    }
    hipLaunchKernelGGL(k_gram, dim3(nb), dim3(bd), lds_bytes(m, bd, 0), (hipStream_t)stream, dev_prog(p), (int)k, Xsoa, ldx, N, m, partial);
    hipLaunchKernelGGL(k_reduce_partials, dim3((m * m + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const double*)partial, nb, m * m, out);
    return check_launch("k_gram");
}

int ttm_gram_many(const ttm_program* p, const int32_t* ks, int32_t nk, const double* Xsoa, int64_t ldx, int64_t N, double* work,
                  double* out, void* stream) {
    if (!ks || nk < 1 || !Xsoa || !work || !out || N < 1 || ldx < N) return set_err(TTM_E_ARG, "ttm_gram_many: bad arguments%s");
    if (nk > TTM_GRAM_BATCH || tuning().gram_mfma == 0) return TTM_E_UNSUPPORTED;
    GramBatch gb;
    int bd = 256, mmax = 0;
    for (int y = 0; y < nk; ++y) {
        const int rc = validate(p, ks[y], ks[y] + 1);
        if (rc) return rc;
        const int m = p->h_coef_off[ks[y] + 1] - p->h_coef_off[ks[y]];
        int b = pick_block(m, 0);
        if (b && m * m > TTM_GRAM_MAXPAIR * b) b = 0;
        if (!b || b < 64 || m > 16) return TTM_E_UNSUPPORTED;            // (the caller takes ttm_gram component by component)
        if (y > 0 && b != bd) return TTM_E_UNSUPPORTED;                    // (one block size for the launch: the same tiles, hence bits)
        bd = b;
        mmax = m > mmax ? m : mmax;
        gb.k[y] = ks[y];
        gb.m[y] = m;
    }
    int nb = grid_for(N, bd);
    if (nb > TTM_RED_BLOCKS) nb = TTM_RED_BLOCKS;
    int64_t poff = 0, ooff = 0;
    for (int y = 0; y < nk; ++y) {
        gb.poff[y] = (int)poff;
        gb.ooff[y] = (int)ooff;
        poff += (int64_t)nb * gb.m[y] * gb.m[y];
        ooff += gb.m[y] * gb.m[y];
    }
    for (int y = nk; y < TTM_GRAM_BATCH; ++y) gb.k[y] = gb.m[y] = gb.poff[y] = gb.ooff[y] = 0;
    if (poff > ttm_reduce_work_size(mmax * mmax) - TTM_OBJ_FOLD_MAX) return TTM_E_UNSUPPORTED;
    const size_t glds = ((size_t)TTM_ERF_TABLE_LEN + (mmax * (bd + 1) >= 1024 ? (size_t)mmax * (bd + 1) : 1024)) * 8;
    if (glds > (size_t)kLdsBudget) return TTM_E_UNSUPPORTED;
    double* partial = work + TTM_OBJ_FOLD_MAX;
    hipLaunchKernelGGL(k_gram_mfma_many, dim3(nb, nk), dim3(bd), glds, (hipStream_t)stream, dev_prog(p), gb, Xsoa, ldx, N, partial);
    hipLaunchKernelGGL(k_reduce_partials_many, dim3((mmax * mmax + 3) / 4, nk), dim3(256), 0, (hipStream_t)stream, (const double*)partial, gb, nb,
                       out);
    return check_launch("k_gram_mfma_many");
}

int ttm_lorenz63_rk4(double* E, int64_t ld, int64_t N, double dt, int32_t nt, void* stream) {
    if (!E || N < 1 || ld < N || nt < 0) return set_err(TTM_E_ARG, "ttm_lorenz63_rk4: bad arguments%s");
    hipLaunchKernelGGL(k_lorenz63, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, E, ld, N, dt, (int)nt);
    return check_launch("k_lorenz63");
}

int ttm_perturb(const double* in, const double* noise, double sd, uint64_t seed, uint32_t stream_id, int64_t row0, int64_t N,
                double* out, void* stream) {
    if (!in || !out || N < 1) return set_err(TTM_E_ARG, "ttm_perturb: bad arguments%s");
    hipLaunchKernelGGL(k_perturb, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, noise, sd, seed, stream_id,
                       row0, N, out);
    return check_launch("k_perturb");
}

int ttm_map_columns(const double* in, int64_t ldi, const int32_t* src, const double* scale, const double* shift, int32_t ncols,
                    int64_t N, double* out, int64_t ldo, void* stream) {
    if (!out || !src || N < 1 || ncols < 1 || ncols > 16 || ldo < N || (in && ldi < N)) return set_err(TTM_E_ARG, "ttm_map_columns: bad arguments%s");
    ColMap cm;
    for (int j = 0; j < 16; ++j) {
        cm.src[j] = j < ncols ? (int)src[j] : -1;
        cm.scale[j] = (j < ncols && scale) ? scale[j] : 1.0;
        cm.shift[j] = (j < ncols && shift) ? shift[j] : 0.0;
        if (j < ncols && src[j] >= 0 && !in) return set_err(TTM_E_ARG, "ttm_map_columns: source column without an input matrix%s");
    }
    hipLaunchKernelGGL(k_map_columns, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, ldi, cm, (int)ncols, N, out, ldo);
    return check_launch("k_map_columns");
}

int ttm_signal(double* flag, double value, void* stream) {
    if (!flag) return set_err(TTM_E_ARG, "ttm_signal: null flag%s");
    hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, value);
    return check_launch("k_signal");
}

}  // extern "C"
