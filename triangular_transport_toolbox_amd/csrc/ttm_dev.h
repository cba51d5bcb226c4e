// ttm_dev.h - device-side helpers shared by the translation units of libttm.so that run the generic per-sample
// evaluators of ttm_eval.h (csrc/ttm_kernels.hip, csrc/ttm_int.hip): the by-value program argument, sample accessors,
// per-thread LDS scratch, the wave accumulators of the reductions, the LDS image with the erf table, and the
// "last workgroup finishes" protocol of the single-launch reductions.
#pragma once

#include <hip/hip_runtime.h>

#include "ttm_eval.h"

using namespace ttm;
struct DevProg {             // by-value kernel argument (all pointers are global memory)
    const int* itab;
    const int* ftab;
    const int* fdesc;
    const int* fints;
    const double* dpar;
    const double* qx;
    const double* qw;
    const int* off;          // 5 x (D+1): comp_off | dpar_off | coef_off | fold_off | ftab_off
    int D;
    int Q, family, mono, rect;
    double delta;
};

struct LdsSlots {
    double* base;
    int stride;
    __device__ __forceinline__ double get(int i) const { return base[i * stride]; }
    __device__ __forceinline__ void set(int i, double v) { base[i * stride] = v; }
};

template <int NS> struct real_of { typedef VecD<NS> type; };
template <> struct real_of<1> { typedef double type; };

// per-thread scratch slots holding NS samples each
template <class R>
struct LdsSlotsN {
    double* base;
    int stride;
    __device__ __forceinline__ R get(int i) const {
        R r;
#pragma unroll
        for (int e = 0; e < lanes_of<R>::value; ++e) set_elem(r, e, base[(i * lanes_of<R>::value + e) * stride]);
        return r;
    }
    __device__ __forceinline__ void set(int i, const R& v) {
#pragma unroll
        for (int e = 0; e < lanes_of<R>::value; ++e) base[(i * lanes_of<R>::value + e) * stride] = elem(v, e);
    }
};

// NS samples of one thread: sample e is row n0 + e*step (rows beyond N are clamped to N-1 for loads)
template <int NS>
struct XSoAN {
    typedef typename real_of<NS>::type R;
    const double* X;
    int64_t ld;
    int64_t n[NS];
    __device__ __forceinline__ R operator()(int var) const {
        R r;
#pragma unroll
        for (int e = 0; e < NS; ++e) set_elem(r, e, X[(int64_t)var * ld + n[e]]);
        return r;
    }
};

__device__ __forceinline__ double wave_sum(double v);
__device__ __forceinline__ double wave_sum_last(double v);
// accumulators of a WAVE: every add sums its argument over the lanes at once (all 64 lanes call, lanes without a sample
// contribute zero) and lane 0 keeps the running sums in one LDS row per wave.  Per-thread accumulator columns would
// be nacc x blockDim doubles of LDS - 22 of the 48 doubles per thread that capped the integrated objective kernel at
// three workgroups (1.5 waves per SIMD) per CU.
struct WaveAcc {
    double* row;
    bool active, first;          // first: the lane that keeps the sums - lane 63, where wave_sum_last leaves the total
    __device__ __forceinline__ void add(int i, double v) {
        const double s = wave_sum_last(active ? v : 0.0);
        if (first) row[i] += s;
    }
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
    __device__ __forceinline__ double get(int var) const { return var == kc ? t : 0.0; }
    __device__ __forceinline__ void get_e(int var, double& xv, double& e) const { xv = get(var); e = fast_exp(-0.25 * (xv * xv)); }
};

static __device__ const double g_erf_table[TTM_ERF_TABLE_LEN] = { TTM_ERF_TABLE_VALUES };
static __device__ const double g_expq_table[TTM_EXPQ_TABLE_LEN] = { TTM_EXPQ_TABLE_VALUES };

extern __shared__ __align__(16) double g_smem[];

#define TTM_CACHE_SLOTS 8     // per-thread column cache (VarCache): 4 x + 4 exp(-x^2/4)

// LDS image: [erf table | column cache (8 x NS x blockDim) | per-thread slots ...]
// stage_erf = false: the launch evaluates no special term (the erf table is neither loaded nor read; its LDS stays reserved so
// that the image layout - and lds_bytes() on the host - is the same)
// bd: the threads that evaluate (the first bd of the workgroup; the layout is that of a workgroup of bd threads)
template <class R>
__device__ __forceinline__ Prog make_prog_lds_n(const DevProg& P, CacheStore<R>& cache, double*& slots, int bd, bool stage_erf = true) {
    double* et = g_smem;
    if (stage_erf) {
        for (int i = threadIdx.x; i < TTM_ERF_TABLE_LEN; i += blockDim.x) et[i] = g_erf_table[i];
        __syncthreads();
    }
    cache.base = et + TTM_ERF_TABLE_LEN + threadIdx.x;
    cache.stride = bd;
    slots = et + TTM_ERF_TABLE_LEN + (size_t)TTM_CACHE_SLOTS * lanes_of<R>::value * bd;
    Prog g;
    g.qx = (cdbl_p)P.qx;
    g.qw = (cdbl_p)P.qw;
    g.erf_tab = et;
    g.Q = P.Q;
    g.family = P.family;
    g.mono = P.mono;
    g.rect = P.rect;
    g.delta = P.delta;
    return g;
}

template <class R>
__device__ __forceinline__ Prog make_prog_lds(const DevProg& P, CacheStore<R>& cache, double*& slots, bool stage_erf = true) {
    return make_prog_lds_n(P, cache, slots, (int)blockDim.x, stage_erf);
}

// component k of the program; coefficient / folded arrays are given relative to component kbase
__device__ __forceinline__ Comp comp_at(const DevProg& P, int k, int kbase, const double* coef, const double* fold) {
    cint_p off = (cint_p)P.off;
    const int D1 = P.D + 1;
    cint_p cb = (cint_p)P.itab + off[k];
    cdbl_p dp = (cdbl_p)P.dpar + off[D1 + k];
    cdbl_p cf = coef ? (cdbl_p)coef + (off[2 * D1 + k] - off[2 * D1 + kbase]) : (cdbl_p)P.dpar;
    cdbl_p fo = fold ? (cdbl_p)fold + (off[3 * D1 + k] - off[3 * D1 + kbase]) : (cdbl_p)P.dpar;
    return make_comp(cb, dp, cf, fo);
}

// Sum over the 64 lanes by data-parallel primitives (DPP): the operand of each add comes straight from another lane's
// register - row_shr 1, 2, 4, 8 inside the rows of 16 lanes, then row_bcast 15 / 31 across them - 6 x (two 32-bit DPP
// moves + v_add_f64) and no LDS round trip (__shfl_down is a ds_bpermute per half and step: 12 LDS instructions and six
// dependent waits per sum; the objective kernels take one sum per coefficient and sample).  The total is valid in LANE 63
// only; fixed order, so run-to-run deterministic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_from(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_last(double v) {
    v += dpp_from<0x111, 0xf>(v);        // row_shr:1
    v += dpp_from<0x112, 0xf>(v);        // row_shr:2
    v += dpp_from<0x114, 0xf>(v);        // row_shr:4
    v += dpp_from<0x118, 0xf>(v);        // row_shr:8   -> lane 15 of every row holds the row's sum
    v += dpp_from<0x142, 0xa>(v);        // row_bcast:15 into rows 1 and 3
    v += dpp_from<0x143, 0xc>(v);        // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}


#define TTM_RED_BLOCKS 1024
#define TTM_HOSTCOEF_MAX 128

// "Which workgroup is the last one?" without a thousand atomics on one address (they serialise in the L2: ~75 ns each,
// 77 us for the 977 workgroups of an N = 1e6 reduction): workgroups draw a ticket from one of 8 group counters
// (counter[1 + (blockIdx & 7)], different addresses proceed in parallel), the last of a group draws one from
// counter[0], and the last of those is the last workgroup of the grid.  Every workgroup has made its partial sums
// visible (__threadfence) before it draws.  counter: 16 uint32, zero before the first launch; left zero.
__device__ __forceinline__ bool last_workgroup(unsigned int* counter) {
    __shared__ int is_last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int g = blockIdx.x & 7u, ngroups = gridDim.x < 8u ? gridDim.x : 8u;
        const unsigned int in_group = (gridDim.x - g + 7u) / 8u;
        int last = 0;
        if (atomicAdd(counter + 1 + g, 1u) == in_group - 1u) {
            __threadfence();
            last = atomicAdd(counter, 1u) == ngroups - 1u ? 1 : 0;
        }
        is_last = last;
    }
    __syncthreads();
    if (is_last) {
        __threadfence();
        if (threadIdx.x < 9) counter[threadIdx.x] = 0u;
    }
    return is_last != 0;
}

// Two-stage finish of a reduction over MANY workgroups and sums (the X-program objective kernel: up to 1 016 workgroups x 192
// sums - one workgroup adding 1 016 rows by itself spends 60-80 us in dependent round trips, measured).  Workgroup b has written
// its nsum partial sums to row b of `partial` (coherent_store).  Stage A: the last workgroup of group g = b mod 8 to arrive
// (ticket on counter[1 + g]) adds the rows of its group into row nb + g - the eight groups finish side by side, all but the
// last one while other workgroups still compute.  Stage B: the last of those (ticket on counter[0]) adds the eight group rows
// into `fin` (LDS, nsum doubles) and returns true.  Rows are added lane = sum index (coalesced), wave = every nw-th row, four
// loads in flight per lane; fixed order: run-to-run deterministic.  `scr`: LDS scratch of nw x nsum doubles.
// Visibility WITHOUT fences (a device-scope fence is an L2 write-back per wave: 1 000 workgroups x 4 waves of them cost this
// kernel 45-55 us, measured): the rows travel as agent-scope atomic stores / loads - written through to / read from the level the
// XCDs share -, every wave drains its stores (vmcnt(0)) before the workgroup draws its ticket.
// partial must hold nb + 8 rows; counter: 16 uint32, zero before the first launch; left zero.
__device__ __forceinline__ void coherent_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double coherent_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain_stores() {
#if !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
#error "drain_stores(): the raw s_waitcnt immediate below is the gfx9 encoding"
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");           // (compiler ordering)
    __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0)
}

__device__ __forceinline__ void add_rows(const double* partial, int first, int step, int count, int nsum, double* scr,
                                         double* dst_lds, double* dst_glob) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    constexpr int B = 16;                                   // loads in flight per lane (a round trip to the shared level is ~1.5 us)
    for (int i0 = 0; i0 < nsum; i0 += 64) {
        const int i = i0 + lane;
        double a0 = 0.0, a1 = 0.0;
        if (i < nsum) {
            for (int r0 = wv; r0 < count; r0 += B * nw) {
                double v[B];
#pragma unroll
                for (int j = 0; j < B; ++j) {
                    const int r = r0 + j * nw;
                    v[j] = r < count ? coherent_load(partial + (int64_t)(first + r * step) * nsum + i) : 0.0;
                }
#pragma unroll
                for (int j = 0; j < B; j += 2) { a0 += v[j]; a1 += v[j + 1]; }
            }
            scr[wv * nsum + i] = a0 + a1;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nsum; i += blockDim.x) {
        double v = 0.0;
        for (int w = 0; w < nw; ++w) v += scr[w * nsum + i];
        if (dst_lds) dst_lds[i] = v;
        if (dst_glob) coherent_store(dst_glob + i, v);
    }
    drain_stores();
    __syncthreads();
}

// (every wave of the workgroup has drained its coherent_store's of row blockIdx.x before the call; flag: an int in LDS)
// stage A: false = this workgroup is done; true = it has added the rows of its group, drawn its second ticket (verdict in
// *flag) and goes on to finish_stage_b
__device__ __forceinline__ bool finish_stage_a(double* partial, int nsum, unsigned int* counter, double* scr, int* flag) {
    const unsigned int nb = gridDim.x, g = blockIdx.x & 7u;
    const unsigned int in_group = (nb - g + 7u) / 8u;
    __syncthreads();
    if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add(counter + 1 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_group - 1u ? 1 : 0;
    __syncthreads();
    if (!*flag) return false;
    add_rows(partial, (int)g, 8, (int)in_group, nsum, scr, nullptr, partial + (int64_t)(nb + g) * nsum);
    if (threadIdx.x == 0) {
        const unsigned int ngroups = nb < 8u ? nb : 8u;
        *flag = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1u ? 1 : 0;
    }
    return true;
}
// stage B (same flag): true = this is the last workgroup of the grid, `fin` (LDS) holds the nsum totals
__device__ __forceinline__ bool finish_stage_b(double* partial, int nsum, unsigned int* counter, double* scr, double* fin, int* flag) {
    const unsigned int nb = gridDim.x, ngroups = nb < 8u ? nb : 8u;
    __syncthreads();
    if (!*flag) return false;
    if (threadIdx.x < 9) __hip_atomic_store(counter + threadIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    add_rows(partial, (int)nb, 1, (int)ngroups, nsum, scr, fin, nullptr);
    return true;
}

// One entry for both shapes of a grid: up to 128 workgroups the last one to arrive adds all rows itself (ONE ticket, one batch of
// loads); beyond, the two stages above.  true = this workgroup is the last of the grid and `fin` (LDS) holds the nsum totals.
// partial: gridDim.x (+ 8) rows of nsum doubles written by coherent_store and drained; scr: LDS, (blockDim.x / 64) x nsum doubles.
__device__ __forceinline__ bool finish_sums(double* partial, int nsum, unsigned int* counter, double* scr, double* fin) {
    __shared__ int verdict;
    if (gridDim.x > 128u) return finish_stage_a(partial, nsum, counter, scr, &verdict) && finish_stage_b(partial, nsum, counter, scr, fin, &verdict);
    __syncthreads();
    if (threadIdx.x == 0) verdict = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u ? 1 : 0;
    __syncthreads();
    if (!verdict) return false;
    if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    add_rows(partial, 0, 1, (int)gridDim.x, nsum, scr, fin, nullptr);
    return true;
}

// Results of a finishing workgroup and the completion mark behind them (all threads of the workgroup call; fin: the n
// results in LDS).  ONE wave stores the results, waits until every one of those stores has been acknowledged
// (s_waitcnt vmcnt(0): the results span several cache lines, which travel through different L2 channels and would
// otherwise be free to overtake each other and the mark) and only then stores the mark.  The destination is
// fine-grained host memory and results and mark are written through at system scope, so the acknowledged stores are on
// their way in order and a host that sees the mark sees the results.  (A workgroup-scope release alone does not wait for global stores;
// device- or system-scope fences in every wave - an L2 write-back each - made this launch take 8.5 us.)
__device__ __forceinline__ void publish(const double* fin, int n, double* out, double* flag, double mark) {
    __syncthreads();
    if (threadIdx.x < 64) {
        // (system-scope stores: write-through.  A plain store may stay dirty in the L2 until the end of the kernel while
        // the mark - written through - is already visible: the host then reads the results of the evaluation before)
        for (int i = threadIdx.x; i < n; i += 64) __hip_atomic_store(out + i, fin[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (flag) {
            // the mark behind the results: the ONE wave that wrote them drains its stores (gfx9 s_waitcnt encoding:
            // vmcnt(0), the other counters untouched - this library is built for gfx950 only) and releases the mark at
            // system scope; the host acquires it (csrc/ttm_optim.cpp: poll_mark)
#if !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
#error "publish(): the raw s_waitcnt immediate below is the gfx9 encoding"
#endif
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");           // (compiler ordering)
            __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0)
            if (threadIdx.x == 0) __hip_atomic_store(flag, mark, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
#define TTM_FIN_MAX 136         /* >= 1 + TTM_HOSTCOEF_MAX */
