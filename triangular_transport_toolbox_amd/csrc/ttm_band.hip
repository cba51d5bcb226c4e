// ttm_band.hip - forward map and table inverse of BANDED U-form maps in push form (ttm_band.h, include/ttm.h
// "push records").  BASELINE config 5 (d = 40, band 2, order 3, N = 1e6) is the map these kernels are written for;
// reference: transport_map.py:2391-2567 (map, s), 3987-4084 (table root search).
//
// What differs from k_forward_hl / k_inverse_rt (csrc/ttm_kernels.hip), and why (profiles/r03_*):
//   * both were VALU-issue bound at ~50 % utilisation with every VALU instruction costing the same ~4 cycles whatever
//     its type (tools/micro/valu_costs.hip): the lever is the instruction COUNT per component evaluation and the
//     number of basic blocks / dependent LDS round trips a step is cut into;
//   * push form: a row carries LAG running sums (registers) instead of LAG columns with their exp(-x^2/4): no column
//     cache in LDS, no loader waves, no barrier per step, one straight-line step per column;
//   * exp(-x^2/4) = E_i exp(w) from the nearest point of a fixed grid (801 pairs {E_i, y_i/4}, correctly rounded,
//     csrc/ttm_band_etab.h) and the degree-7 Taylor polynomial of exp(w), |w| <= 0.025 for |x| <= 5: 15 instructions
//     instead of 22, relative error <= 2.2e-16 + 4e-18 (|x| <= 5), no clamp / ldexp / NaN repair;
//   * the spline's column and local coordinate come from one fma + cvt + med3 + mad and one fma against a per-column
//     offset read with the coefficients (slot 12 of the column): 5 instructions instead of 9.
// One workgroup of 1024 threads per CU owns a contiguous chunk of rows; a thread carries four rows (two adjacent pairs:
// 16-byte accesses) through all columns; the splines of all components (140 KB at C5) are resident in LDS.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "ttm_band.h"
#include "ttm_band_etab.h"
#include "ttm_band_image.h"

namespace ttm_band {

typedef const __attribute__((address_space(4))) double* cdbl_p;      // uniform data: scalar loads
typedef const __attribute__((address_space(4))) int* cint_p;
struct __attribute__((aligned(16))) D2 { double x, y; };

__device__ const double g_band_etab[2 * TTM_BAND_ET_N] = { TTM_BAND_ETAB_VALUES };
// Taylor coefficients of exp(w): 1/7! .. 1/2! (scalar operands of the fma chain)
__device__ double g_band_taylor[6] = {1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5};

__host__ __device__ constexpr int cls_db(int cls) { return cls == 1 ? 3 : (cls == 2 ? 5 : (cls == 3 ? 7 : 10)); }   // (class 4: few-component kernels only)
__host__ __device__ constexpr int cls_da(int cls) { return cls == 1 ? 1 : (cls == 2 ? 5 : (cls == 3 ? 7 : 10)); }
__host__ __device__ constexpr int cls_gs(int cls) { return cls == 1 ? 8 : (cls == 2 ? 16 : 24); }
__host__ __device__ constexpr int cls_gp(int cls) { return cls_db(cls) + 1 + cls_da(cls); }
__host__ __device__ constexpr int rec_stride(int cls, int lag) { return (TTM_P_HDR + lag * cls_gp(cls) + 7) / 8 * 8; }

#define BAND_ET_DOUBLES (2 * TTM_BAND_ET_N)          /* 12 816 bytes: a multiple of 16 */
#define BAND_CT 1024                                 /* threads per workgroup */
#define BAND_NS 4                                    /* rows per thread: pairs (2t, 2t+1) of the two halves of a tile */
#ifndef BAND_ROWS_TOGETHER
#define BAND_ROWS_TOGETHER 2                         /* rows of a thread whose instructions the scheduler may interleave (1, 2, 4) */
#endif
#ifndef BAND_FWD_NT
#define BAND_FWD_NT 0                                /* non-temporal stores of the forward map's columns */
#endif
#ifndef BAND_INV_NT
#define BAND_INV_NT 0
#endif
#ifndef BAND_FWD_NTL
#define BAND_FWD_NTL 0                               /* non-temporal loads of the forward map's columns */
#endif
#ifndef BAND_INV_NTL
#define BAND_INV_NTL 0
#endif
#ifndef BAND_PF
#define BAND_PF 1                                    /* columns requested ahead of the one being evaluated */
#endif

// One LDS-DMA instruction (16 bytes per lane: the wave's 1 KB lands at lds_wave_base, an LDS byte address, lane after lane),
// written in assembly so that the COMPILER DOES NOT KNOW IT: with a global_load_lds it knows to be in flight hipcc waits for
// vmcnt(0) at the next use of any ordinary load (the column prefetch would be drained every step).  Unknown to it, its
// counted waits are merely one stricter per copy among the younger operations: place the copy behind the step's wait for z,
// and that wait retires the PREVIOUS step's copy (a whole step old) and nothing else.
__device__ __forceinline__ void band_dma16(const void* g, unsigned int lds_wave_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_wave_base) : "m0");
}
__device__ __forceinline__ unsigned int band_lds_addr(const void* p) {
    return (unsigned int)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

// n doubles (even; both sides 16-byte aligned) from memory into LDS: every 16-byte unit requested at once by DMA, then waited
// for - one memory round trip.  (A loop of load / store pairs through registers is compiled to load, wait, store per trip:
// nine dependent round trips for the 140 KB of splines of C5.)  The caller's barrier publishes the bytes.
template <bool WAIT = true>
__device__ __forceinline__ void band_stage(double* lds_dst, const double* src, int n) {
    const int units = n >> 1;
    const unsigned int base = band_lds_addr(lds_dst);
    for (int u0 = 0; u0 < units; u0 += BAND_CT) {
        const int u = u0 + (int)threadIdx.x;
        if (u < units) band_dma16(src + 2 * (size_t)u, __builtin_amdgcn_readfirstlane(base + (unsigned int)(u0 + ((int)threadIdx.x & ~63)) * 16u));
    }
    if (WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------
// push records from the hot records (one workgroup of 64 threads per record)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_band_records(const int* __restrict__ ucomp, const int* __restrict__ ugrp, double* __restrict__ U,
                                                     int64_t h_off, int cls, int ng, int64_t p_off, int lag, int ps, int D) {
    const int r = blockIdx.x, k = r - lag, t = threadIdx.x;
    const int DB = cls_db(cls), DA = cls_da(cls), GS = cls_gs(cls), GP = cls_gp(cls);
    const int hs = TTM_H_HDR + ng * GS;
    double* rec = U + p_off + (int64_t)r * ps;
    auto hot = [&](int kk) { return U + h_off + (int64_t)kk * hs; };
    // the linear term of a component's own variable, if it has one (maps of a few components: A[0], A[1] of the group behind
    // the nonmonotone ones in the component's U-form block)
    auto own = [&](int kk, int deg) {
        const int* uc = ucomp + kk * TTM_UC_LEN;
        return (uc[TTM_UC_FLAGS] & TTM_UCF_OWN) ? U[uc[TTM_UC_DBL_OFF] + 4 + TTM_U_GSTRIDE * uc[TTM_UC_N_GRP] + TTM_U_GHALF + deg] : 0.0;
    };
    // the group of component kk that reads the column `lg` columns in front of it: index into its hot record, -1: none
    auto group_at = [&](int kk, int lg) {
        const int* uc = ucomp + kk * TTM_UC_LEN;
        for (int g = 0; g < uc[TTM_UC_N_GRP]; ++g)
            if (uc[TTM_UC_KC] - ugrp[(uc[TTM_UC_GRP_OFF] + g) * TTM_UG_LEN + TTM_UG_VAR] == lg) return g;
        return -1;
    };
    for (int i = t; i < ps; i += 64) {
        double v = 0.0;
        if (i < 2) {                                          // chain start of the component `lag` columns on
            const int kk = k + lag;
            if (kk >= 0 && kk < D) {
                const double* h = hot(kk);
                v = i == 0 ? h[2] : h[7];
                const int* uc = ucomp + kk * TTM_UC_LEN;
                for (int g = 0; g < uc[TTM_UC_N_GRP]; ++g) v += h[TTM_H_HDR + g * GS + 2 + DB];      // A[0] of the group
                if (i == 0) v += own(kk, 0);                  // (the monotone part's own constant: forward map only)
            }
        } else if (i < TTM_P_HDR) {
            if (k >= 0) {
                const double* h = hot(k);
                const int* uc = ucomp + k * TTM_UC_LEN;
                if (i == 2) v = h[3] + 1.0;
                else if (i == 3) v = h[4];
                else if (i == 4) v = h[5];
                else if (i == 5) { int2 w = {uc[TTM_UC_NI], uc[TTM_UC_TAB_OFF]}; v = *(double*)&w; }
                else if (i == 6) { int2 w = {k, 0}; v = *(double*)&w; }
                else if (i == 7) v = own(k, 1);
            } else if (i == 6) { int2 w = {-1, 0}; v = *(double*)&w; }
        } else {
            const int l = (i - TTM_P_HDR) / GP, j = (i - TTM_P_HDR) % GP;
            const int kk = k + l + 1;
            if (l < lag && kk >= 0 && kk < D) {
                const int g = group_at(kk, l + 1);
                if (g >= 0) {
                    const double* gr = hot(kk) + TTM_H_HDR + g * GS;
                    v = j <= DB ? gr[1 + j] : gr[2 + DB + (j - DB)];              // B[j] | A[j - DB], j - DB = 1..DA
                }
            }
        }
        rec[i] = v;
    }
    if (k >= 0) {                                             // local-coordinate offsets of the spline columns
        const double* h = hot(k);
        const int* uc = ucomp + k * TTM_UC_LEN;
        double* tab = U + uc[TTM_UC_TAB_OFF];
        for (int c = t; c < uc[TTM_UC_NI]; c += 64) tab[c * TTM_U_TSTRIDE + 12] = fma(2.0, h[3], -(double)(2 * c - 1));
    }
}

// column stream accesses: 16 bytes per lane.  The forward map's stores are non-temporal (written once, 320 MB: -6 % launch
// time against plain stores, profiles/r03_*; for the loads and for the inverse's stores the hint changes nothing)
template <bool NT>
__device__ __forceinline__ void band_store2(char* p, double a, double b) {
    typedef double v2 __attribute__((ext_vector_type(2)));
    const v2 v = {a, b};
    if (NT) __builtin_nontemporal_store(v, (v2*)p);
    else *(v2*)p = v;
}
template <bool NT = false>
__device__ __forceinline__ D2 band_load2(const char* p) {
    typedef double v2 __attribute__((ext_vector_type(2)));
    v2 v;
    if (NT) v = __builtin_nontemporal_load((const v2*)p);
    else v = *(const v2*)p;
    const D2 r = {v.x, v.y};
    return r;
}

// ---------------------------------------------------------------------------
// per-row pieces of a step
// ---------------------------------------------------------------------------
__device__ __forceinline__ int band_med3(int v, int lo, int hi) {             // clamp(v, lo, hi), lo <= hi
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "s"(hi));
    return r;
}
__device__ __forceinline__ double band_absmin(double x, double hi) {            // min(|x|, hi); NaN -> hi
    double r;
    asm("v_min_f64 %0, |%1|, %2" : "=v"(r) : "v"(x), "s"(hi));
    return r;
}

// exp(-x^2/4) from the resident pair table
__device__ __forceinline__ double band_expq(const double* etab, double x, cdbl_p kt) {
    const double xe = band_absmin(x, TTM_BAND_ET_XMAX);
    const unsigned int i = (unsigned int)(int)fma(xe, TTM_BAND_ET_INV_STEP, 0.5);
    const D2 ey = *(const D2*)((const char*)etab + (i << 4));
    const double dq = fma(xe, -0.25, ey.y);                   // -(|x| - y_i) / 4
    const double sm = fma(ey.y, 4.0, xe);                     // y_i + |x|
    const double w = dq * sm;
    double p = fma(kt[0], w, kt[1]);
    p = fma(p, w, kt[2]);
    p = fma(p, w, kt[3]);
    p = fma(p, w, kt[4]);
    p = fma(p, w, kt[5]);
    p = fma(p, w, 1.0);
    p = fma(p, w, 1.0);
    return ey.x * p;
}

// pushes of one column value: pend[l] <- pend[l+1] (or the chain start) + group l of the record; g: the LAG group
// blocks of the record (uniform), E = exp(-x^2/4)
template <int DB, int DA, int LAG, class G>
__device__ __forceinline__ void band_push(const G& g, double start, double x, double E, double (&pend)[LAG]) {
    constexpr int GP = DB + 1 + DA;
#pragma unroll
    for (int l = 0; l < LAG; ++l) {
        const auto c = &g[l * GP];
        double b = c[DB];
#pragma unroll
        for (int i = DB - 1; i >= 0; --i) b = fma(b, x, c[i]);
        double a = c[DB + DA];
#pragma unroll
        for (int i = DA - 1; i >= 1; --i) a = fma(a, x, c[DB + i]);
        a = fma(a, x, l + 1 < LAG ? pend[l + 1] : start);
        pend[l] = fma(E, b, a);
    }
}

// the same for the kernels of maps with a few components: only the first LAGE groups of a record (no group of the sweep
// reaches further back), and without the plain polynomials when no group has any (PLAIN false).  GP: doubles per group
// of the record.
template <int DB, int DA, int GP, int LAGE, bool PLAIN, class G>
__device__ __forceinline__ void band_push_e(const G& g, double start, double x, double E, double (&pend)[LAGE]) {
#pragma unroll
    for (int l = 0; l < LAGE; ++l) {
        const auto c = &g[l * GP];
        double b = c[DB];
#pragma unroll
        for (int i = DB - 1; i >= 0; --i) b = fma(b, x, c[i]);
        double a = l + 1 < LAGE ? pend[l + 1] : start;
        if (PLAIN) {
            double q = c[DB + DA];
#pragma unroll
            for (int i = DA - 1; i >= 1; --i) q = fma(q, x, c[DB + i]);
            a = fma(q, x, a);
        }
        pend[l] = fma(E, b, a);
    }
}

// the special-term spline of a component at x: tab = the component's resident table, spl = {1 - t_lo/h, 1/h, 2/h}
__device__ __forceinline__ double band_spline(const double* tab, int nI, double sp_a, double sp_b, double sp_ds, double x) {
    const int col = band_med3((int)fma(x, sp_b, sp_a), 0, nI - 1);
    const double* cp = (const double*)((const char*)tab + __umul24((unsigned int)col, TTM_U_TSTRIDE * 8));
    double c[12];
#pragma unroll
    for (int i = 0; i < 12; i += 2) { const D2 v = *(const D2*)(cp + i); c[i] = v.x; c[i + 1] = v.y; }
    const double s = fma(x, sp_ds, cp[12]);
    double m = c[11];
#pragma unroll
    for (int i = 10; i >= 0; --i) m = fma(m, s, c[i]);
    return m;
}

// ---------------------------------------------------------------------------
// forward map
// ---------------------------------------------------------------------------
// the columns [kb, ke) of one tile; FULL: every row of the tile exists (unmasked stores)
template <int CLS, int LAG, bool FULL>
__device__ __forceinline__ void band_forward_tile(cdbl_p P, cdbl_p kt, const double* etab, const double* tabs, int tab0, int kb, int ke,
                                                  const char* xcol, int64_t ldxb, char* zcol, int64_t ldzb, unsigned int tbase,
                                                  const unsigned int (&roff)[BAND_NS / 2], unsigned int c1_32,
                                                  double (&pend)[BAND_NS][LAG]) {
    constexpr int DB = cls_db(CLS), DA = cls_da(CLS), PS = rec_stride(CLS, LAG);
    constexpr int NS = BAND_NS, NP = NS / 2, HALF = 2 * BAND_CT;
    // The column stream is latency bound (a step is shorter than a loaded memory round trip: profiles/r03_*): the column
    // of step j + PF is requested at the top of step j, PF + 1 register sets take turns (steps are issued PF + 1 at a time
    // with the roles rotated, so no set is ever copied).
    constexpr int PF = BAND_PF, RING = PF + 1;
    D2 xr[RING][NP];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        const char* xc0 = xcol + (int64_t)(kb + i < ke ? i : 0) * ldxb;
#pragma unroll
        for (int q = 0; q < NP; ++q) xr[i][q] = band_load2<BAND_FWD_NTL != 0>(xc0 + roff[q]);
    }
    cdbl_p rec = P + (int64_t)(kb + LAG) * PS;
    auto step = [&](int j, const D2 (&xc)[NP], D2 (&xn)[NP]) {
        {
            const char* xnext = j + PF < ke ? xcol + PF * ldxb : xcol;        // (past the block: a harmless re-read)
#pragma unroll
            for (int q = 0; q < NP; ++q) xn[q] = band_load2<BAND_FWD_NTL != 0>(xnext + roff[q]);
        }
        __builtin_amdgcn_sched_barrier(0);                    // (the scheduler would sink the loads to the end of the step)
        // ---- uniform data of the step ------------------------------------------------------------------------
        const double start = rec[0], sp_a = rec[2], sp_b = rec[3], sp_ds = rec[4];
        cint_p ri = (cint_p)rec;
        const int nI = ri[10];
        const double* tab = tabs + (ri[11] - tab0);
        // ---- row by row (BAND_ROWS_TOGETHER of them scheduled together: each needs 24 registers for its coefficients; the
        // waves of the SIMD, not the rows of a thread, fill each other's latencies) ------------------------------------
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            double zv[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = 2 * q + h;
                const double x = h ? xc[q].y : xc[q].x;
#ifdef BAND_X_NOSPLINE                                       /* (timing experiments: results wrong by construction) */
                const double m = x;
#else
                const double m = band_spline(tab, nI, sp_a, sp_b, sp_ds, x);
#endif
#ifdef BAND_X_NOEXP
                const double E = x;
#else
                const double E = band_expq(etab, x, kt);
#endif
                zv[h] = pend[e][0] + m;
#ifdef BAND_X_NOPUSH
                pend[e][0] = E;
#else
                band_push<DB, DA, LAG>(rec + TTM_P_HDR, start, x, E, pend[e]);
#endif
                if (BAND_ROWS_TOGETHER == 1) __builtin_amdgcn_sched_barrier(0);
            }
            const unsigned int n = tbase + (unsigned int)(q * HALF);
            const D2 o = {zv[0], zv[1]};
            char* zp = zcol + (size_t)(n * 8u);
#ifdef BAND_X_NOSTORE
            if (o.x == 1.2345e300) *(D2*)zp = o;
#else
            if (FULL || n + 1 < c1_32) band_store2<BAND_FWD_NT != 0>(zp, o.x, o.y);
            else if (n < c1_32) *(double*)zp = o.x;
#endif
            if (BAND_ROWS_TOGETHER <= 2) __builtin_amdgcn_sched_barrier(0);
        }
        rec += PS; xcol += ldxb; zcol += ldzb;
    };
    int j = kb;
    for (; j + RING <= ke; j += RING) {
#pragma unroll
        for (int i = 0; i < RING; ++i) step(j + i, xr[i], xr[(i + PF) % RING]);
    }
#pragma unroll
    for (int i = 0; i < RING - 1; ++i)
        if (j + i < ke) step(j + i, xr[i], xr[(i + PF) % RING]);
}

// ---------------------------------------------------------------------------
// fused density pass: S (optional), sum_k log(dS_k/dx_k / sigma_k) and sum_k S_k^2 per row (TM:2569-2712)
// ---------------------------------------------------------------------------
// log(x): fdlibm's polynomial on the reduced argument, division by a Newton reciprocal (<= 2 ulp)
__device__ __forceinline__ double band_log(double x) {
    int e;
    double m = frexp(x, &e);                                  // [0.5, 1)
    const bool small = m < 0.70710678118654752440;
    m = small ? m * 2.0 : m;
    e = small ? e - 1 : e;
    const double f = m - 1.0;
    const double den = 2.0 + f;
    double rc = __builtin_amdgcn_rcp(den);
    rc = fma(fma(-den, rc, 1.0), rc, rc);
    rc = fma(fma(-den, rc, 1.0), rc, rc);
    const double q = f * rc;
    const double sq = fma(fma(-den, q, f), rc, q);
    const double z = sq * sq;
    const double w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                              6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    double res = dk * 6.93147180369123816490e-01 - ((hfsq - (sq * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
    // (0, negative, NaN, inf by selects - the library call here would be inlined into the column loop; denormals are
    // handled by frexp itself: the kernels run with fp64 denormals on)
    res = x == 0.0 ? -INFINITY : res;
    res = x < 0.0 ? NAN : res;
    res = (x != x || x == INFINITY) ? x : res;
    return res;
}

// the spline and its derivative with respect to the local coordinate
__device__ __forceinline__ void band_spline_d(const double* tab, int nI, double sp_a, double sp_b, double sp_ds, double x, double& m, double& dm) {
    const int col = band_med3((int)fma(x, sp_b, sp_a), 0, nI - 1);
    const double* cp = (const double*)((const char*)tab + __umul24((unsigned int)col, TTM_U_TSTRIDE * 8));
    double c[12];
#pragma unroll
    for (int i = 0; i < 12; i += 2) { const D2 v = *(const D2*)(cp + i); c[i] = v.x; c[i + 1] = v.y; }
    const double s = fma(x, sp_ds, cp[12]);
    double a = c[11], da = 0.0;
#pragma unroll
    for (int i = 10; i >= 0; --i) {
        da = fma(da, s, a);
        a = fma(a, s, c[i]);
    }
    m = a; dm = da;
}

// The log-determinant is taken as the log of the PRODUCT of a row's derivatives, renormalised every BAND_LD_CHUNK = 2 columns
// (mantissa kept, binary exponent counted - round 5: every TWO columns; four tail derivatives of 1e-90 each - grid points far
// outside the samples of a monotone part that saturates - underflow a product of four where the reference's sum of logarithms
// is finite): ONE logarithm per row instead of one per evaluation; the uniform factors of the derivatives (2/h of every
// spline, 1/sigma_k) enter as one number per launch.
#define BAND_LD_CHUNK 2
#define BAND_UNI 256                                  /* components of a density pass, at most */
#ifndef BAND_DENS_RT
#define BAND_DENS_RT 2
#endif
#ifndef BAND_DENS_NS
#define BAND_DENS_NS 4
#endif
template <int CLS, int LAG, bool WRITE_Z>
__global__ __launch_bounds__(BAND_CT) void k_band_density(const double* __restrict__ U_, int64_t p_off, int k0, int k1, int kcol0,
                                                          const double* __restrict__ X, int64_t ldx, int64_t N,
                                                          double* __restrict__ Z, int64_t ldz, double* __restrict__ logdet,
                                                          const double* __restrict__ sigma, double* __restrict__ sumsq,
                                                          int64_t rows_per_wg, int Bc) {
    constexpr int DB = cls_db(CLS), DA = cls_da(CLS), PS = rec_stride(CLS, LAG);
    // (two rows per thread: the pass carries three more running values per row than the plain map and is bound by its
    // arithmetic, not by the column stream)
    // (four rows per thread without Z; with it the masked stores of every step copy leave registers for two)
    constexpr int NS = WRITE_Z ? 2 : BAND_DENS_NS, NP = NS / 2, CT = BAND_CT, ROWS = NS * CT, HALF = 2 * CT;
    extern __shared__ __align__(16) double g_lds[];
    double* etab = g_lds;
    double* tabs = g_lds + BAND_ET_DOUBLES;
    const int tid = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * rows_per_wg;
    if (c0 >= N) return;
    const int64_t c1 = c0 + rows_per_wg < N ? c0 + rows_per_wg : N;
    const int ntile = (int)((c1 - c0 + ROWS - 1) / ROWS);
    band_stage<false>(etab, g_band_etab, 2 * TTM_BAND_ET_N);      // (waited for with the first block's splines)
    cdbl_p P = (cdbl_p)(U_ + p_off);
    cdbl_p kt = (cdbl_p)g_band_taylor;
    const unsigned int last_pair = (unsigned int)(((N + 1) & ~(int64_t)1) - 2);
    const unsigned int c1_32 = (unsigned int)c1;
    const int64_t ldxb = ldx * 8, ldzb = ldz * 8;
    // uniform part of the log-determinant: sum_k log(2 / h_k) - sum_k log(sigma_k): one logarithm per thread, summed in
    // component order by thread 0 (every thread taking all of them was a fifth of the launch)
    __shared__ double s_uni[BAND_UNI];
    double luni;
    {
        const int ncomp = k1 - k0;
        for (int k = tid; k < ncomp; k += CT) {
            double v = band_log(P[(int64_t)(k0 + k + LAG) * PS + 4]);
            if (sigma) v -= band_log(sigma[k]);
            s_uni[k] = v;
        }
        __syncthreads();
        if (tid == 0) {
            double acc = 0.0;
            for (int k = 0; k < ncomp; ++k) acc += s_uni[k];
            s_uni[0] = acc;
        }
        __syncthreads();
        luni = s_uni[0];
    }
    for (int tile = 0; tile < ntile; ++tile) {
        const unsigned int tbase = (unsigned int)c0 + (unsigned int)tile * (unsigned int)ROWS + 2u * (unsigned int)tid;
        unsigned int roff[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            unsigned int n = tbase + (unsigned int)(q * HALF);
            n = n < last_pair ? n : last_pair;
            roff[q] = n * 8u;
        }
        // log det = log of the PRODUCT of the row's derivatives: every BAND_LD_CHUNK columns the product is renormalised
        // (mantissa kept, exponent counted: two instructions where a logarithm takes 35), ONE logarithm per row at the end.
        // dmin: the smallest derivative of the row - a negative one makes the row NaN as the reference's log does (two of
        // them would cancel in the product)
        double ss[NS], prod[NS], dmin[NS];
        int pexp[NS];
#pragma unroll
        for (int e = 0; e < NS; ++e) { ss[e] = 0.0; prod[e] = 1.0; dmin[e] = 0.0; pexp[e] = 0; }
        double pend[NS][LAG];
        for (int kb = k0; kb < k1; kb += Bc) {
            const int ke = kb + Bc < k1 ? kb + Bc : k1;
            __syncthreads();
            int tab0;
            {
                cint_p rb = (cint_p)(P + (int64_t)(kb + LAG) * PS), re = (cint_p)(P + (int64_t)(ke - 1 + LAG) * PS);
                tab0 = rb[11];
                const int n = re[11] + TTM_U_TSTRIDE * re[10] - tab0;
                band_stage(tabs, U_ + tab0, n);
            }
            __syncthreads();
            const int colb = kcol0 + (kb - k0);
            if (kb == k0 || true) {
#pragma unroll
                for (int l = 0; l < LAG; ++l) {
                    const double s0 = P[(int64_t)(kb + l) * PS];
#pragma unroll
                    for (int e = 0; e < NS; ++e) pend[e][l] = s0;
                }
                if (colb > 0) {
                    for (int i = 0; i < LAG; ++i) {
                        const int cc = colb - LAG + i;
                        cdbl_p rec = P + (int64_t)(kb + i) * PS;
                        const char* col = (const char*)X + (int64_t)(cc < 0 ? 0 : cc) * ldxb;
#pragma unroll
                        for (int q = 0; q < NP; ++q) {
                            D2 xv = *(const D2*)(col + roff[q]);
                            if (cc < 0) { xv.x = 0.0; xv.y = 0.0; }
                            band_push<DB, DA, LAG>(rec + TTM_P_HDR, rec[0], xv.x, band_expq(etab, xv.x, kt), pend[2 * q]);
                            band_push<DB, DA, LAG>(rec + TTM_P_HDR, rec[0], xv.y, band_expq(etab, xv.y, kt), pend[2 * q + 1]);
                        }
                    }
                }
            }
            const char* xcol = (const char*)X + (int64_t)colb * ldxb;
            char* zcol = WRITE_Z ? (char*)Z + (int64_t)(kb - k0) * ldzb : nullptr;
            cdbl_p rec = P + (int64_t)(kb + LAG) * PS;
            D2 xa[NP], xb[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) xa[q] = band_load2(xcol + roff[q]);
            auto step = [&](int j, const D2 (&xc)[NP], D2 (&xn)[NP]) {
                {
                    const char* xnext = j + 1 < ke ? xcol + ldxb : xcol;
#pragma unroll
                    for (int q = 0; q < NP; ++q) xn[q] = band_load2(xnext + roff[q]);
                }
                __builtin_amdgcn_sched_barrier(0);
                const double start = rec[0], sp_a = rec[2], sp_b = rec[3], sp_ds = rec[4];
                cint_p ri = (cint_p)rec;
                const int nI = ri[10];
                const double* tab = tabs + (ri[11] - tab0);
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    double zv[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int e = 2 * q + h;
                        const double x = h ? xc[q].y : xc[q].x;
                        double m, dm;
                        band_spline_d(tab, nI, sp_a, sp_b, sp_ds, x, m, dm);
                        const double E = band_expq(etab, x, kt);
                        zv[h] = pend[e][0] + m;
                        ss[e] = fma(zv[h], zv[h], ss[e]);
                        prod[e] *= dm;
                        dmin[e] = fmin(dmin[e], dm);
                        band_push<DB, DA, LAG>(rec + TTM_P_HDR, start, x, E, pend[e]);
                        if (BAND_DENS_RT == 1) __builtin_amdgcn_sched_barrier(0);
                    }
                    if (BAND_DENS_RT == 2) __builtin_amdgcn_sched_barrier(0);
                    if (WRITE_Z) {
                        const unsigned int n = tbase + (unsigned int)(q * HALF);
                        char* zp = zcol + (size_t)(n * 8u);
                        if (n + 1 < c1_32) band_store2<false>(zp, zv[0], zv[1]);
                        else if (n < c1_32) *(double*)zp = zv[0];
                    }
                }
                if (WRITE_Z && (((j - k0) & (BAND_LD_CHUNK - 1)) == BAND_LD_CHUNK - 1)) {
#pragma unroll
                    for (int e = 0; e < NS; ++e) { int ex; prod[e] = frexp(prod[e], &ex); pexp[e] += ex; }
                }
                rec += PS; xcol += ldxb;
                if (WRITE_Z) zcol += ldzb;
            };
            // four columns at a time, then ONE place where the chunk's products go through the logarithm (inside the step the
            // four inlined logarithms of every step copy cost the registers of two rows)
            auto flush_ld = [&]() {
#pragma unroll
                for (int e = 0; e < NS; ++e) { int ex; prod[e] = frexp(prod[e], &ex); pexp[e] += ex; }
            };
            int j = kb;
            if (WRITE_Z) {                                    // (fewer step copies: two columns at a time)
                for (; j + 1 < ke; j += 2) {
                    step(j, xa, xb);
                    step(j + 1, xb, xa);
                }
            } else {
                for (; j + 3 < ke; j += 4) {
                    step(j, xa, xb);
                    step(j + 1, xb, xa);
                    flush_ld();
                    step(j + 2, xa, xb);
                    step(j + 3, xb, xa);
                    flush_ld();
                }
                for (; j + 1 < ke; j += 2) {
                    step(j, xa, xb);
                    step(j + 1, xb, xa);
                    flush_ld();
                }
            }
            if (j < ke) step(j, xa, xb);
            flush_ld();
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const unsigned int n = tbase + (unsigned int)(q * HALF);
            if (logdet) {
                double lv[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int e = 2 * q + h;
                    lv[h] = fma((double)pexp[e], 6.93147180559945286e-01, band_log(prod[e])) + luni;
                    lv[h] = dmin[e] < 0.0 ? NAN : lv[h];
                }
                if (n + 1 < c1_32) band_store2<false>((char*)(logdet + n), lv[0], lv[1]);
                else if (n < c1_32) logdet[n] = lv[0];
            }
            if (sumsq) {
                if (n + 1 < c1_32) band_store2<false>((char*)(sumsq + n), ss[2 * q], ss[2 * q + 1]);
                else if (n < c1_32) sumsq[n] = ss[2 * q];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// log-determinant only (TM:2569-2712: sum_k log(dS_k/dx_k / sigma_k) per row, no map values): what the pullback /
// pushforward densities take on the raw samples (the reference evaluates its derivative basis on un-standardised x).
// dS_k/dx_k of a separable component is a function of x_k ALONE - the derivative of its special-term spline (+ the slope of
// a linear own term) - so the pass needs no running sums, no exp(-x^2/4), no pushes: per evaluation the spline's column
// (fma, cvt, med3, mad), its coefficients (six 16-byte LDS reads + the column offset), 21 FMAs for the derivative of the
// degree-11 local polynomial, one multiplication into the row's running product.  29 vector instructions (the fused
// density pass k_band_density: 65), bound by the LDS gathers (104 bytes per row and column) and the column stream.
// The product is renormalised every TWO columns (mantissa kept, binary exponent counted): derivatives of 1e-150 - grid
// points far outside the samples of a map whose monotone part saturates - cannot underflow it; ONE logarithm per row.
// The uniform factors (2/h_k of a spline without a linear term beside it, 1/sigma_k) enter once per launch.
// LDS: [splines of a block of components, as they stand in the U section]
// ---------------------------------------------------------------------------
__device__ __forceinline__ double band_spline_dd(const double* tab, int nI, double sp_a, double sp_b, double sp_ds, double x) {
    const int col = band_med3((int)fma(x, sp_b, sp_a), 0, nI - 1);
    const double* cp = (const double*)((const char*)tab + __umul24((unsigned int)col, TTM_U_TSTRIDE * 8));
    double c[12];
#pragma unroll
    for (int i = 0; i < 12; i += 2) { const D2 v = *(const D2*)(cp + i); c[i] = v.x; c[i + 1] = v.y; }
    // (c[0] is not part of the derivative; left unused, its half of the first 16-byte read is dropped and the remaining
    // coefficients are re-paired (c1, c2), (c3, c4) ... as ds_read2_b64 - half the rate of ds_read_b128 and banked modulo 32:
    // 36 % of the LDS cycles were bank conflicts.  The empty statement keeps the six aligned 16-byte reads.)
    asm volatile("" :: "v"(c[0]));
    const double s = fma(x, sp_ds, cp[12]);
    double a = c[11], da = 0.0;
#pragma unroll
    for (int i = 10; i >= 1; --i) {
        da = fma(da, s, a);
        a = fma(a, s, c[i]);
    }
    return fma(da, s, a);
}

#ifndef BAND_LD_PF
#define BAND_LD_PF 3
#endif
template <int LAG, int NS>
__global__ __launch_bounds__(BAND_CT) void k_band_logdet(const double* __restrict__ U_, int64_t p_off, int k0, int k1, int kcol0, int ps,
                                                         const double* __restrict__ X, int64_t ldx, int64_t N,
                                                         double* __restrict__ logdet, const double* __restrict__ sigma,
                                                         int64_t rows_per_wg, int Bc) {
    constexpr int NP = NS / 2, CT = BAND_CT, ROWS = NS * CT, HALF = 2 * CT;
    extern __shared__ __align__(16) double g_lds[];
    double* tabs = g_lds;
    const int tid = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * rows_per_wg;
    if (c0 >= N) return;
    const int64_t c1 = c0 + rows_per_wg < N ? c0 + rows_per_wg : N;
    const int ntile = (int)((c1 - c0 + ROWS - 1) / ROWS);
    cdbl_p P = (cdbl_p)(U_ + p_off);
    const int64_t PS = ps;
    const unsigned int last_pair = (unsigned int)(((N + 1) & ~(int64_t)1) - 2);
    const unsigned int c1_32 = (unsigned int)c1;
    const int64_t ldxb = ldx * 8;
    // uniform part: sum_k [log(2 / h_k) for a spline without a linear term beside it] - sum_k log(sigma_k), one logarithm
    // per thread, summed in component order by thread 0
    __shared__ double s_uni[BAND_UNI];
    double luni;
    {
        const int ncomp = k1 - k0;
        for (int k = tid; k < ncomp; k += CT) {
            cdbl_p rec = P + (int64_t)(k0 + k + LAG) * PS;
            cint_p ri = (cint_p)rec;
            double v = (ri[10] > 0 && rec[7] == 0.0) ? band_log(rec[4]) : 0.0;
            if (sigma) v -= band_log(sigma[k]);
            s_uni[k] = v;
        }
        __syncthreads();
        if (tid == 0) {
            double acc = 0.0;
            for (int k = 0; k < ncomp; ++k) acc += s_uni[k];
            s_uni[0] = acc;
        }
        __syncthreads();
        luni = s_uni[0];
    }
    for (int kb = k0; kb < k1; kb += Bc) {
        const int ke = kb + Bc < k1 ? kb + Bc : k1;
        const bool first = kb == k0, last = ke == k1;
        __syncthreads();
        int tab0;
        {
            // (components without special terms have no spline: the block's splines lie between the first and the last that has one)
            int tb = -1, te = 0;
            for (int k = kb; k < ke; ++k) {
                cint_p r = (cint_p)(P + (int64_t)(k + LAG) * PS);
                if (r[10] > 0) { if (tb < 0) tb = r[11]; te = r[11] + TTM_U_TSTRIDE * r[10]; }
            }
            tab0 = tb < 0 ? 0 : tb;
            const int n = tb < 0 ? 0 : te - tb;
            band_stage(tabs, U_ + tab0, n);
        }
        __syncthreads();
        for (int tile = 0; tile < ntile; ++tile) {
            const unsigned int tbase = (unsigned int)c0 + (unsigned int)tile * (unsigned int)ROWS + 2u * (unsigned int)tid;
            unsigned int roff[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                unsigned int n = tbase + (unsigned int)(q * HALF);
                n = n < last_pair ? n : last_pair;
                roff[q] = n * 8u;
            }
            double prod[NS], dmin[NS];
            int pexp[NS];
#pragma unroll
            for (int e = 0; e < NS; ++e) { prod[e] = 1.0; dmin[e] = 0.0; pexp[e] = 0; }
            const char* xcol = (const char*)X + (int64_t)(kcol0 + (kb - k0)) * ldxb;
            cdbl_p rec = P + (int64_t)(kb + LAG) * PS;
            // The column stream is latency bound (a step of a SIMD's four waves is shorter than a loaded memory round trip): the
            // column of step j + PF is requested at the top of step j, PF + 1 register sets take turns (steps are issued
            // PF + 1 at a time with the roles rotated: no set is ever copied).  This pass has the registers for PF = 3
            // (six 1 KB loads in flight per wave: 96 KB per CU).
            constexpr int PF = BAND_LD_PF, RING = PF + 1;
            D2 xr[RING][NP];
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const char* xc0 = xcol + (int64_t)(kb + i < ke ? i : 0) * ldxb;
#pragma unroll
                for (int q = 0; q < NP; ++q) xr[i][q] = band_load2(xc0 + roff[q]);
            }
            auto step = [&](int j, const D2 (&xc)[NP], D2 (&xn)[NP]) {
                {
                    const char* xnext = j + PF < ke ? xcol + PF * ldxb : xcol;       // (past the block: a harmless re-read)
#if defined(BAND_XL) && BAND_XL == 2                          /* (timing experiment: no column stream) */
#pragma unroll
                    for (int q = 0; q < NP; ++q) { xn[q].x = xc[q].x + 1e-9; xn[q].y = xc[q].y - 1e-9; }
                    (void)xnext;
#else
#pragma unroll
                    for (int q = 0; q < NP; ++q) xn[q] = band_load2(xnext + roff[q]);
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
                const double sp_a = rec[2], sp_b = rec[3], sp_ds = rec[4], own1 = rec[7];
                cint_p ri = (cint_p)rec;
                const int nI = ri[10];
                const double* tab = tabs + (ri[11] - tab0);
                if (nI > 0 && own1 == 0.0) {
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int e = 2 * q + h;
#if defined(BAND_XL) && BAND_XL == 1                          /* (timing experiment: results wrong by construction) */
                            const double dm = h ? xc[q].y : xc[q].x;
#else
                            const double dm = band_spline_dd(tab, nI, sp_a, sp_b, sp_ds, h ? xc[q].y : xc[q].x);
#endif
                            prod[e] *= dm;
                            dmin[e] = fmin(dmin[e], dm);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int e = 2 * q + h;
                            const double x = h ? xc[q].y : xc[q].x;
                            const double dm = nI > 0 ? band_spline_dd(tab, nI, sp_a, sp_b, sp_ds, x) : 0.0;
                            const double dx = fma(dm, sp_ds, fma(x, 0.0, own1));     // (x 0: a NaN / infinite sample stays NaN)
                            prod[e] *= dx;
                            dmin[e] = fmin(dmin[e], dx);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                rec += PS; xcol += ldxb;
            };
            auto flush = [&]() {
#pragma unroll
                for (int e = 0; e < NS; ++e) { int ex; prod[e] = frexp(prod[e], &ex); pexp[e] += ex; }
            };
            int j = kb;
            for (; j + RING <= ke; j += RING) {
#pragma unroll
                for (int i = 0; i < RING; ++i) {
                    step(j + i, xr[i], xr[(i + PF) % RING]);
                    if (i & 1) flush();
                }
                if (RING & 1) flush();
            }
#pragma unroll
            for (int i = 0; i < RING - 1; ++i) {
                if (j + i < ke) {
                    step(j + i, xr[i], xr[(i + PF) % RING]);
                    if (i & 1) flush();
                }
            }
            flush();
            // a block's share of the row's log-determinant: kept in the output between the blocks (one block at C5)
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const unsigned int n = tbase + (unsigned int)(q * HALF);
                double lv[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int e = 2 * q + h;
                    lv[h] = fma((double)pexp[e], 6.93147180559945286e-01, band_log(prod[e]));
                    lv[h] = dmin[e] < 0.0 ? NAN : lv[h];
                    if (last) lv[h] += luni;
                }
                if (!first) {
                    if (n + 1 < c1_32) { const D2 o = *(const D2*)(logdet + n); lv[0] += o.x; lv[1] += o.y; }
                    else if (n < c1_32) lv[0] += logdet[n];
                }
                if (n + 1 < c1_32) band_store2<false>((char*)(logdet + n), lv[0], lv[1]);
                else if (n < c1_32) logdet[n] = lv[0];
            }
        }
    }
}

// LDS: [E table: 2 x 801 | splines of the block's components, as they stand in the U section]
template <int CLS, int LAG>
__global__ __launch_bounds__(BAND_CT) void k_band_forward(const double* __restrict__ U_, int64_t p_off, int k0, int k1, int kcol0,
                                                          const double* __restrict__ X, int64_t ldx, int64_t N,
                                                          double* __restrict__ Z, int64_t ldz, int64_t rows_per_wg, int Bc) {
    constexpr int DB = cls_db(CLS), DA = cls_da(CLS), PS = rec_stride(CLS, LAG);
    constexpr int NS = BAND_NS, NP = NS / 2, CT = BAND_CT, ROWS = NS * CT, HALF = 2 * CT;
    extern __shared__ __align__(16) double g_lds[];
    double* etab = g_lds;
    double* tabs = g_lds + BAND_ET_DOUBLES;
    const int tid = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * rows_per_wg;
    if (c0 >= N) return;
    const int64_t c1 = c0 + rows_per_wg < N ? c0 + rows_per_wg : N;
    const int ntile = (int)((c1 - c0 + ROWS - 1) / ROWS);
    band_stage<false>(etab, g_band_etab, 2 * TTM_BAND_ET_N);      // (waited for with the first block's splines)
    cdbl_p P = (cdbl_p)(U_ + p_off);
    cdbl_p kt = (cdbl_p)g_band_taylor;
    const unsigned int last_pair = (unsigned int)(((N + 1) & ~(int64_t)1) - 2);      // first row of the last readable pair
    const unsigned int c1_32 = (unsigned int)c1;
    const int64_t ldxb = ldx * 8, ldzb = ldz * 8;

    for (int kb = k0; kb < k1; kb += Bc) {
        const int ke = kb + Bc < k1 ? kb + Bc : k1;
        __syncthreads();                                      // every wave is done with the previous block's splines
        int tab0;
        {
            cint_p rb = (cint_p)(P + (int64_t)(kb + LAG) * PS), re = (cint_p)(P + (int64_t)(ke - 1 + LAG) * PS);
            tab0 = rb[11];
            const int n = re[11] + TTM_U_TSTRIDE * re[10] - tab0;           // doubles (even)
            band_stage(tabs, U_ + tab0, n);
        }
        __syncthreads();
        const int colb = kcol0 + (kb - k0);                   // column of component kb
        for (int tile = 0; tile < ntile; ++tile) {
            const unsigned int tbase = (unsigned int)c0 + (unsigned int)tile * (unsigned int)ROWS + 2u * (unsigned int)tid;
            const bool full = c0 + (int64_t)(tile + 1) * ROWS <= c1;
            unsigned int roff[NP];                            // byte offsets of this thread's pairs within a column (clamped: loads)
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                unsigned int n = tbase + (unsigned int)(q * HALF);
                n = n < last_pair ? n : last_pair;
                roff[q] = n * 8u;
            }
            // running sums of the components kb .. kb + LAG - 1 at this point of the sweep
            double pend[NS][LAG];
#pragma unroll
            for (int l = 0; l < LAG; ++l) {
                const double s = P[(int64_t)(kb + l) * PS];
#pragma unroll
                for (int e = 0; e < NS; ++e) pend[e][l] = s;
            }
            if (colb > 0) {
                // the LAG columns in front of the block, pushed without being evaluated (a column that does not exist has
                // an all-zero record: with x = 0, E = 1 its step only shifts the sums)
                for (int i = 0; i < LAG; ++i) {
                    const int cc = colb - LAG + i;
                    cdbl_p rec = P + (int64_t)(kb + i) * PS;
                    const char* col = (const char*)X + (int64_t)(cc < 0 ? 0 : cc) * ldxb;
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        D2 xv = *(const D2*)(col + roff[q]);
                        if (cc < 0) { xv.x = 0.0; xv.y = 0.0; }
                        band_push<DB, DA, LAG>(rec + TTM_P_HDR, rec[0], xv.x, band_expq(etab, xv.x, kt), pend[2 * q]);
                        band_push<DB, DA, LAG>(rec + TTM_P_HDR, rec[0], xv.y, band_expq(etab, xv.y, kt), pend[2 * q + 1]);
                    }
                }
            }
            const char* xcol = (const char*)X + (int64_t)colb * ldxb;
            char* zcol = (char*)Z + (int64_t)(kb - k0) * ldzb;
            if (full) band_forward_tile<CLS, LAG, true>(P, kt, etab, tabs, tab0, kb, ke, xcol, ldxb, zcol, ldzb, tbase, roff, c1_32, pend);
            else band_forward_tile<CLS, LAG, false>(P, kt, etab, tabs, tab0, kb, ke, xcol, ldxb, zcol, ldzb, tbase, roff, c1_32, pend);
        }
    }
}

// ---------------------------------------------------------------------------
// maps of a few components (at most TTM_P_FEW_D: the spiral / temperature / banana examples, the filter and smoother
// blocks): a launch is a handful of microseconds, so what counts is the chain of memory round trips in front of the
// first store.  Here it is ONE: the E table, every spline and ALL columns of the tile are requested together (tables
// first - the counter of outstanding loads retires in order), the tables go to LDS while the columns travel, then the
// columns are walked in registers.  Same arithmetic per row as k_band_forward / k_band_density.
// DENS: also sum_k log(dS_k/dx_k / sigma_k) and sum_k S_k^2 per row (Z optional)
// ---------------------------------------------------------------------------
#ifndef BAND_FEW_STAGE
#define BAND_FEW_STAGE 0
#endif
#ifndef BAND_FEWI_STAGE
#define BAND_FEWI_STAGE 0
#endif
#ifndef BAND_FEW_NS
#define BAND_FEW_NS 2                                 /* rows per thread and tile of the few-component kernels */
#endif
#ifndef BAND_FEW_NT
#define BAND_FEW_NT 1                                 /* non-temporal stores of the few-component kernels */
#endif
#define BAND_FEW_TAB (4 * BAND_CT)                    /* doubles of splines a launch can stage (two 16-byte loads per thread) */
// LAG: groups per push record (u_p_lag); LAGE <= LAG: how far back a group of the sweep reaches; PLAIN: some group has
// plain polynomial terms
template <int CLS, int LAG, int LAGE, bool PLAIN, bool DENS>
__global__ __launch_bounds__(BAND_CT) void k_band_few(const double* __restrict__ U_, int64_t p_off, int k0, int k1, int kcol0,
                                                      const double* __restrict__ X, int64_t ldx, int64_t N,
                                                      double* __restrict__ Z, int64_t ldz, double* __restrict__ logdet,
                                                      const double* __restrict__ sigma, double* __restrict__ sumsq, int ntiles,
                                                      int tab0, int ntab) {
    constexpr int DB = cls_db(CLS), DA = cls_da(CLS), GP = cls_gp(CLS), PS = rec_stride(CLS, LAG);
    // two rows (one 16-byte pair) per thread and tile: with tiles of 2048 rows even half a million rows reach every CU;
    // the columns of a workgroup's next tile are requested before the current one is evaluated
    constexpr int NS = BAND_FEW_NS, NP = NS / 2, CT = BAND_CT, ROWS = NS * CT, HALF = 2 * CT, FD = TTM_P_FEW_D;
    extern __shared__ __align__(16) double g_lds[];
    double* etab = g_lds;
    double* tabs = g_lds + BAND_ET_DOUBLES;
    const int tid = threadIdx.x;
    const int nc = k1 - k0;
#if BAND_FEW_STAGE == 1                                      /* (timing experiments: results wrong by construction) */
    if (N > 0) return;
#endif
    cdbl_p P = (cdbl_p)(U_ + p_off);
    cdbl_p kt = (cdbl_p)g_band_taylor;
    const unsigned int last_pair = (unsigned int)(((N + 1) & ~(int64_t)1) - 2);
    const unsigned int N32 = (unsigned int)N;
    const int64_t ldxb = ldx * 8, ldzb = ldz * 8;
    // tables (tab0: offset of the first spline in the U section, ntab doubles - even, <= BAND_FEW_TAB: from the host, which
    // knows them, so the request does not wait for a record): requested now, written to LDS after the first tile's columns
    // have been requested
    const D2 ev = *(const D2*)(g_band_etab + 2 * min(tid, TTM_BAND_ET_N - 1));
    D2 sv[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) sv[r] = *(const D2*)(U_ + tab0 + min(2 * tid + r * 2 * CT, ntab - 2));
    __builtin_amdgcn_sched_barrier(0);
    double luni = 0.0;
    bool staged = false;
    // every column of a tile: the LAGE columns in front of the first component (conditioning columns; none: zeros) and the
    // components' own (clamped duplicates beyond the last: unconditional loads stay in flight together)
    auto request = [&](int tile, D2 (&xf)[LAGE][NP], D2 (&xin)[FD][NP]) {
        unsigned int roff[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            unsigned int n = (unsigned int)tile * (unsigned int)ROWS + 2u * (unsigned int)tid + (unsigned int)(q * HALF);
            n = n < last_pair ? n : last_pair;
            roff[q] = n * 8u;
        }
        if (kcol0 > 0) {
#pragma unroll
            for (int i = 0; i < LAGE; ++i) {
                const int cc = kcol0 - LAGE + i;
                const char* col = (const char*)X + (int64_t)(cc < 0 ? 0 : cc) * ldxb;
#pragma unroll
                for (int q = 0; q < NP; ++q) xf[i][q] = band_load2(col + roff[q]);
            }
        }
#pragma unroll
        for (int j = 0; j < FD; ++j) {
            const char* col = (const char*)X + (int64_t)(kcol0 + min(j, nc - 1)) * ldxb;
#pragma unroll
            for (int q = 0; q < NP; ++q) xin[j][q] = band_load2(col + roff[q]);
        }
    };
    D2 xf[LAGE][NP], xin[FD][NP], xfn[LAGE][NP], xinn[FD][NP];
    if ((int)blockIdx.x < ntiles) request(blockIdx.x, xf, xin);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const unsigned int tbase = (unsigned int)tile * (unsigned int)ROWS + 2u * (unsigned int)tid;
        __builtin_amdgcn_sched_barrier(0);
        if (!staged) {
            if (tid < TTM_BAND_ET_N) *(D2*)(etab + 2 * tid) = ev;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i = 2 * tid + r * 2 * CT;
                if (i < ntab) *(D2*)(tabs + i) = sv[r];
            }
            if (DENS && sigma) {
                // uniform part of the log-determinant: - sum_k log(sigma_k), in component order (the 2 / h_k of the splines
                // are applied per evaluation here: a component may have a linear term next to its spline, or no spline)
                for (int k = 0; k < nc; ++k) luni -= band_log(sigma[k]);
            }
            __syncthreads();
            staged = true;
        }
        // (the density pass has no registers to spare for a second set of columns: its next tile is requested after this one)
        const bool more = tile + (int)gridDim.x < ntiles;
        if (more && !DENS) request(tile + gridDim.x, xfn, xinn);
        __builtin_amdgcn_sched_barrier(0);
#if BAND_FEW_STAGE == 2 || BAND_FEW_STAGE == 3
        {
            double acc = etab[tid & 511] + tabs[tid & 255];
#pragma unroll
            for (int j = 0; j < FD; ++j)
#pragma unroll
                for (int q = 0; q < NP; ++q) acc += xin[j][q].x + xin[j][q].y;
            if (BAND_FEW_STAGE == 3) {
#pragma unroll
                for (int j = 0; j < FD; ++j)
                    if (j < nc)
#pragma unroll
                        for (int q = 0; q < NP; ++q) {
                            const unsigned int n = tbase + (unsigned int)(q * HALF);
                            if (n + 1 < N32) band_store2<false>((char*)Z + (int64_t)j * ldzb + (size_t)(n * 8u), acc, xin[j][q].y);
                        }
            } else if (acc == 1.2345e300) Z[tid] = acc;
            continue;
        }
#endif
        double pend[NS][LAGE];
#pragma unroll
        for (int l = 0; l < LAGE; ++l) {
            const double s0 = P[(int64_t)(k0 + l) * PS];
#pragma unroll
            for (int e = 0; e < NS; ++e) pend[e][l] = s0;
        }
        if (kcol0 > 0) {
#pragma unroll
            for (int i = 0; i < LAGE; ++i) {
                // (record of the column LAGE - i in front of component k0; the chain it starts is component k0 + i's)
                const bool there = kcol0 - LAGE + i >= 0;
                cdbl_p rec = P + (int64_t)(k0 + LAG - LAGE + i) * PS;
                const double start = P[(int64_t)(k0 + i) * PS];
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const double xa = there ? xf[i][q].x : 0.0, xb = there ? xf[i][q].y : 0.0;
                    band_push_e<DB, DA, GP, LAGE, PLAIN>(rec + TTM_P_HDR, start, xa, band_expq(etab, xa, kt), pend[2 * q]);
                    band_push_e<DB, DA, GP, LAGE, PLAIN>(rec + TTM_P_HDR, start, xb, band_expq(etab, xb, kt), pend[2 * q + 1]);
                }
            }
        }
        double ss[NS], prod[NS], dmin[NS];
        int pexp[NS];
#pragma unroll
        for (int e = 0; e < NS; ++e) { ss[e] = 0.0; prod[e] = 1.0; dmin[e] = 0.0; pexp[e] = 0; }
#pragma unroll
        for (int j = 0; j < FD; ++j) {
            if (DENS && j == 2 && nc > 2) {
                // the product of a row's derivatives is renormalised after two factors (mantissa kept, binary exponent counted):
                // derivatives of 1e-150 in every column - grid points far outside the samples of a monotone part that saturates -
                // would underflow a product of four where the reference's sum of logarithms is finite
#pragma unroll
                for (int e = 0; e < NS; ++e) { int ex; prod[e] = frexp(prod[e], &ex); pexp[e] = ex; }
            }
            if (j < nc) {
                cdbl_p rec = P + (int64_t)(k0 + j + LAG) * PS;
                const double start = P[(int64_t)(k0 + j + LAGE) * PS];        // (of the component LAGE columns on)
                const double sp_a = rec[2], sp_b = rec[3], sp_ds = rec[4], own1 = rec[7];      // own1: slope of the component's linear term
                cint_p ri = (cint_p)rec;
                const int nI = ri[10];                                                  // (0: no special terms, no spline)
                const double* tab = tabs + (ri[11] - tab0);
                char* zcol = (char*)Z + (int64_t)j * ldzb;
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    double zv[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int e = 2 * q + h;
                        const double x = h ? xin[j][q].y : xin[j][q].x;
                        double m = 0.0, dm = 0.0;
                        if (nI > 0) {
                            if (DENS) band_spline_d(tab, nI, sp_a, sp_b, sp_ds, x, m, dm);
                            else m = band_spline(tab, nI, sp_a, sp_b, sp_ds, x);
                        }
                        const double E = band_expq(etab, x, kt);
                        zv[h] = fma(own1, x, pend[e][0] + m);
                        if (DENS) {
                            const double dx = fma(dm, sp_ds, own1);           // dS_k/dx_k in standardised units
                            ss[e] = fma(zv[h], zv[h], ss[e]);
                            prod[e] *= dx;
                            dmin[e] = fmin(dmin[e], dx);
                        }
                        band_push_e<DB, DA, GP, LAGE, PLAIN>(rec + TTM_P_HDR, start, x, E, pend[e]);
                    }
                    if (!DENS || Z) {
                        const unsigned int n = tbase + (unsigned int)(q * HALF);
                        char* zp = zcol + (size_t)(n * 8u);
                        if (n + 1 < N32) band_store2<BAND_FEW_NT != 0>(zp, zv[0], zv[1]);
                        else if (n < N32) *(double*)zp = zv[0];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (DENS) {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const unsigned int n = tbase + (unsigned int)(q * HALF);
                if (logdet) {
                    // (a negative derivative makes the row NaN, as the reference's log does: two would cancel in the product)
                    const double la = dmin[2 * q] < 0.0 ? NAN : fma((double)pexp[2 * q], 6.93147180559945286e-01, band_log(prod[2 * q])) + luni;
                    const double lb = dmin[2 * q + 1] < 0.0 ? NAN : fma((double)pexp[2 * q + 1], 6.93147180559945286e-01, band_log(prod[2 * q + 1])) + luni;
                    if (n + 1 < N32) band_store2<false>((char*)(logdet + n), la, lb);
                    else if (n < N32) logdet[n] = la;
                }
                if (sumsq) {
                    if (n + 1 < N32) band_store2<false>((char*)(sumsq + n), ss[2 * q], ss[2 * q + 1]);
                    else if (n < N32) sumsq[n] = ss[2 * q];
                }
            }
        }
        if (more && DENS) request(tile + gridDim.x, xf, xin);
        if (more && !DENS) {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int i = 0; i < LAGE; ++i) xf[i][q] = xfn[i][q];
#pragma unroll
                for (int jj = 0; jj < FD; ++jj) xin[jj][q] = xinn[jj][q];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// table inverse (TM:3987-4084) in push form
// ---------------------------------------------------------------------------
// The resident-table machinery is that of k_inverse_rt (csrc/ttm_kernels.hip: blocks of components, windowed tables,
// 16-bit bucket index, the index kernel's bucket function bit for bit, outliers redone from the table in memory); the
// step is new: the nonmonotone offset of a component is the running sum pend[0] its predecessors pushed, the solved
// x_k is pushed on at once (exp(-x_k^2/4) from the located interval), and a step whose rows all lie inside the
// resident window - all but a handful per million - is ONE basic block.
// LDS (doubles): [tables: B x tab_slot | {E_i, y_i}: 2 x Weven]
// table slot: [wl, wh, bucket scale, bucket bias, int32 {entries per bucket at most, 0}, int32 {degenerate, 0} | xs: W
//              entries + 4 sentinels (+inf), rounded up to even | bucket index: nb + 1 uint16]
#define BAND_RT_KMAX 24                                /* components per block, at most (BAND_RT_HDR, band_bucket_params: ttm_band_image.h) */

// exp(-x^2/4) by the series (block boundaries, outliers): Cody-Waite reduction + degree-12 Taylor, <= 2 ulp
__device__ __forceinline__ double band_expq_series(double x) {
    const double y = fmax(-0.25 * (x * x), -800.0);
    const double k = rint(y * 1.4426950408889634);
    double r = fma(-k, 6.93147180369123816490e-01, y);
    r = fma(-k, 1.90821492927058770002e-10, r);
    double p = 2.08767569878681e-09;
    p = fma(p, r, 2.505210838544172e-08);
    p = fma(p, r, 2.755731922398589e-07);
    p = fma(p, r, 2.7557319223985893e-06);
    p = fma(p, r, 2.48015873015873e-05);
    p = fma(p, r, 0.0001984126984126984);
    p = fma(p, r, 0.001388888888888889);
    p = fma(p, r, 0.008333333333333333);
    p = fma(p, r, 0.041666666666666664);
    p = fma(p, r, 0.16666666666666666);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return fma(x, 0.0, ldexp(p, (int)k));
}

// LDS addresses as 32-bit pointers (a laundered generic pointer would be read through the flat path)
typedef const __attribute__((address_space(3))) char* lds_p;
__device__ __forceinline__ lds_p band_lds(const void* p) { return (lds_p)p; }
__device__ __forceinline__ double band_lds_f64(lds_p p) { return *(const __attribute__((address_space(3))) double*)p; }
// ... an 8-byte read the load/store optimiser cannot pair with its neighbour (ds_read2_b64 is served at a quarter of the
// rate of two ds_read_b64 and banks modulo 32)
__device__ __forceinline__ double band_lds_f64_single(lds_p p) {
    asm volatile("" : "+v"(p));
    return *(const __attribute__((address_space(3))) double*)p;
}
typedef double band_v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ D2 band_lds_pair(lds_p p) {
    const band_v2 v = *(const __attribute__((address_space(3))) band_v2*)p;
    const D2 r = {v.x, v.y};
    return r;
}

// everything a tile needs to walk the columns [kb, ke) of a block
struct BandInvCtx {
    cdbl_p P, kt;
    const double* tabs;            // resident tables of the block
    lds_p etab1;                   // pairs {E_i, y_i}: entry i - 1 of the whole table at etab1 + 16 i
    const double* tab_x; const double* tmin; const double* tmax;
    int T, nb, tab_slot, w0, Weven, k0;
    double y0, ystep, y0m;
    int64_t ldzb, ldxb;
    unsigned int c1_32;
};

// The ring of resident tables (k_band_inverse_ring): step s of a workgroup's flat sequence (tile by tile, column by column)
// finds the image of its component in slot s mod R.  EVERY step copies ONE image by DMA - each wave one 1 KB piece, without a
// condition, so that the compiler's count of the loads in flight stays exact and the wait for the next column of z covers
// nothing else: the image of step s - G + R goes into the slot step s - G used.  The workgroup meets at a barrier every G
// steps, so that slot has been free since the last meeting; a piece requested before the last meeting has landed by the next
// one (counted wait: every step since put at least three vector-memory operations behind it), the barrier makes that true
// of every wave's piece, and the image is used at step s - G + R >= s + 2 G (R >= 3 G).  Past the last column the copies are
// harmless repeats.
struct BandRing {
    const char* src;               // this lane's 16 bytes inside the image of component k0
    unsigned int dst;              // where the wave's piece starts inside a slot (bytes)
    unsigned int tabs_lds;         // LDS address of slot 0
    unsigned int stride;           // bytes per image / slot
    int R, ncomp;
    int pos;                       // slot of the next step
    int tpos, tcomp;               // slot the next step refills, component whose image goes there
    int gcnt;                      // steps since the last meeting
};

// the images of the steps [0, n) into the slots [0, n) (in front of the first column: 16 bytes per lane and round)
__device__ __forceinline__ void band_ring_fill(const double* img, int ncomp, double* tabs, int tab_slot, int n) {
    const int U = tab_slot >> 1;                                              // 16-byte units per image
    const int total = n * U;
    for (int u0 = 0; u0 < total; u0 += BAND_CT) {
        const int u = u0 + (int)threadIdx.x;
        if (u < total) {
            const int t = u / U, o = u - t * U;
            band_dma16(img + (size_t)(t % ncomp) * tab_slot + 2 * o,
                       __builtin_amdgcn_readfirstlane(band_lds_addr(tabs) + (unsigned int)(u0 + ((int)threadIdx.x & ~63)) * 16u));
        }
    }
}

template <int CLS, int LAG, bool RING = false, int G = 0>
__device__ __forceinline__ void band_inverse_tile(const BandInvCtx& cx, bool full, int kb, int ke, const char* zcol, char* xcol, unsigned int tbase,
                                                  const unsigned int (&roff)[BAND_NS / 2], double (&pend)[BAND_NS][LAG],
                                                  const D2 (&zfirst)[BAND_NS / 2], bool have_first, BandRing* rg = nullptr) {
    constexpr int DB = cls_db(CLS), DA = cls_da(CLS), PS = rec_stride(CLS, LAG);
    constexpr int NS = BAND_NS, NP = NS / 2, HALF = 2 * BAND_CT;
    cdbl_p rec = cx.P + (int64_t)(kb + LAG) * PS;
    cdbl_p kt = cx.kt;
    const double* slot = RING ? cx.tabs + (size_t)rg->pos * cx.tab_slot : cx.tabs;
    const double ystep = cx.ystep;
    const int nb1 = cx.nb - 1;
    D2 za[NP], zb[NP];
    if (have_first) {                                         // (the chunk's first tile: requested before the block's tables)
#pragma unroll
        for (int q = 0; q < NP; ++q) za[q] = zfirst[q];
    } else {
#pragma unroll
        for (int q = 0; q < NP; ++q) za[q] = band_load2<BAND_INV_NTL != 0>(zcol + roff[q]);
    }
    // interp1d slope form (TM:4062-4065) in the located interval and exp(-x^2/4) = E[i-1] exp(w), w = -delta (y_lo + x) / 4
    auto interp = [&](double y_lo, double x_lo, double x_hi, double e_lo, double tgt, double& rr, double& ee) {
        const double dx = fmax(x_hi - x_lo, 1e-300);                          // (tie at a flat start: k_inverse_rt)
        double rc = __builtin_amdgcn_rcp(dx);
        rc = fma(fma(-dx, rc, 1.0), rc, rc);
        const double delta = (ystep * rc) * (tgt - x_lo);
        rr = delta + y_lo;
        const double w = (delta * -0.25) * (y_lo + rr);
        double p = fma(kt[0], w, kt[1]);
        p = fma(p, w, kt[2]);
        p = fma(p, w, kt[3]);
        p = fma(p, w, kt[4]);
        p = fma(p, w, kt[5]);
        p = fma(p, w, 1.0);
        p = fma(p, w, 1.0);
        ee = e_lo * p;
    };
    auto step = [&](int j, const D2 (&zc)[NP], D2 (&zn)[NP]) {
        {
            const char* znext = j + 1 < ke ? zcol + cx.ldzb : zcol;           // (past the block: a harmless re-read)
#pragma unroll
            for (int q = 0; q < NP; ++q) zn[q] = band_load2<BAND_INV_NTL != 0>(znext + roff[q]);
        }
        // the record of the step, requested NOW (left to itself the compiler loads the group coefficients where the pushes
        // use them - at the end of the step's dependent chain, a scalar-cache round trip on the critical path)
        double start = rec[1];
        double gc[LAG * (DB + 1 + DA)];
#pragma unroll
        for (int i = 0; i < LAG * (DB + 1 + DA); ++i) gc[i] = rec[TTM_P_HDR + i];
        asm volatile("" : "+s"(start));
#pragma unroll
        for (int i = 0; i < LAG * (DB + 1 + DA); ++i) asm volatile("" : "+s"(gc[i]));
        __builtin_amdgcn_sched_barrier(0);
        double wl, wh, scale, bias;
        { const D2 a = *(const D2*)slot; wl = a.x; wh = a.y; const D2 b = *(const D2*)(slot + 2); scale = b.x; bias = b.y; }
        const lds_p xsl = band_lds(slot + BAND_RT_HDR) - 8 * cx.w0;           // xs indexed by the entry's number in the whole table
        const lds_p bkl = band_lds(slot + BAND_RT_HDR + cx.Weven);
        const bool deg = __builtin_amdgcn_readfirstlane(((const int*)slot)[10]) != 0;
        const int per = __builtin_amdgcn_readfirstlane(((const int*)slot)[8]);
        double traw[NS], tg[NS], r[NS], E[NS];
        unsigned long long outl = deg ? ~0ull : 0ull;
#pragma unroll
        for (int e = 0; e < NS; ++e) {
            const double z = (e & 1) ? zc[e >> 1].y : zc[e >> 1].x;
            traw[e] = z - pend[e][0];
            tg[e] = fmin(fmax(traw[e], wl), wh);
            outl |= __builtin_amdgcn_ballot_w64(traw[e] != tg[e]);
        }
        if (RING && G > 0) {
            // the step's piece of an image (BandRing), behind the wait for this step's z (the ballot needs every row's target)
            band_dma16(rg->src + (size_t)rg->tcomp * rg->stride, rg->tabs_lds + (unsigned int)rg->tpos * rg->stride + rg->dst);
            if (++rg->tpos == rg->R) rg->tpos = 0;
            if (++rg->tcomp == rg->ncomp) rg->tcomp = 0;
        }
        // the four rows in phases, so that their dependent LDS reads travel together: bucket -> the bucket's entries ->
        // the interval.  np.searchsorted(xs, target) (left) = entries in lower buckets + entries of the target's own
        // bucket that are below it (the bucket function is monotone: k_table_index); NC = entries compared (>= the most
        // any bucket of this table holds).  The target is clipped to the window's value range, so every read is resident
        // whatever the row (an outlier's result is replaced below).
        auto search = [&](auto nc) {
            constexpr int NC = decltype(nc)::value;
            int pos[NS];
#pragma unroll
            for (int e = 0; e < NS; ++e) {
                const int bi = min((int)fma(tg[e], scale, bias), nb1);
                pos[e] = (int)*(const __attribute__((address_space(3))) unsigned short*)(bkl + 2 * bi);
            }
            double qv[NS][NC];
#pragma unroll
            for (int e = 0; e < NS; ++e) {
                const lds_p q4 = xsl + 8 * pos[e];
                qv[e][0] = band_lds_f64(q4);
#pragma unroll
                for (int i = 1; i < NC; ++i) qv[e][i] = band_lds_f64_single(q4 + 8 * i);
            }
            D2 ey[NS];
            double xlo[NS], xhi[NS];
#pragma unroll
            for (int e = 0; e < NS; ++e) {
#pragma unroll
                for (int i = 0; i < NC; ++i) pos[e] += qv[e][i] < tg[e] ? 1 : 0;
                ey[e] = band_lds_pair(cx.etab1 + 16 * pos[e]);
                const lds_p xp = xsl + 8 * pos[e];
                xlo[e] = band_lds_f64(xp - 8);
                xhi[e] = band_lds_f64_single(xp);
            }
#pragma unroll
            for (int e = 0; e < NS; ++e) interp(ey[e].y, xlo[e], xhi[e], ey[e].x, tg[e], r[e], E[e]);
        };
#if defined(BAND_XI_NOSEARCH)                                /* (timing experiments: results wrong by construction) */
#pragma unroll
        for (int e = 0; e < NS; ++e) interp(tg[e], wl, wh, scale, tg[e], r[e], E[e]);
#elif defined(BAND_XI_NOSOLVE)
#pragma unroll
        for (int e = 0; e < NS; ++e) { r[e] = tg[e]; E[e] = traw[e]; }
#else
        if (per <= 2) search(std::integral_constant<int, 2>());
        else search(std::integral_constant<int, 4>());
#endif
#ifndef BAND_XI_NOOUTL
        if (outl != 0) {
            // outliers (the tails of the table, beyond the window, NaN; every row of a degenerate table): clip as
            // TM:4074-4076, np.searchsorted (left) over the whole row in memory, the same interpolation
            const double* xg = cx.tab_x + (int64_t)(j - cx.k0) * cx.T;
            const double lo = cx.tmin[j - cx.k0], hi = cx.tmax[j - cx.k0];
#pragma unroll
            for (int e = 0; e < NS; ++e) {
                if (deg || traw[e] != tg[e]) {
                    double t = traw[e];
                    const double cl = fmin(fmax(t, lo), hi);
                    t = t != t ? t : cl;                                      // a NaN target stays NaN
                    int a = 0, b = cx.T;
                    while (a < b) {
                        const int mid = (a + b) >> 1;
                        if (xg[mid] < t) a = mid + 1; else b = mid;
                    }
                    const int i = min(max(a, 1), cx.T - 1);
                    interp(fma((double)i, ystep, cx.y0m), xg[i - 1], xg[i], band_expq_series((double)(i - 1) * ystep + cx.y0), t, r[e], E[e]);
                }
            }
        }
#endif
        // x_k is pushed on to the components that read it, and stored
#ifdef BAND_XI_NOPUSH
#pragma unroll
        for (int e = 0; e < NS; ++e) pend[e][0] = E[e] * start;
#else
#pragma unroll
        for (int e = 0; e < NS; ++e) band_push<DB, DA, LAG>(gc, start, r[e], E[e], pend[e]);
#endif
#ifdef BAND_XI_NOSTORE
        if (r[0] == 1.2345e300)
#endif
        if (full) {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                band_store2<BAND_INV_NT != 0>(xcol + (size_t)((tbase + (unsigned int)(q * HALF)) * 8u), r[2 * q], r[2 * q + 1]);
            }
        } else {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const unsigned int n = tbase + (unsigned int)(q * HALF);
                const D2 o = {r[2 * q], r[2 * q + 1]};
                char* xp = xcol + (size_t)(n * 8u);
                if (n + 1 < cx.c1_32) *(D2*)xp = o;
                else if (n < cx.c1_32) *(double*)xp = o.x;
            }
        }
        rec += PS; slot += cx.tab_slot; zcol += cx.ldzb; xcol += cx.ldxb;
        if (RING) {
            if (++rg->pos == rg->R) { rg->pos = 0; slot = cx.tabs; }
            if (G > 0 && ++rg->gcnt == G) {
                rg->gcnt = 0;
                // this wave's pieces from before the last meeting have landed: every step since put two column loads and a piece behind them
                static_assert(G == 0 || G == 4, "the counted wait below is 3 G - 2");
                asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                __syncthreads();
            }
        }
#ifdef BAND_XI_BARRIERS                                      /* (timing experiment: what a workgroup barrier every so many columns costs) */
        if ((j - kb) % BAND_XI_BARRIERS == BAND_XI_BARRIERS - 1) __syncthreads();
#endif
    };
    int j = kb;
    for (; j + 1 < ke; j += 2) {
        step(j, za, zb);
        step(j + 1, zb, za);
    }
    if (j < ke) step(j, za, zb);
}

// KM: components per block, at most (what the table load is unrolled for: 4 for maps of a few components, BAND_RT_KMAX)
template <int CLS, int LAG, int KM>
__global__ __launch_bounds__(BAND_CT) void k_band_inverse(const double* __restrict__ U_, int64_t p_off, int k0, int k1, int kcol0,
                                                          const double* __restrict__ Z, int64_t ldz, double* X, int64_t ldx, int64_t N,
                                                          const double* __restrict__ tab_x, int T, double y0, double ystep, double ylast,
                                                          const double* __restrict__ tmin, const double* __restrict__ tmax,
                                                          const int* __restrict__ bkt, int nb, int tab_slot, int B, int64_t rows_per_wg,
                                                          int w0, int W) {
    constexpr int DB = cls_db(CLS), DA = cls_da(CLS), PS = rec_stride(CLS, LAG);
    constexpr int NS = BAND_NS, NP = NS / 2, CT = BAND_CT, ROWS = NS * CT, HALF = 2 * CT;
    extern __shared__ __align__(16) double g_lds[];
    const int tid = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * rows_per_wg;
    if (c0 >= N) return;
    const int64_t c1 = c0 + rows_per_wg < N ? c0 + rows_per_wg : N;
    const int ntile = (int)((c1 - c0 + ROWS - 1) / ROWS);
    const int Weven = (W + 4 + 1) & ~1;
    double* tabs = g_lds;
    double* etab = tabs + (size_t)B * tab_slot;
    cdbl_p P = (cdbl_p)(U_ + p_off);
    const unsigned int last_pair = (unsigned int)(((N + 1) & ~(int64_t)1) - 2);
    const int64_t ldzb = ldz * 8, ldxb = ldx * 8;
    const double y0m = y0 - ystep;
    BandInvCtx cx;
    cx.P = P; cx.kt = (cdbl_p)g_band_taylor; cx.tabs = tabs;
    cx.etab1 = band_lds(etab) - 16 * w0 - 16;
    cx.tab_x = tab_x; cx.tmin = tmin; cx.tmax = tmax;
    cx.T = T; cx.nb = nb; cx.tab_slot = tab_slot; cx.w0 = w0; cx.Weven = Weven; cx.k0 = k0;
    cx.y0 = y0; cx.ystep = ystep; cx.y0m = y0m; cx.ldzb = ldzb; cx.ldxb = ldxb; cx.c1_32 = (unsigned int)c1;
    for (int i = tid; i < W; i += CT) {
        // E of grid point w0 + i and, next to it, that point's abscissa exactly as the interpolation forms it from the
        // interval number (fma(i + 1, step, y0 - step))
        etab[2 * i] = band_expq_series(w0 + i == T - 1 ? ylast : (double)(w0 + i) * ystep + y0);
        etab[2 * i + 1] = fma((double)(w0 + i + 1), ystep, y0m);
    }
    double pend[NS][LAG];
#pragma unroll
    for (int e = 0; e < NS; ++e)
#pragma unroll
        for (int l = 0; l < LAG; ++l) pend[e][l] = 0.0;

    // Blocks of at most B components; odd workgroups take the short block first, so that neighbouring CUs reload their
    // tables at different times (a block switch stalls a CU for ~6 us; out of phase, the others use the bandwidth)
    int kfirst = B;
    if ((blockIdx.x & 1) && (k1 - k0) % B) kfirst = (k1 - k0) % B;
    for (int kb = k0, ke; kb < k1; kb = ke) {
        ke = kb + (kb == k0 ? kfirst : B);
        ke = ke < k1 ? ke : k1;
        const int nk = ke - kb;
        // the first column of z of the chunk's first tile is requested before the tables: one memory round trip instead of two
        // in a row (a block switch is a string of them; 8 us each before: profiles/r03_*)
        D2 zfirst[NP];
        {
            const char* zc0 = (const char*)Z + (int64_t)(kb - k0) * ldzb;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                unsigned int n = (unsigned int)c0 + 2u * (unsigned int)tid + (unsigned int)(q * HALF);
                n = n < last_pair ? n : last_pair;
                zfirst[q] = band_load2<BAND_INV_NTL != 0>(zc0 + (size_t)(n * 8u));
            }
        }
        __syncthreads();                                      // every wave is done with the previous block's tables
        // the tables of the block: every load of the block in flight at once (element i of every table by thread i; the
        // bucket indices sixteen bytes at a time), the search parameters next to them
#ifdef BAND_XI_NOSWITCH                                      /* (timing experiment: later blocks search the first block's tables) */
        if (kb == k0)
#endif
        {
            constexpr int KMAX = KM;                          // components per block, at most (the host plans for it)
            double lo = 0.0, hi = 0.0;
            if (tid < nk) { lo = tmin[kb - k0 + tid]; hi = tmax[kb - k0 + tid]; }
            {
                double v[KMAX];
                const int i = tid < Weven ? tid : Weven - 1;  // (clamped, unconditional: all loads in flight together)
                const double* src = tab_x + (int64_t)(kb - k0) * T + w0 + min(i, W - 1);
#pragma unroll
                for (int u = 0; u < KMAX; ++u) v[u] = src[(int64_t)min(u, nk - 1) * T];
                // (stores unconditional too: a slot beyond the block's last table repeats that table's value - the clamped
                // load - into that table's slot; branches here would spill the values in flight)
#pragma unroll
                for (int u = 0; u < KMAX; ++u) tabs[(size_t)min(u, nk - 1) * tab_slot + BAND_RT_HDR + i] = i < W ? v[u] : INFINITY;
            }
            {
                constexpr int BR = (KMAX * 256 + CT - 1) / CT; // rounds of the bucket copy: a component's nb + 1 = 1024 int32 = 256 x 16 bytes
                int4 bv[BR];
#pragma unroll
                for (int rd = 0; rd < BR; ++rd) {
                    const int idx = rd * CT + tid, c = min(idx >> 8, nk - 1), w = idx & 255;
                    bv[rd] = *(const int4*)(bkt + (int64_t)(kb - k0 + c) * (nb + 1) + 4 * w);
                }
#pragma unroll
                for (int rd = 0; rd < BR; ++rd) {
                    const int idx = rd * CT + tid, c = min(idx >> 8, nk - 1), w = idx & 255;
                    unsigned short* bs = (unsigned short*)(tabs + (size_t)c * tab_slot + BAND_RT_HDR + Weven) + 4 * w;
                    const uint2 pk = {(unsigned int)(bv[rd].x & 0xffff) | ((unsigned int)bv[rd].y << 16),
                                      (unsigned int)(bv[rd].z & 0xffff) | ((unsigned int)bv[rd].w << 16)};
                    *(uint2*)bs = pk;
                }
            }
            if (tid < nk) {
                double* slot = tabs + (size_t)tid * tab_slot;
                double scale, bias;
                band_bucket_params(lo, hi, nb, scale, bias);
                slot[2] = scale; slot[3] = bias;
                ((int*)slot)[8] = 0; ((int*)slot)[9] = 0;
                slot[5] = 0.0;
            }
        }
        __syncthreads();
        // entries per bucket, at most; the value range [wl, wh] whose search stays inside the window (k_inverse_rt)
#ifdef BAND_XI_NOSWITCH
        if (kb == k0)
#endif
        for (int c = tid >> 6; c < nk; c += CT >> 6) {
            const unsigned short* bs = (const unsigned short*)(tabs + (size_t)c * tab_slot + BAND_RT_HDR + Weven);
            int per = 0;
            {                                                 // lane l: buckets 16 l .. 16 l + 15 from two 16-byte reads (nb + 1 = 1024 entries)
                const int l = tid & 63;
                const uint4 qa = *(const uint4*)(bs + 16 * l), qb = *(const uint4*)(bs + 16 * l + 8);
                const unsigned int nxt = l < 63 ? bs[16 * l + 16] : 0u;
                const unsigned int w[9] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w, nxt};
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int b0 = (int)((w[i >> 1] >> (16 * (i & 1))) & 0xffffu);
                    const int b1 = (int)((w[(i + 1) >> 1] >> (16 * ((i + 1) & 1))) & 0xffffu);
                    if (16 * l + i < nb && b1 > w0 && b0 < w0 + W) per = max(per, b1 - b0);
                }
            }
            for (int o = 32; o > 0; o >>= 1) per = max(per, __shfl_xor(per, o));
            double* slot = tabs + (size_t)c * tab_slot;
            const int pm = max(per, 1);
            const bool deg = pm + 3 >= W || per > 4;                           // (more entries in a bucket than the resident search compares: every row by the outlier path)
            if ((tid & 63) == 0) {
                const double* xw = slot + BAND_RT_HDR;
                ((int*)slot)[8] = per;
                ((int*)slot)[10] = deg ? 1 : 0;
                slot[0] = deg ? xw[1] : xw[pm];
                slot[1] = deg ? xw[1] : xw[W - 2];
            }
            if (deg)
                for (int i = tid & 63; i <= nb; i += 64) const_cast<unsigned short*>(bs)[i] = (unsigned short)(w0 + 1);
        }
        __syncthreads();

        const int colb = kcol0 + (kb - k0);                   // column of component kb
        for (int tile = 0; tile < ntile; ++tile) {
            const unsigned int tbase = (unsigned int)c0 + (unsigned int)tile * (unsigned int)ROWS + 2u * (unsigned int)tid;
            const bool full = c0 + (int64_t)(tile + 1) * ROWS <= c1;
            unsigned int roff[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                unsigned int n = tbase + (unsigned int)(q * HALF);
                n = n < last_pair ? n : last_pair;
                roff[q] = n * 8u;
            }
            if (!(ntile == 1 && kb > k0)) {
                // running sums at the block's first column from the LAG columns in front of it (conditioning columns, or
                // what the same thread stored in the block before); a chunk of ONE tile keeps them in registers instead
                for (int i = 0; i < LAG; ++i) {
                    const int cc = colb - LAG + i;
                    cdbl_p rec = P + (int64_t)(kb + i) * PS;
                    const char* col = (const char*)X + (int64_t)(cc < 0 ? 0 : cc) * ldxb;
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        D2 xv = {0.0, 0.0};
                        if (cc >= 0) xv = *(const D2*)(col + roff[q]);
                        band_push<DB, DA, LAG>(rec + TTM_P_HDR, rec[1], xv.x, cc >= 0 ? band_expq_series(xv.x) : 1.0, pend[2 * q]);
                        band_push<DB, DA, LAG>(rec + TTM_P_HDR, rec[1], xv.y, cc >= 0 ? band_expq_series(xv.y) : 1.0, pend[2 * q + 1]);
                    }
                }
            }
            const char* zcol = (const char*)Z + (int64_t)(kb - k0) * ldzb;
            char* xcol = (char*)X + (int64_t)colb * ldxb;
            band_inverse_tile<CLS, LAG>(cx, full, kb, ke, zcol, xcol, tbase, roff, pend, zfirst, tile == 0);
        }
    }
}

// ---------------------------------------------------------------------------
// k_band_inverse with the tables of a RING of components resident (BandRing, ttm_band_image.h): the images come ready-made
// from the kernel that built the tables and are copied into LDS by DMA - all R slots in front of the first column (one round
// trip: the first column of z is requested before them), then G at a time behind a barrier every G columns, two groups
// ahead of their use.  A tile walks ALL its columns in one go: the running sums never leave the registers, so the result of a
// row does not depend on the chunking (k_band_inverse re-reads LAG columns at a block boundary when a chunk has several tiles
// and takes their exp(-x^2/4) from the series there).  The step itself is band_inverse_tile, shared with k_band_inverse.
template <int CLS, int LAG, int G>
__global__ __launch_bounds__(BAND_CT) void k_band_inverse_ring(const double* __restrict__ U_, int64_t p_off, int k0, int k1, int kcol0,
                                                               const double* __restrict__ Z, int64_t ldz, double* X, int64_t ldx, int64_t N,
                                                               const double* __restrict__ tab_x, int T, double y0, double ystep, double ylast,
                                                               const double* __restrict__ tmin, const double* __restrict__ tmax,
                                                               const double* __restrict__ img, int nb, int tab_slot, int R,
                                                               int64_t rows_per_wg, int w0, int W) {
    constexpr int DB = cls_db(CLS), DA = cls_da(CLS), PS = rec_stride(CLS, LAG);
    constexpr int NS = BAND_NS, NP = NS / 2, CT = BAND_CT, ROWS = NS * CT, HALF = 2 * CT;
    extern __shared__ __align__(16) double g_lds[];
    const int tid = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * rows_per_wg;
    if (c0 >= N) return;
    const int64_t c1 = c0 + rows_per_wg < N ? c0 + rows_per_wg : N;
    const int ntile = (int)((c1 - c0 + ROWS - 1) / ROWS);
    const int ncomp = k1 - k0;
    const int Weven = (W + 4 + 1) & ~1;
    double* tabs = g_lds;
    double* etab = tabs + (size_t)R * tab_slot;
    cdbl_p P = (cdbl_p)(U_ + p_off);
    const unsigned int last_pair = (unsigned int)(((N + 1) & ~(int64_t)1) - 2);
    const int64_t ldzb = ldz * 8, ldxb = ldx * 8;
    const double y0m = y0 - ystep;
    BandInvCtx cx;
    cx.P = P; cx.kt = (cdbl_p)g_band_taylor; cx.tabs = tabs;
    cx.etab1 = band_lds(etab) - 16 * w0 - 16;
    cx.tab_x = tab_x; cx.tmin = tmin; cx.tmax = tmax;
    cx.T = T; cx.nb = nb; cx.tab_slot = tab_slot; cx.w0 = w0; cx.Weven = Weven; cx.k0 = k0;
    cx.y0 = y0; cx.ystep = ystep; cx.y0m = y0m; cx.ldzb = ldzb; cx.ldxb = ldxb; cx.c1_32 = (unsigned int)c1;
    BandRing rg;
    {
        const int U = tab_slot >> 1, nw = (U + 63) >> 6;      // 16-byte units per image, waves that cover one
        const int piece = (tid >> 6) % nw;                    // (the other waves repeat a piece: the same bytes to the same place)
        const int start = 64 * piece < U - 64 ? 64 * piece : U - 64;
        rg.src = (const char*)img + (size_t)(start + (tid & 63)) * 16;
        rg.dst = __builtin_amdgcn_readfirstlane((unsigned int)start * 16u);
        rg.tabs_lds = band_lds_addr(tabs);
        rg.stride = (unsigned int)tab_slot * 8u;
        rg.R = R; rg.ncomp = ncomp; rg.pos = 0; rg.gcnt = 0;
        rg.tpos = G > 0 ? R - G : 0;
        rg.tcomp = G > 0 ? (R - G) % ncomp : 0;
    }
    // the first column of z of the first tile, then the images of the first R steps: one memory round trip
    D2 zfirst[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        unsigned int n = (unsigned int)c0 + 2u * (unsigned int)tid + (unsigned int)(q * HALF);
        n = n < last_pair ? n : last_pair;
        zfirst[q] = band_load2<BAND_INV_NTL != 0>((const char*)Z + (size_t)(n * 8u));
    }
    band_ring_fill(img, ncomp, tabs, tab_slot, R);
    for (int i = tid; i < W; i += CT) {
        etab[2 * i] = band_expq_series(w0 + i == T - 1 ? ylast : (double)(w0 + i) * ystep + y0);
        etab[2 * i + 1] = fma((double)(w0 + i + 1), ystep, y0m);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int tile = 0; tile < ntile; ++tile) {
        const unsigned int tbase = (unsigned int)c0 + (unsigned int)tile * (unsigned int)ROWS + 2u * (unsigned int)tid;
        const bool full = c0 + (int64_t)(tile + 1) * ROWS <= c1;
        unsigned int roff[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            unsigned int n = tbase + (unsigned int)(q * HALF);
            n = n < last_pair ? n : last_pair;
            roff[q] = n * 8u;
        }
        // running sums at the first column from the LAG columns in front of it (conditioning columns, or nothing)
        double pend[NS][LAG];
#pragma unroll
        for (int e = 0; e < NS; ++e)
#pragma unroll
            for (int l = 0; l < LAG; ++l) pend[e][l] = 0.0;
        for (int i = 0; i < LAG; ++i) {
            const int cc = kcol0 - LAG + i;
            cdbl_p rec = P + (int64_t)(k0 + i) * PS;
            const char* col = (const char*)X + (int64_t)(cc < 0 ? 0 : cc) * ldxb;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                D2 xv = {0.0, 0.0};
                if (cc >= 0) xv = *(const D2*)(col + roff[q]);
                band_push<DB, DA, LAG>(rec + TTM_P_HDR, rec[1], xv.x, cc >= 0 ? band_expq_series(xv.x) : 1.0, pend[2 * q]);
                band_push<DB, DA, LAG>(rec + TTM_P_HDR, rec[1], xv.y, cc >= 0 ? band_expq_series(xv.y) : 1.0, pend[2 * q + 1]);
            }
        }
        band_inverse_tile<CLS, LAG, true, G>(cx, full, k0, k1, (const char*)Z, (char*)X + (int64_t)kcol0 * ldxb, tbase, roff, pend, zfirst, tile == 0, &rg);
    }
}

// ---------------------------------------------------------------------------
// table inverse of maps with a few components: as k_band_few, ONE memory round trip in front of the first store - every
// table (whole, no window), every bucket index and all columns of the tile are requested together.  Tables of any shape:
// the bucket of a target is searched four ways at a time (three resident reads per phase, as many phases as the
// fullest bucket of the component needs: 1 up to 3 entries, 2 up to 15, 3 up to 63, 4 up to 255), where k_band_inverse
// compares up to four entries and sends fuller tables row by row to memory.
// LDS (doubles): [tables: nc x tab_slot | {E_i, y_i}: 2 x Weven];  table slot: [scale, bias, int32 {fullest bucket, 0},
// 0, 0, 0 (the entry "in front of the first") | xs: T entries + sentinels (+inf) up to Weven | bucket index: nb + 1 uint16]
// Needs nb + 1 = 1024 (one 16-byte load of bucket starts per thread) and T + 4 <= 1024 (entry i by thread i).
// ---------------------------------------------------------------------------
template <int CLS, int LAG, int LAGE, bool PLAIN>
__global__ __launch_bounds__(BAND_CT) void k_band_few_inverse(const double* __restrict__ U_, int64_t p_off, int k0, int k1, int kcol0,
                                                              const double* __restrict__ Z, int64_t ldz, double* X, int64_t ldx, int64_t N,
                                                              const double* __restrict__ tab_x, int T, double y0, double ystep, double ylast,
                                                              const double* __restrict__ tmin, const double* __restrict__ tmax,
                                                              const int* __restrict__ bkt, int nb, int tab_slot, int ntiles) {
    constexpr int DB = cls_db(CLS), DA = cls_da(CLS), GP = cls_gp(CLS), PS = rec_stride(CLS, LAG);
    constexpr int NS = BAND_FEW_NS, NP = NS / 2, CT = BAND_CT, ROWS = NS * CT, HALF = 2 * CT, FD = TTM_P_FEW_D, HDR = BAND_RT_HDR;
    extern __shared__ __align__(16) double g_lds[];
    const int tid = threadIdx.x;
    const int nc = k1 - k0;
    const int Weven = (T + 4 + 1) & ~1;
    double* tabs = g_lds;
    double* etab = tabs + (size_t)nc * tab_slot;
    const lds_p etab1 = band_lds(etab) - 16;                  // pair of entry i - 1 at etab1 + 16 i
    cdbl_p P = (cdbl_p)(U_ + p_off);
    cdbl_p kt = (cdbl_p)g_band_taylor;
    const unsigned int last_pair = (unsigned int)(((N + 1) & ~(int64_t)1) - 2);
    const unsigned int N32 = (unsigned int)N;
    const int64_t ldzb = ldz * 8, ldxb = ldx * 8;
    const double y0m = y0 - ystep;
    const int nb1 = nb - 1;
    // tables and bucket starts: requested now (first: the counter of outstanding loads retires in order)
    double tv[FD];
    {
        const double* src = tab_x + min(tid, T - 1);
#pragma unroll
        for (int u = 0; u < FD; ++u) tv[u] = src[(int64_t)min(u, nc - 1) * T];
    }
    const int bc = min(tid >> 8, nc - 1), bw = tid & 255;     // this thread's four bucket starts: component, first bucket / 4
    const int4 bv = *(const int4*)(bkt + (int64_t)bc * (nb + 1) + 4 * bw);
    __builtin_amdgcn_sched_barrier(0);
    // interp1d slope form (TM:4062-4065) in the located interval and exp(-x^2/4) = E[i-1] exp(w), w = -delta (y_lo + x) / 4
    auto interp = [&](double y_lo, double x_lo, double x_hi, double e_lo, double tgt, double& rr, double& ee) {
        const double dx = fmax(x_hi - x_lo, 1e-300);                          // (tie at a flat start: k_inverse_rt)
        double rc = __builtin_amdgcn_rcp(dx);
        rc = fma(fma(-dx, rc, 1.0), rc, rc);
        const double delta = (ystep * rc) * (tgt - x_lo);
        rr = delta + y_lo;
        const double w = (delta * -0.25) * (y_lo + rr);
        double p = fma(kt[0], w, kt[1]);
        p = fma(p, w, kt[2]);
        p = fma(p, w, kt[3]);
        p = fma(p, w, kt[4]);
        p = fma(p, w, kt[5]);
        p = fma(p, w, 1.0);
        p = fma(p, w, 1.0);
        ee = e_lo * p;
    };
    // every column of a tile: the conditioning columns in front of the first component (already in X) and z
    auto request = [&](int tile, D2 (&xf)[LAGE][NP], D2 (&zin)[FD][NP]) {
        unsigned int roff[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            unsigned int n = (unsigned int)tile * (unsigned int)ROWS + 2u * (unsigned int)tid + (unsigned int)(q * HALF);
            n = n < last_pair ? n : last_pair;
            roff[q] = n * 8u;
        }
        if (kcol0 > 0) {
#pragma unroll
            for (int i = 0; i < LAGE; ++i) {
                const int cc = kcol0 - LAGE + i;
                const char* col = (const char*)X + (int64_t)(cc < 0 ? 0 : cc) * ldxb;
#pragma unroll
                for (int q = 0; q < NP; ++q) xf[i][q] = band_load2(col + roff[q]);
            }
        }
#pragma unroll
        for (int j = 0; j < FD; ++j) {
            const char* col = (const char*)Z + (int64_t)min(j, nc - 1) * ldzb;
#pragma unroll
            for (int q = 0; q < NP; ++q) zin[j][q] = band_load2(col + roff[q]);
        }
    };
    D2 xf[LAGE][NP], zin[FD][NP], xfn[LAGE][NP], zinn[FD][NP];
    if ((int)blockIdx.x < ntiles) request(blockIdx.x, xf, zin);
    bool staged = false;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const unsigned int tbase = (unsigned int)tile * (unsigned int)ROWS + 2u * (unsigned int)tid;
        __builtin_amdgcn_sched_barrier(0);
        if (!staged) {
            // E of grid point i and, next to it, that point's abscissa exactly as the interpolation forms it from the interval
            // number (computed while the loads travel)
            if (tid < T) {
                etab[2 * tid] = band_expq_series(tid == T - 1 ? ylast : (double)tid * ystep + y0);
                etab[2 * tid + 1] = fma((double)(tid + 1), ystep, y0m);
            }
            if (tid < nc) {
                double* slot = tabs + (size_t)tid * tab_slot;
                double scale, bias;
                band_bucket_params(tmin[tid], tmax[tid], nb, scale, bias);
                slot[0] = scale; slot[1] = bias;
                ((int*)slot)[4] = 0; ((int*)slot)[5] = 0;
                slot[3] = 0.0; slot[4] = 0.0; slot[5] = 0.0;
            }
            // (stores unconditional: a slot beyond the last table repeats that table's value - the clamped load - into
            // that table's slot)
            if (tid < Weven) {
#pragma unroll
                for (int u = 0; u < FD; ++u) tabs[(size_t)min(u, nc - 1) * tab_slot + HDR + tid] = tid < T ? tv[u] : INFINITY;
            }
            {
                unsigned short* bs = (unsigned short*)(tabs + (size_t)bc * tab_slot + HDR + Weven) + 4 * bw;
                const uint2 pk = {(unsigned int)(bv.x & 0xffff) | ((unsigned int)bv.y << 16), (unsigned int)(bv.z & 0xffff) | ((unsigned int)bv.w << 16)};
                *(uint2*)bs = pk;
            }
            __syncthreads();
            // the fullest bucket of every table
            if ((tid >> 8) < nc) {
                const unsigned short* bs = (const unsigned short*)(tabs + (size_t)bc * tab_slot + HDR + Weven);
                const int b4 = bw < 255 ? (int)bs[4 * bw + 4] : bv.w;
                int per = max(bv.y - bv.x, max(bv.z - bv.y, bv.w - bv.z));
                per = max(per, b4 - bv.w);                    // (bucket nb does not exist: b4 = its start there, 0 entries)
                for (int o = 32; o > 0; o >>= 1) per = max(per, __shfl_xor(per, o));
                if ((tid & 63) == 0) atomicMax((int*)(tabs + (size_t)bc * tab_slot) + 4, per);
            }
            __syncthreads();
            staged = true;
        }
        const bool more = tile + (int)gridDim.x < ntiles;
        if (more) request(tile + gridDim.x, xfn, zinn);
        __builtin_amdgcn_sched_barrier(0);
#if BAND_FEWI_STAGE == 2                                     /* (timing experiments: results wrong by construction) */
        {
#pragma unroll
            for (int j = 0; j < FD; ++j)
                if (j < nc)
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        const unsigned int n = tbase + (unsigned int)(q * HALF);
                        if (n + 1 < N32) band_store2<BAND_FEW_NT != 0>((char*)X + (int64_t)(kcol0 + j) * ldxb + (size_t)(n * 8u), zin[j][q].x + tabs[tid & 255], zin[j][q].y + etab[tid & 255]);
                    }
            continue;
        }
#endif
        double pend[NS][LAGE];
#pragma unroll
        for (int l = 0; l < LAGE; ++l) {
            const double s0 = P[(int64_t)(k0 + l) * PS + 1];
#pragma unroll
            for (int e = 0; e < NS; ++e) pend[e][l] = s0;
        }
        if (kcol0 > 0) {
#pragma unroll
            for (int i = 0; i < LAGE; ++i) {
                const bool there = kcol0 - LAGE + i >= 0;
                cdbl_p rec = P + (int64_t)(k0 + LAG - LAGE + i) * PS;
                const double start = P[(int64_t)(k0 + i) * PS + 1];
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const double xa = there ? xf[i][q].x : 0.0, xb = there ? xf[i][q].y : 0.0;
                    band_push_e<DB, DA, GP, LAGE, PLAIN>(rec + TTM_P_HDR, start, xa, there ? band_expq_series(xa) : 1.0, pend[2 * q]);
                    band_push_e<DB, DA, GP, LAGE, PLAIN>(rec + TTM_P_HDR, start, xb, there ? band_expq_series(xb) : 1.0, pend[2 * q + 1]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < FD; ++j) {
            if (j < nc) {
                cdbl_p rec = P + (int64_t)(k0 + j + LAG) * PS;
                const double start = P[(int64_t)(k0 + j + LAGE) * PS + 1];
                const double* slot = tabs + (size_t)j * tab_slot;
                const double scale = slot[0], bias = slot[1];
                const int per = __builtin_amdgcn_readfirstlane(((const int*)slot)[4]);
                const int pm = max(per, 1);
                const bool deg = pm + 3 >= T || per > 255;
                const lds_p xsl = band_lds(slot + HDR);
                const lds_p bkl = band_lds(slot + HDR + Weven);
                const double wl = band_lds_f64(xsl + 8 * (deg ? 1 : pm)), wh = band_lds_f64(xsl + 8 * (deg ? 1 : T - 2));
                const int nph = per <= 3 ? 1 : per <= 15 ? 2 : per <= 63 ? 3 : 4;
                double traw[NS], tg[NS], r[NS], E[NS];
                unsigned long long outl = deg ? ~0ull : 0ull;
                int pos[NS];
#pragma unroll
                for (int e = 0; e < NS; ++e) {
                    const double z = (e & 1) ? zin[j][e >> 1].y : zin[j][e >> 1].x;
                    traw[e] = z - pend[e][0];
                    tg[e] = fmin(fmax(traw[e], wl), wh);
                    outl |= __builtin_amdgcn_ballot_w64(traw[e] != tg[e]);
                    const int bi = min((int)fma(tg[e], scale, bias), nb1);
                    pos[e] = (int)*(const __attribute__((address_space(3))) unsigned short*)(bkl + 2 * bi);
                }
                // np.searchsorted(xs, target) (left) = entries in lower buckets + entries of the target's own bucket below it
                // (the bucket function is monotone: k_table_index).  A phase of stride s tests the entries s, 2s, 3s on from
                // pos: every one below the target moves pos on by s; what is left are fewer than s candidates.
                for (int ph = 0, s = 1 << (2 * (nph - 1)); ph < nph; ++ph, s >>= 2) {
                    double qv[NS][3];
#pragma unroll
                    for (int e = 0; e < NS; ++e)
#pragma unroll
                        for (int i = 0; i < 3; ++i) qv[e][i] = band_lds_f64_single(xsl + 8 * min(pos[e] + (i + 1) * s - 1, T + 3));
#pragma unroll
                    for (int e = 0; e < NS; ++e) {
                        int c = 0;
#pragma unroll
                        for (int i = 0; i < 3; ++i) c += qv[e][i] < tg[e] ? 1 : 0;
                        pos[e] += c * s;
                    }
                }
                {
                    D2 ey[NS];
                    double xlo[NS], xhi[NS];
#pragma unroll
                    for (int e = 0; e < NS; ++e) {
                        const int ps1 = band_med3(pos[e], 1, T - 1);          // (a flat start: the first interval, as the search in memory)
                        ey[e] = band_lds_pair(etab1 + 16 * ps1);
                        const lds_p xp = xsl + 8 * ps1;
                        xlo[e] = band_lds_f64(xp - 8);
                        xhi[e] = band_lds_f64_single(xp);
                    }
#if BAND_FEWI_STAGE == 3
#pragma unroll
                    for (int e = 0; e < NS; ++e) { r[e] = xlo[e] + ey[e].y; E[e] = xhi[e] * ey[e].x; }
#else
#pragma unroll
                    for (int e = 0; e < NS; ++e) interp(ey[e].y, xlo[e], xhi[e], ey[e].x, tg[e], r[e], E[e]);
#endif
                }
                if (outl != 0) {
                    // outliers (the tails of the table, NaN; every row of a degenerate table): clip as TM:4074-4076,
                    // np.searchsorted (left) over the whole row in memory, the same interpolation
                    const double* xg = tab_x + (int64_t)j * T;
                    const double lo = tmin[j], hi = tmax[j];
#pragma unroll
                    for (int e = 0; e < NS; ++e) {
                        if (deg || traw[e] != tg[e]) {
                            double t = traw[e];
                            const double cl = fmin(fmax(t, lo), hi);
                            t = t != t ? t : cl;                              // a NaN target stays NaN
                            int a = 0, b = T;
                            while (a < b) {
                                const int mid = (a + b) >> 1;
                                if (xg[mid] < t) a = mid + 1; else b = mid;
                            }
                            const int i = min(max(a, 1), T - 1);
                            interp(fma((double)i, ystep, y0m), xg[i - 1], xg[i], band_expq_series((double)(i - 1) * ystep + y0), t, r[e], E[e]);
                        }
                    }
                }
                // x_k is pushed on to the components that read it, and stored
#pragma unroll
                for (int e = 0; e < NS; ++e) band_push_e<DB, DA, GP, LAGE, PLAIN>(rec + TTM_P_HDR, start, r[e], E[e], pend[e]);
                char* xcol = (char*)X + (int64_t)(kcol0 + j) * ldxb;
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const unsigned int n = tbase + (unsigned int)(q * HALF);
                    char* xp = xcol + (size_t)(n * 8u);
                    if (n + 1 < N32) band_store2<BAND_FEW_NT != 0>(xp, r[2 * q], r[2 * q + 1]);
                    else if (n < N32) *(double*)xp = r[2 * q];
                }
            }
        }
        if (more) {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int i = 0; i < LAGE; ++i) xf[i][q] = xfn[i][q];
#pragma unroll
                for (int jj = 0; jj < FD; ++jj) zin[jj][q] = zinn[jj][q];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// forward map, (optionally) the density terms, and the table inverse of the image in ONE launch - what BASELINE configs[1]
// times as a step ("forward + inverse + pullback" of a map of a few components: two or three launches of ~10 us on 32 MB, each
// a launch floor and a memory round trip of its own).  A tile's columns are read once; z_k stays in registers between the
// forward sweep (k_band_few's arithmetic, statement for statement) and the inverse sweep (k_band_few_inverse's), which writes
// S^-1(S(x)) to Xr.  Same bits as the two (three) launches: tests/test_band.py::test_roundtrip_in_one_launch...
// LDS (doubles): [inverse tables: nc x tab_slot | {E_i, y_i}: 2 x Weven | forward exp table | forward splines: ntab]
// ---------------------------------------------------------------------------
template <int CLS, int LAG, int LAGE, bool PLAIN, bool DENS>
__global__ __launch_bounds__(BAND_CT) void k_band_few_roundtrip(const double* __restrict__ U_, int64_t p_off, int k0, int k1, int kcol0,
                                                                const double* __restrict__ X, int64_t ldx, int64_t N,
                                                                double* __restrict__ Z, int64_t ldz, double* __restrict__ Xr, int64_t ldr,
                                                                double* __restrict__ logdet, const double* __restrict__ sigma,
                                                                double* __restrict__ sumsq, const double* __restrict__ tab_x, int T, double y0,
                                                                double ystep, double ylast, const double* __restrict__ tmin,
                                                                const double* __restrict__ tmax, const int* __restrict__ bkt, int nb, int tab_slot,
                                                                int ntiles, int tab0, int ntab) {
    constexpr int DB = cls_db(CLS), DA = cls_da(CLS), GP = cls_gp(CLS), PS = rec_stride(CLS, LAG);
    constexpr int NS = BAND_FEW_NS, NP = NS / 2, CT = BAND_CT, ROWS = NS * CT, HALF = 2 * CT, FD = TTM_P_FEW_D, HDR = BAND_RT_HDR;
    extern __shared__ __align__(16) double g_lds[];
    const int tid = threadIdx.x;
    const int nc = k1 - k0;
    const int Weven = (T + 4 + 1) & ~1;
    double* itabs = g_lds;                                    // the inverse's tables
    double* ietab = itabs + (size_t)nc * tab_slot;            // {E_i, y_i}
    double* etab = ietab + 2 * Weven;                         // the forward map's exp table
    double* tabs = etab + BAND_ET_DOUBLES;                    // the forward map's splines
    const lds_p etab1 = band_lds(ietab) - 16;
    cdbl_p P = (cdbl_p)(U_ + p_off);
    cdbl_p kt = (cdbl_p)g_band_taylor;
    const unsigned int last_pair = (unsigned int)(((N + 1) & ~(int64_t)1) - 2);
    const unsigned int N32 = (unsigned int)N;
    const int64_t ldxb = ldx * 8, ldzb = ldz * 8, ldrb = ldr * 8;
    const double y0m = y0 - ystep;
    const int nb1 = nb - 1;
    // everything a workgroup stages, requested now
    const D2 ev = *(const D2*)(g_band_etab + 2 * min(tid, TTM_BAND_ET_N - 1));
    D2 sv[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) sv[r] = *(const D2*)(U_ + tab0 + min(2 * tid + r * 2 * CT, ntab - 2));
    double tv[FD];
    {
        const double* src = tab_x + min(tid, T - 1);
#pragma unroll
        for (int u = 0; u < FD; ++u) tv[u] = src[(int64_t)min(u, nc - 1) * T];
    }
    const int bc = min(tid >> 8, nc - 1), bw = tid & 255;
    const int4 bv = *(const int4*)(bkt + (int64_t)bc * (nb + 1) + 4 * bw);
    __builtin_amdgcn_sched_barrier(0);
    auto interp = [&](double y_lo, double x_lo, double x_hi, double e_lo, double tgt, double& rr, double& ee) {
        const double dx = fmax(x_hi - x_lo, 1e-300);
        double rc = __builtin_amdgcn_rcp(dx);
        rc = fma(fma(-dx, rc, 1.0), rc, rc);
        const double delta = (ystep * rc) * (tgt - x_lo);
        rr = delta + y_lo;
        const double w = (delta * -0.25) * (y_lo + rr);
        double p = fma(kt[0], w, kt[1]);
        p = fma(p, w, kt[2]);
        p = fma(p, w, kt[3]);
        p = fma(p, w, kt[4]);
        p = fma(p, w, kt[5]);
        p = fma(p, w, 1.0);
        p = fma(p, w, 1.0);
        ee = e_lo * p;
    };
    double luni = 0.0;
    bool staged = false;
    auto request = [&](int tile, D2 (&xf)[LAGE][NP], D2 (&xin)[FD][NP]) {
        unsigned int roff[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            unsigned int n = (unsigned int)tile * (unsigned int)ROWS + 2u * (unsigned int)tid + (unsigned int)(q * HALF);
            n = n < last_pair ? n : last_pair;
            roff[q] = n * 8u;
        }
        if (kcol0 > 0) {
#pragma unroll
            for (int i = 0; i < LAGE; ++i) {
                const int cc = kcol0 - LAGE + i;
                const char* col = (const char*)X + (int64_t)(cc < 0 ? 0 : cc) * ldxb;
#pragma unroll
                for (int q = 0; q < NP; ++q) xf[i][q] = band_load2(col + roff[q]);
            }
        }
#pragma unroll
        for (int j = 0; j < FD; ++j) {
            const char* col = (const char*)X + (int64_t)(kcol0 + min(j, nc - 1)) * ldxb;
#pragma unroll
            for (int q = 0; q < NP; ++q) xin[j][q] = band_load2(col + roff[q]);
        }
    };
    D2 xf[LAGE][NP], xin[FD][NP];
    if ((int)blockIdx.x < ntiles) request(blockIdx.x, xf, xin);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const unsigned int tbase = (unsigned int)tile * (unsigned int)ROWS + 2u * (unsigned int)tid;
        __builtin_amdgcn_sched_barrier(0);
        if (!staged) {
            if (tid < TTM_BAND_ET_N) *(D2*)(etab + 2 * tid) = ev;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i = 2 * tid + r * 2 * CT;
                if (i < ntab) *(D2*)(tabs + i) = sv[r];
            }
            if (DENS && sigma) {
                for (int k = 0; k < nc; ++k) luni -= band_log(sigma[k]);
            }
            if (tid < T) {
                ietab[2 * tid] = band_expq_series(tid == T - 1 ? ylast : (double)tid * ystep + y0);
                ietab[2 * tid + 1] = fma((double)(tid + 1), ystep, y0m);
            }
            if (tid < nc) {
                double* slot = itabs + (size_t)tid * tab_slot;
                double scale, bias;
                band_bucket_params(tmin[tid], tmax[tid], nb, scale, bias);
                slot[0] = scale; slot[1] = bias;
                ((int*)slot)[4] = 0; ((int*)slot)[5] = 0;
                slot[3] = 0.0; slot[4] = 0.0; slot[5] = 0.0;
            }
            if (tid < Weven) {
#pragma unroll
                for (int u = 0; u < FD; ++u) itabs[(size_t)min(u, nc - 1) * tab_slot + HDR + tid] = tid < T ? tv[u] : INFINITY;
            }
            {
                unsigned short* bs = (unsigned short*)(itabs + (size_t)bc * tab_slot + HDR + Weven) + 4 * bw;
                const uint2 pk = {(unsigned int)(bv.x & 0xffff) | ((unsigned int)bv.y << 16), (unsigned int)(bv.z & 0xffff) | ((unsigned int)bv.w << 16)};
                *(uint2*)bs = pk;
            }
            __syncthreads();
            if ((tid >> 8) < nc) {
                const unsigned short* bs = (const unsigned short*)(itabs + (size_t)bc * tab_slot + HDR + Weven);
                const int b4 = bw < 255 ? (int)bs[4 * bw + 4] : bv.w;
                int per = max(bv.y - bv.x, max(bv.z - bv.y, bv.w - bv.z));
                per = max(per, b4 - bv.w);
                for (int o = 32; o > 0; o >>= 1) per = max(per, __shfl_xor(per, o));
                if ((tid & 63) == 0) atomicMax((int*)(itabs + (size_t)bc * tab_slot) + 4, per);
            }
            __syncthreads();
            staged = true;
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- forward sweep (k_band_few) ----
        double zk[FD][NS];
        {
            double pend[NS][LAGE];
#pragma unroll
            for (int l = 0; l < LAGE; ++l) {
                const double s0 = P[(int64_t)(k0 + l) * PS];
#pragma unroll
                for (int e = 0; e < NS; ++e) pend[e][l] = s0;
            }
            if (kcol0 > 0) {
#pragma unroll
                for (int i = 0; i < LAGE; ++i) {
                    const bool there = kcol0 - LAGE + i >= 0;
                    cdbl_p rec = P + (int64_t)(k0 + LAG - LAGE + i) * PS;
                    const double start = P[(int64_t)(k0 + i) * PS];
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        const double xa = there ? xf[i][q].x : 0.0, xb = there ? xf[i][q].y : 0.0;
                        band_push_e<DB, DA, GP, LAGE, PLAIN>(rec + TTM_P_HDR, start, xa, band_expq(etab, xa, kt), pend[2 * q]);
                        band_push_e<DB, DA, GP, LAGE, PLAIN>(rec + TTM_P_HDR, start, xb, band_expq(etab, xb, kt), pend[2 * q + 1]);
                    }
                }
            }
            double ss[NS], prod[NS], dmin[NS];
            int pexp[NS];
#pragma unroll
            for (int e = 0; e < NS; ++e) { ss[e] = 0.0; prod[e] = 1.0; dmin[e] = 0.0; pexp[e] = 0; }
#pragma unroll
            for (int j = 0; j < FD; ++j) {
                if (DENS && j == 2 && nc > 2) {
#pragma unroll
                    for (int e = 0; e < NS; ++e) { int ex; prod[e] = frexp(prod[e], &ex); pexp[e] = ex; }
                }
                if (j < nc) {
                    cdbl_p rec = P + (int64_t)(k0 + j + LAG) * PS;
                    const double start = P[(int64_t)(k0 + j + LAGE) * PS];
                    const double sp_a = rec[2], sp_b = rec[3], sp_ds = rec[4], own1 = rec[7];
                    cint_p ri = (cint_p)rec;
                    const int nI = ri[10];
                    const double* tab = tabs + (ri[11] - tab0);
                    char* zcol = (char*)Z + (int64_t)j * ldzb;
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        double zv[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int e = 2 * q + h;
                            const double x = h ? xin[j][q].y : xin[j][q].x;
                            double m = 0.0, dm = 0.0;
                            if (nI > 0) {
                                if (DENS) band_spline_d(tab, nI, sp_a, sp_b, sp_ds, x, m, dm);
                                else m = band_spline(tab, nI, sp_a, sp_b, sp_ds, x);
                            }
                            const double E = band_expq(etab, x, kt);
                            zv[h] = fma(own1, x, pend[e][0] + m);
                            zk[j][e] = zv[h];
                            if (DENS) {
                                const double dx = fma(dm, sp_ds, own1);
                                ss[e] = fma(zv[h], zv[h], ss[e]);
                                prod[e] *= dx;
                                dmin[e] = fmin(dmin[e], dx);
                            }
                            band_push_e<DB, DA, GP, LAGE, PLAIN>(rec + TTM_P_HDR, start, x, E, pend[e]);
                        }
                        if (Z) {
                            const unsigned int n = tbase + (unsigned int)(q * HALF);
                            char* zp = zcol + (size_t)(n * 8u);
                            if (n + 1 < N32) band_store2<BAND_FEW_NT != 0>(zp, zv[0], zv[1]);
                            else if (n < N32) *(double*)zp = zv[0];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (DENS) {
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const unsigned int n = tbase + (unsigned int)(q * HALF);
                    if (logdet) {
                        const double la = dmin[2 * q] < 0.0 ? NAN : fma((double)pexp[2 * q], 6.93147180559945286e-01, band_log(prod[2 * q])) + luni;
                        const double lb = dmin[2 * q + 1] < 0.0 ? NAN : fma((double)pexp[2 * q + 1], 6.93147180559945286e-01, band_log(prod[2 * q + 1])) + luni;
                        if (n + 1 < N32) band_store2<false>((char*)(logdet + n), la, lb);
                        else if (n < N32) logdet[n] = la;
                    }
                    if (sumsq) {
                        if (n + 1 < N32) band_store2<false>((char*)(sumsq + n), ss[2 * q], ss[2 * q + 1]);
                        else if (n < N32) sumsq[n] = ss[2 * q];
                    }
                }
            }
        }
        // the columns of the workgroup's next tile travel while this tile is inverted (the conditioning columns of THIS tile are
        // still needed: the next tile's go to a second set)
        const bool more = tile + (int)gridDim.x < ntiles;
        D2 xfc[LAGE][NP];
#pragma unroll
        for (int i = 0; i < LAGE; ++i)
#pragma unroll
            for (int q = 0; q < NP; ++q) xfc[i][q] = xf[i][q];
        if (more && !DENS) request(tile + gridDim.x, xf, xin);       // (the density variant has no registers to spare: after the sweep)
        __builtin_amdgcn_sched_barrier(0);
        // ---- inverse sweep (k_band_few_inverse), targets from registers ----
        {
            double pend[NS][LAGE];
#pragma unroll
            for (int l = 0; l < LAGE; ++l) {
                const double s0 = P[(int64_t)(k0 + l) * PS + 1];
#pragma unroll
                for (int e = 0; e < NS; ++e) pend[e][l] = s0;
            }
            if (kcol0 > 0) {
#pragma unroll
                for (int i = 0; i < LAGE; ++i) {
                    const bool there = kcol0 - LAGE + i >= 0;
                    cdbl_p rec = P + (int64_t)(k0 + LAG - LAGE + i) * PS;
                    const double start = P[(int64_t)(k0 + i) * PS + 1];
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        const double xa = there ? xfc[i][q].x : 0.0, xb = there ? xfc[i][q].y : 0.0;
                        band_push_e<DB, DA, GP, LAGE, PLAIN>(rec + TTM_P_HDR, start, xa, there ? band_expq_series(xa) : 1.0, pend[2 * q]);
                        band_push_e<DB, DA, GP, LAGE, PLAIN>(rec + TTM_P_HDR, start, xb, there ? band_expq_series(xb) : 1.0, pend[2 * q + 1]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < FD; ++j) {
                if (j < nc) {
                    cdbl_p rec = P + (int64_t)(k0 + j + LAG) * PS;
                    const double start = P[(int64_t)(k0 + j + LAGE) * PS + 1];
                    const double* slot = itabs + (size_t)j * tab_slot;
                    const double scale = slot[0], bias = slot[1];
                    const int per = __builtin_amdgcn_readfirstlane(((const int*)slot)[4]);
                    const int pm = max(per, 1);
                    const bool deg = pm + 3 >= T || per > 255;
                    const lds_p xsl = band_lds(slot + HDR);
                    const lds_p bkl = band_lds(slot + HDR + Weven);
                    const double wl = band_lds_f64(xsl + 8 * (deg ? 1 : pm)), wh = band_lds_f64(xsl + 8 * (deg ? 1 : T - 2));
                    const int nph = per <= 3 ? 1 : per <= 15 ? 2 : per <= 63 ? 3 : 4;
                    double traw[NS], tg[NS], r[NS], E[NS];
                    unsigned long long outl = deg ? ~0ull : 0ull;
                    int pos[NS];
#pragma unroll
                    for (int e = 0; e < NS; ++e) {
                        const double z = zk[j][e];
                        traw[e] = z - pend[e][0];
                        tg[e] = fmin(fmax(traw[e], wl), wh);
                        outl |= __builtin_amdgcn_ballot_w64(traw[e] != tg[e]);
                        const int bi = min((int)fma(tg[e], scale, bias), nb1);
                        pos[e] = (int)*(const __attribute__((address_space(3))) unsigned short*)(bkl + 2 * bi);
                    }
                    for (int ph = 0, s = 1 << (2 * (nph - 1)); ph < nph; ++ph, s >>= 2) {
                        double qv[NS][3];
#pragma unroll
                        for (int e = 0; e < NS; ++e)
#pragma unroll
                            for (int i = 0; i < 3; ++i) qv[e][i] = band_lds_f64_single(xsl + 8 * min(pos[e] + (i + 1) * s - 1, T + 3));
#pragma unroll
                        for (int e = 0; e < NS; ++e) {
                            int c = 0;
#pragma unroll
                            for (int i = 0; i < 3; ++i) c += qv[e][i] < tg[e] ? 1 : 0;
                            pos[e] += c * s;
                        }
                    }
                    {
                        D2 ey[NS];
                        double xlo[NS], xhi[NS];
#pragma unroll
                        for (int e = 0; e < NS; ++e) {
                            const int ps1 = band_med3(pos[e], 1, T - 1);
                            ey[e] = band_lds_pair(etab1 + 16 * ps1);
                            const lds_p xp = xsl + 8 * ps1;
                            xlo[e] = band_lds_f64(xp - 8);
                            xhi[e] = band_lds_f64_single(xp);
                        }
#pragma unroll
                        for (int e = 0; e < NS; ++e) interp(ey[e].y, xlo[e], xhi[e], ey[e].x, tg[e], r[e], E[e]);
                    }
                    if (outl != 0) {
                        const double* xg = tab_x + (int64_t)j * T;
                        const double lo = tmin[j], hi = tmax[j];
#pragma unroll
                        for (int e = 0; e < NS; ++e) {
                            if (deg || traw[e] != tg[e]) {
                                double t = traw[e];
                                const double cl = fmin(fmax(t, lo), hi);
                                t = t != t ? t : cl;
                                int a = 0, b = T;
                                while (a < b) {
                                    const int mid = (a + b) >> 1;
                                    if (xg[mid] < t) a = mid + 1; else b = mid;
                                }
                                const int i = min(max(a, 1), T - 1);
                                interp(fma((double)i, ystep, y0m), xg[i - 1], xg[i], band_expq_series((double)(i - 1) * ystep + y0), t, r[e], E[e]);
                            }
                        }
                    }
#pragma unroll
                    for (int e = 0; e < NS; ++e) band_push_e<DB, DA, GP, LAGE, PLAIN>(rec + TTM_P_HDR, start, r[e], E[e], pend[e]);
                    char* xcol = (char*)Xr + (int64_t)(kcol0 + j) * ldrb;
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        const unsigned int n = tbase + (unsigned int)(q * HALF);
                        char* xp = xcol + (size_t)(n * 8u);
                        if (n + 1 < N32) band_store2<BAND_FEW_NT != 0>(xp, r[2 * q], r[2 * q + 1]);
                        else if (n < N32) *(double*)xp = r[2 * q];
                    }
                }
            }
        }
        if (more && DENS) request(tile + gridDim.x, xf, xin);
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static void allow_lds(const void* kern, size_t bytes) {
    static thread_local const void* seen[16];
    static thread_local size_t granted[16];
    static thread_local int n = 0;
    for (int i = 0; i < n; ++i)
        if (seen[i] == kern) {
            if (granted[i] >= bytes) return;
            granted[i] = bytes;
            (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            return;
        }
    (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (n < 16) { seen[n] = kern; granted[n] = bytes; ++n; }
}

// rows of a workgroup's chunk: N / #CUs rounded up to whole 128-byte lines of a column, so that every wave's 1 KB
// accesses are line-aligned (a chunk that starts inside a line makes every wave's store touch nine lines, two of them
// partially: the column stream then runs at 72 % of the copy rate instead of ...: profiles/r03_*)
static int64_t chunk_rows(int64_t N, int cus) {
    static const int align = [] { const char* e = getenv("TTM_BAND_ROWALIGN"); int a = e ? atoi(e) : 32; return a < 2 ? 2 : a; }();
    static const int forced = [] { const char* e = getenv("TTM_BAND_ROWS"); return e ? atoi(e) : 0; }();      // (tuning)
    int64_t rows = (N + cus - 1) / cus;
    rows = (rows + align - 1) / align * align;
    if (forced > 0 && forced >= rows) rows = forced;
    return rows;
}

int record_stride(int cls, int lag) { return rec_stride(cls, lag); }

bool usable(const ttm_program* p, int k0, int k1) {
    return p && p->u_enabled && (p->u_p_lag == 2 || ((p->u_p_lag == 3 || p->u_p_lag == TTM_P_LAG_MAX) && p->D <= TTM_P_FEW_D)) && p->u_h_cls >= 1 &&
           (p->u_h_cls <= 3 || (p->u_h_cls == 4 && p->D <= TTM_P_FEW_D)) &&
           p->h_ucomp && k0 >= 0 && k1 <= p->D && k0 < k1 && p->u_p_stride == rec_stride(p->u_h_cls, p->u_p_lag);
}

// how far back the groups of the components [k0, k1) reach, and whether any of them has plain polynomial terms
static void sweep_shape(const ttm_program* p, int k0, int k1, int* lage, bool* plain) {
    *lage = 1; *plain = false;
    for (int k = k0; k < k1; ++k) {
        const int32_t* uc = p->h_ucomp + k * TTM_UC_LEN;
        for (int g = 0; g < uc[TTM_UC_N_GRP]; ++g) {
            const int32_t* G = p->h_ugrp + (uc[TTM_UC_GRP_OFF] + g) * TTM_UG_LEN;
            const int lag = uc[TTM_UC_KC] - G[TTM_UG_VAR];
            *lage = lag > *lage ? lag : *lage;
            *plain = *plain || (G[TTM_UG_FLAGS] & TTM_UGF_POLY);
        }
    }
}

int build_records(const ttm_program* p, double* U, void* stream) {
    if (!p || !p->u_enabled || p->u_p_lag <= 0 || p->u_h_cls <= 0) return 0;
    hipLaunchKernelGGL(k_band_records, dim3(p->D + p->u_p_lag), dim3(64), 0, (hipStream_t)stream, p->ucomp, p->ugrp, U, (int64_t)p->u_h_off,
                       (int)p->u_h_cls, (int)p->u_h_ng, (int64_t)p->u_p_off, (int)p->u_p_lag, (int)p->u_p_stride, (int)p->D);
    return 0;
}

// components per block so that the splines of a block fit `budget` bytes (0: not even one)
static int plan_blocks(const ttm_program* p, int k0, int k1, size_t budget, int* nblk_out) {
    int worst = 0;
    for (int k = k0; k < k1; ++k) {
        const int b = p->h_ucomp[k * TTM_UC_LEN + TTM_UC_NI] * TTM_U_TSTRIDE * 8;
        worst = b > worst ? b : worst;
    }
    if (worst <= 0 || (size_t)worst > budget) return 0;
    const int ncomp = k1 - k0;
    for (int nblk = 1; nblk <= ncomp; ++nblk) {
        const int Bc = (ncomp + nblk - 1) / nblk;
        bool ok = true;
        for (int kb = k0; kb < k1 && ok; kb += Bc) {
            size_t s = 0;
            for (int k = kb; k < kb + Bc && k < k1; ++k) s += (size_t)p->h_ucomp[k * TTM_UC_LEN + TTM_UC_NI] * TTM_U_TSTRIDE * 8;
            ok = s <= budget;
        }
        if (ok) { *nblk_out = nblk; return Bc; }
    }
    return 0;
}

int forward(const ttm_program* p, const double* U, int k0, int k1, const double* Xsoa, int64_t ldx, int64_t N, double* Zsoa, int64_t ldz,
            double* logdet, const double* sigma, double* sumsq, int cus, size_t lds_per_cu, int block, void* stream, const char** kernel_name) {
    if (!usable(p, k0, k1) || (!Zsoa && !logdet && !sumsq) || N >= ((int64_t)1 << 28)) return 1;
    const bool aligned = ((uintptr_t)Xsoa % 16 == 0) && (ldx % 2 == 0) && ldx >= ((N + 1) & ~(int64_t)1) &&
                         (!Zsoa || ((uintptr_t)Zsoa % 16 == 0 && ldz % 2 == 0)) && ((uintptr_t)U % 16 == 0) &&
                         (!logdet || (uintptr_t)logdet % 16 == 0) && (!sumsq || (uintptr_t)sumsq % 16 == 0);
    if (!aligned) return 1;
    const size_t stat = (logdet || sumsq) ? (size_t)BAND_UNI * 8 : 0;                    // (static array of the density pass)
    if ((logdet || sumsq) && k1 - k0 > BAND_UNI) return 1;
    const size_t fixed = (size_t)BAND_ET_DOUBLES * 8 + stat;
    if (lds_per_cu <= fixed) return 1;
    int nblk = 0;
    int Bc = plan_blocks(p, k0, k1, lds_per_cu - fixed, &nblk);
    if (Bc <= 0) {
        // a sweep without any spline (linear monotone parts: the smoother's block map) of a few components: one block, no tables
        bool any = false;
        for (int k = k0; k < k1; ++k) any = any || p->h_ucomp[k * TTM_UC_LEN + TTM_UC_NI] > 0;
        if (any || k1 - k0 > TTM_P_FEW_D) return 1;
        Bc = k1 - k0;
        nblk = 1;
    }
    if (block > 0 && block < Bc) Bc = block;
    size_t lds = 0;
    for (int kb = k0; kb < k1; kb += Bc) {
        size_t s = 0;
        for (int k = kb; k < kb + Bc && k < k1; ++k) s += (size_t)p->h_ucomp[k * TTM_UC_LEN + TTM_UC_NI] * TTM_U_TSTRIDE * 8;
        lds = s > lds ? s : lds;
    }
    lds += fixed - stat;                                      // (dynamic part)
    // log-determinant only: the derivative of a separable component is a function of its own column alone (k_band_logdet)
    static const int ld_on = [] { const char* e = getenv("TTM_BAND_LOGDET"); return e ? atoi(e) : 1; }();
    if (logdet && !Zsoa && !sumsq && ld_on && k1 - k0 <= BAND_UNI) {
        const size_t lbudget = lds_per_cu - (size_t)BAND_UNI * 8;
        int lblk = 0;
        int LBc = plan_blocks(p, k0, k1, lbudget, &lblk);
        if (LBc <= 0) {                                       // (a range without any spline: one block)
            bool any = false;
            for (int k = k0; k < k1; ++k) any = any || p->h_ucomp[k * TTM_UC_LEN + TTM_UC_NI] > 0;
            if (!any) LBc = k1 - k0;
        }
        if (LBc > 0) {
            if (block > 0 && block < LBc) LBc = block;
            size_t llds = 16;
            for (int kb = k0; kb < k1; kb += LBc) {
                int tb = -1, te = 0;
                for (int k = kb; k < kb + LBc && k < k1; ++k) {
                    const int ni = p->h_ucomp[k * TTM_UC_LEN + TTM_UC_NI], to = p->h_ucomp[k * TTM_UC_LEN + TTM_UC_TAB_OFF];
                    if (ni > 0) { if (tb < 0) tb = to; te = to + TTM_U_TSTRIDE * ni; }
                }
                const size_t sbytes = tb < 0 ? 0 : (size_t)(te - tb) * 8;
                llds = sbytes + 16 > llds ? sbytes + 16 : llds;
            }
            if (llds <= lbudget) {
                typedef void (*lkern_t)(const double*, int64_t, int, int, int, int, const double*, int64_t, int64_t, double*, const double*, int64_t, int);
                int64_t rows = chunk_rows(N, cus);
                const bool wide = rows >= 3 * BAND_CT;
                lkern_t lk = p->u_p_lag == 5 ? (wide ? k_band_logdet<5, 4> : k_band_logdet<5, 2>)
                           : p->u_p_lag == 3 ? (wide ? k_band_logdet<3, 4> : k_band_logdet<3, 2>) : (wide ? k_band_logdet<2, 4> : k_band_logdet<2, 2>);
                const int64_t grid = (N + rows - 1) / rows;
                allow_lds((const void*)lk, llds);
                hipLaunchKernelGGL(lk, dim3((unsigned)grid), dim3(BAND_CT), llds, (hipStream_t)stream, U, (int64_t)p->u_p_off, k0, k1,
                                   (int)p->h_ucomp[k0 * TTM_UC_LEN + TTM_UC_KC], (int)p->u_p_stride, Xsoa, ldx, N, logdet, sigma, rows, LBc);
                if (kernel_name) *kernel_name = "k_band_logdet";
                return 0;
            }
        }
    }
    const int cls = p->u_h_cls;
    // a few components: everything requested at once, tiles of 2048 rows (k_band_few)
    static const int few_on = [] { const char* e = getenv("TTM_BAND_FEW"); return e ? atoi(e) : 1; }();
    if (k1 - k0 <= TTM_P_FEW_D && few_on) {
        // the splines of the sweep as they stand in the U section (padding between them included)
        int tab0 = p->h_ucomp[k0 * TTM_UC_LEN + TTM_UC_TAB_OFF];
        int ntab = p->h_ucomp[(k1 - 1) * TTM_UC_LEN + TTM_UC_TAB_OFF] + TTM_U_TSTRIDE * p->h_ucomp[(k1 - 1) * TTM_UC_LEN + TTM_UC_NI] - tab0;
        if (ntab <= 0) { tab0 = 0; ntab = 2; }                 // (no spline in the sweep - linear monotone parts: two doubles of the U section stand in)
        const size_t flds = (size_t)BAND_ET_DOUBLES * 8 + ((size_t)(ntab > 0 ? ntab : 0) + 2) * 8;
        if (ntab >= 2 && ntab % 2 == 0 && tab0 % 2 == 0 && ntab <= BAND_FEW_TAB && flds <= lds_per_cu && (!sigma || logdet)) {
            typedef void (*fkern_t)(const double*, int64_t, int, int, int, const double*, int64_t, int64_t, double*, int64_t, double*, const double*,
                                    double*, int, int, int);
            const bool dens = logdet || sumsq;
            int lage; bool plain;
            sweep_shape(p, k0, k1, &lage, &plain);
            if (lage > p->u_p_lag) return 1;
            fkern_t fk = nullptr;
#define BAND_FEW_C(L, E, PL, DN) (cls == 1 ? k_band_few<1, L, E, PL, DN> : cls == 2 ? k_band_few<2, L, E, PL, DN> : cls == 3 ? k_band_few<3, L, E, PL, DN> : k_band_few<4, L, E, PL, DN>)
#define BAND_FEW_P(L, E, DN) (plain ? BAND_FEW_C(L, E, true, DN) : BAND_FEW_C(L, E, false, DN))
#define BAND_FEW_E(L, E) (dens ? BAND_FEW_P(L, E, true) : BAND_FEW_P(L, E, false))
            if (p->u_p_lag == 5) {
                fk = BAND_FEW_E(5, 5);                        // (records of five groups - the smoother's block map: one shape, zeros beyond a sweep's reach)
            } else if (p->u_p_lag == 3) {
                fk = lage == 1 ? BAND_FEW_E(3, 1) : lage == 2 ? BAND_FEW_E(3, 2) : BAND_FEW_E(3, 3);
            } else {
                fk = lage == 1 ? BAND_FEW_E(2, 1) : BAND_FEW_E(2, 2);
            }
#undef BAND_FEW_P
#undef BAND_FEW_E
#undef BAND_FEW_C
            const int64_t trows = BAND_FEW_NS * BAND_CT;
            const int64_t ntiles = (N + trows - 1) / trows;
            const int64_t grid = ntiles < cus ? ntiles : cus;
            allow_lds((const void*)fk, flds);
            hipLaunchKernelGGL(fk, dim3((unsigned)grid), dim3(BAND_CT), flds, (hipStream_t)stream, U, (int64_t)p->u_p_off, k0, k1,
                               (int)p->h_ucomp[k0 * TTM_UC_LEN + TTM_UC_KC], Xsoa, ldx, N, Zsoa, ldz, logdet, sigma, sumsq, (int)ntiles, tab0, ntab);
            if (kernel_name) *kernel_name = dens ? "k_band_few<density>" : "k_band_few";
            return 0;
        }
    }
    if (p->u_p_lag != 2 || cls > 3) return 1;             // (lag-3 records / order class 4: the few-component kernels only)
    if (logdet || sumsq) {
        typedef void (*dkern_t)(const double*, int64_t, int, int, int, const double*, int64_t, int64_t, double*, int64_t, double*, const double*,
                                double*, int64_t, int);
        dkern_t dk = Zsoa ? (p->u_h_cls == 1 ? k_band_density<1, 2, true> : p->u_h_cls == 2 ? k_band_density<2, 2, true> : k_band_density<3, 2, true>)
                          : (p->u_h_cls == 1 ? k_band_density<1, 2, false> : p->u_h_cls == 2 ? k_band_density<2, 2, false> : k_band_density<3, 2, false>);
        const int64_t rows = chunk_rows(N, cus);
        const int64_t grid = (N + rows - 1) / rows;
        allow_lds((const void*)dk, lds);
        hipLaunchKernelGGL(dk, dim3((unsigned)grid), dim3(BAND_CT), lds, (hipStream_t)stream, U, (int64_t)p->u_p_off, k0, k1,
                           (int)p->h_ucomp[k0 * TTM_UC_LEN + TTM_UC_KC], Xsoa, ldx, N, Zsoa, ldz, logdet, sigma, sumsq, rows, Bc);
        if (kernel_name) *kernel_name = "k_band_density";
        return 0;
    }
    typedef void (*kern_t)(const double*, int64_t, int, int, int, const double*, int64_t, int64_t, double*, int64_t, int64_t, int);
    kern_t kern = p->u_h_cls == 1 ? k_band_forward<1, 2> : p->u_h_cls == 2 ? k_band_forward<2, 2> : k_band_forward<3, 2>;
    const int64_t rows = chunk_rows(N, cus);
    const int64_t grid = (N + rows - 1) / rows;
    allow_lds((const void*)kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BAND_CT), lds, (hipStream_t)stream, U, (int64_t)p->u_p_off, k0, k1,
                       (int)p->h_ucomp[k0 * TTM_UC_LEN + TTM_UC_KC], Xsoa, ldx, N, Zsoa, ldz, rows, Bc);
    if (kernel_name) *kernel_name = "k_band_forward";
    return 0;
}

// forward (+ density terms) + table inverse of a map of a few components in one launch (k_band_few_roundtrip); 1: not for this map /
// these buffers (the caller makes the separate calls)
int roundtrip(const ttm_program* p, const double* U, int k0, int k1, const double* Xsoa, int64_t ldx, int64_t N, double* Zsoa, int64_t ldz,
              double* Xr, int64_t ldr, double* logdet, const double* sigma, double* sumsq, const double* tab_x, int T, const double* y_affine,
              const double* tmin, const double* tmax, const int32_t* bkt, int nb, int cus, size_t lds_per_cu, bool force, void* stream,
              const char** kernel_name) {
    const int nc = k1 - k0;
    if (!usable(p, k0, k1) || nc > TTM_P_FEW_D || !Xr || !y_affine || T < 64 || T + 4 > BAND_CT || nb + 1 != 1024 || N >= ((int64_t)1 << 28)) return 1;
    if ((uintptr_t)bkt % 16 != 0 || (sigma && !logdet)) return 1;
    const double ymax = fabs(y_affine[0]) > fabs(y_affine[2]) ? fabs(y_affine[0]) : fabs(y_affine[2]);
    if (!(y_affine[1] > 0.0 && y_affine[1] * ymax * 0.5 <= 0.1)) return 1;
    const int64_t need = (N + 1) & ~(int64_t)1;
    const bool aligned = ((uintptr_t)Xsoa % 16 == 0) && (ldx % 2 == 0) && ldx >= need && (!Zsoa || ((uintptr_t)Zsoa % 16 == 0 && ldz % 2 == 0 && ldz >= need)) &&
                         ((uintptr_t)Xr % 16 == 0) && (ldr % 2 == 0) && ldr >= need && ((uintptr_t)U % 16 == 0) &&
                         (!logdet || (uintptr_t)logdet % 16 == 0) && (!sumsq || (uintptr_t)sumsq % 16 == 0);
    if (!aligned) return 1;
    int tab0 = p->h_ucomp[k0 * TTM_UC_LEN + TTM_UC_TAB_OFF];
    int ntab = p->h_ucomp[(k1 - 1) * TTM_UC_LEN + TTM_UC_TAB_OFF] + TTM_U_TSTRIDE * p->h_ucomp[(k1 - 1) * TTM_UC_LEN + TTM_UC_NI] - tab0;
    if (ntab <= 0) { tab0 = 0; ntab = 2; }
    if (ntab < 2 || ntab % 2 != 0 || tab0 % 2 != 0 || ntab > BAND_FEW_TAB) return 1;
    const int Weven = (T + 4 + 1) & ~1;
    const int tab_slot = BAND_RT_HDR + Weven + (((nb + 1 + 3) / 4 + 1) & ~1);
    const size_t lds = ((size_t)nc * tab_slot + (size_t)2 * Weven + (size_t)BAND_ET_DOUBLES + (size_t)ntab + 2) * 8;
    int lage; bool plain;
    sweep_shape(p, k0, k1, &lage, &plain);
    if (lds > lds_per_cu || lage > p->u_p_lag) return 1;
    const int cls = p->u_h_cls;
    const bool dens = logdet || sumsq;
    // Where it pays (measured, graph replay): sweeps that reach one or two columns back, without the density terms - C2b 19.9 us
    // against 10.1 + 11.8 us.  A reach of three columns (C3: 29.8 against 10.5 + 13.1 us) and the density variant (C2b 31.6
    // against 15.9 + 11.8 us) run out of registers (34-117 spilled with the inverse sweep's 128 in use): the launches are bound by the
    // latency of a row's chain of dependent operations, not by the columns' traffic, so one pass instead of two saves a launch
    // floor and a round trip, not the arithmetic.  `force`: tests (every shape through the fused kernel).
    if (!force && (lage > 2 || dens)) return 1;
    typedef void (*rkern_t)(const double*, int64_t, int, int, int, const double*, int64_t, int64_t, double*, int64_t, double*, int64_t, double*,
                            const double*, double*, const double*, int, double, double, double, const double*, const double*, const int*, int, int, int,
                            int, int);
    rkern_t rk = nullptr;
#define BAND_RTF_C(L, E, PL, DN) (cls == 1 ? k_band_few_roundtrip<1, L, E, PL, DN> : cls == 2 ? k_band_few_roundtrip<2, L, E, PL, DN> : cls == 3 ? k_band_few_roundtrip<3, L, E, PL, DN> : k_band_few_roundtrip<4, L, E, PL, DN>)
#define BAND_RTF_P(L, E, DN) (plain ? BAND_RTF_C(L, E, true, DN) : BAND_RTF_C(L, E, false, DN))
#define BAND_RTF_E(L, E) (dens ? BAND_RTF_P(L, E, true) : BAND_RTF_P(L, E, false))
    if (p->u_p_lag == 5) rk = BAND_RTF_E(5, 5);
    else if (p->u_p_lag == 3) rk = lage == 1 ? BAND_RTF_E(3, 1) : lage == 2 ? BAND_RTF_E(3, 2) : BAND_RTF_E(3, 3);
    else rk = lage == 1 ? BAND_RTF_E(2, 1) : BAND_RTF_E(2, 2);
#undef BAND_RTF_E
#undef BAND_RTF_P
#undef BAND_RTF_C
    const int64_t trows = BAND_FEW_NS * BAND_CT;
    const int64_t ntiles = (N + trows - 1) / trows;
    const int64_t grid = ntiles < cus ? ntiles : cus;
    allow_lds((const void*)rk, lds);
    hipLaunchKernelGGL(rk, dim3((unsigned)grid), dim3(BAND_CT), lds, (hipStream_t)stream, U, (int64_t)p->u_p_off, k0, k1,
                       (int)p->h_ucomp[k0 * TTM_UC_LEN + TTM_UC_KC], Xsoa, ldx, N, Zsoa, ldz, Xr, ldr, logdet, sigma, sumsq, tab_x, T, y_affine[0],
                       y_affine[1], y_affine[2], tmin, tmax, bkt, nb, tab_slot, (int)ntiles, tab0, ntab);
    if (kernel_name) *kernel_name = dens ? "k_band_few_roundtrip<density>" : "k_band_few_roundtrip";
    return 0;
}

static double window_fraction() {
    static const double wfrac = [] { const char* e = getenv("TTM_BAND_WFRAC"); return e ? atof(e) : 0.52; }();
    return wfrac;
}

// resident-table images (ttm_band_image.h) of the components [k0, k1) for this table geometry: the plan, or false
bool image_plan(const ttm_program* p, int k0, int k1, int T, int nb, size_t lds_per_cu, int window, int block, int* w0, int* W, int* tab_slot) {
    if (!usable(p, k0, k1) || p->u_p_lag != 2 || k1 - k0 <= TTM_P_FEW_D) return false;
    BandRingPlan pl;
    if (!band_ring_plan(T, nb, k1 - k0, lds_per_cu, window, block, window_fraction(), &pl)) return false;
    *w0 = pl.w0; *W = pl.W; *tab_slot = pl.tab_slot;
    return true;
}

int inverse(const ttm_program* p, const double* U, int k0, int k1, const double* Zsoa, int64_t ldz, double* Xsoa, int64_t ldx, int64_t N,
            const double* tab_x, int T, const double* y_affine, const double* tmin, const double* tmax, const int32_t* bkt, int nb, const double* img,
            int img_doubles, int cus, size_t lds_per_cu, int window, int block, void* stream, const char** kernel_name) {
    if (!usable(p, k0, k1) || !y_affine || T < 64 || T > 4096 || nb + 1 != 1024 || N >= ((int64_t)1 << 28)) return 1;      // (bucket rows copied 16 bytes at a time, 256 units per row)
    if ((uintptr_t)bkt % 16 != 0) return 1;
    const double ymax = fabs(y_affine[0]) > fabs(y_affine[2]) ? fabs(y_affine[0]) : fabs(y_affine[2]);
    if (!(y_affine[1] > 0.0 && y_affine[1] * ymax * 0.5 <= 0.1)) return 1;                 // (exp(w) by its Taylor polynomial)
    const bool aligned = ((uintptr_t)Zsoa % 16 == 0) && (ldz % 2 == 0) && ldz >= ((N + 1) & ~(int64_t)1) && ((uintptr_t)Xsoa % 16 == 0) &&
                         (ldx % 2 == 0) && ldx >= ((N + 1) & ~(int64_t)1);
    if (!aligned) return 1;
    const int ncomp = k1 - k0;
    // a few components: whole tables, everything requested at once, tiles of 2048 rows
    static const int few_on = [] { const char* e = getenv("TTM_BAND_FEW"); return e ? atoi(e) : 1; }();
    if (ncomp <= TTM_P_FEW_D && few_on && T + 4 <= BAND_CT && window <= 0) {          // (no blocks, no windows: `block` does not apply)
        const int Weven = (T + 4 + 1) & ~1;
        const int tab_slot = BAND_RT_HDR + Weven + (((nb + 1 + 3) / 4 + 1) & ~1);
        const size_t lds = ((size_t)ncomp * tab_slot + (size_t)2 * Weven) * 8;
        int lage; bool plain;
        sweep_shape(p, k0, k1, &lage, &plain);
        if (lds <= lds_per_cu && lage <= p->u_p_lag) {
            typedef void (*fkern_t)(const double*, int64_t, int, int, int, const double*, int64_t, double*, int64_t, int64_t, const double*, int,
                                    double, double, double, const double*, const double*, const int*, int, int, int);
            const int cls = p->u_h_cls;
            fkern_t fk = nullptr;
#define BAND_FEWI_C(L, E, PL) (cls == 1 ? k_band_few_inverse<1, L, E, PL> : cls == 2 ? k_band_few_inverse<2, L, E, PL> : cls == 3 ? k_band_few_inverse<3, L, E, PL> : k_band_few_inverse<4, L, E, PL>)
#define BAND_FEWI_E(L, E) (plain ? BAND_FEWI_C(L, E, true) : BAND_FEWI_C(L, E, false))
            if (p->u_p_lag == 5) fk = BAND_FEWI_E(5, 5);
            else if (p->u_p_lag == 3) fk = lage == 1 ? BAND_FEWI_E(3, 1) : lage == 2 ? BAND_FEWI_E(3, 2) : BAND_FEWI_E(3, 3);
            else fk = lage == 1 ? BAND_FEWI_E(2, 1) : BAND_FEWI_E(2, 2);
#undef BAND_FEWI_E
#undef BAND_FEWI_C
            const int64_t ntiles = (N + BAND_FEW_NS * BAND_CT - 1) / (BAND_FEW_NS * BAND_CT);
            const int64_t grid = ntiles < cus ? ntiles : cus;
            allow_lds((const void*)fk, lds);
            hipLaunchKernelGGL(fk, dim3((unsigned)grid), dim3(BAND_CT), lds, (hipStream_t)stream, U, (int64_t)p->u_p_off, k0, k1,
                               (int)p->h_ucomp[k0 * TTM_UC_LEN + TTM_UC_KC], Zsoa, ldz, Xsoa, ldx, N, tab_x, T, y_affine[0], y_affine[1], y_affine[2],
                               tmin, tmax, bkt, nb, tab_slot, (int)ntiles);
            if (kernel_name) *kernel_name = "k_band_few_inverse";
            return 0;
        }
    }
    if (p->u_p_lag != 2 || p->u_h_cls > 3) return 1;       // (lag-3 records / order class 4: the few-component kernels only)
    static const int stagger = [] { const char* e = getenv("TTM_BAND_STAGGER"); return e ? atoi(e) : 1; }();
    const double wfrac = window_fraction();
    // resident-table images at hand (ttm_inverse_table_build_index wrote them): the ring kernel
    BandRingPlan pl;
    if (img && (uintptr_t)img % 16 == 0 && band_ring_plan(T, nb, ncomp, lds_per_cu, window, block, wfrac, &pl) &&
        pl.tab_slot == img_doubles) {                         // (laid out for another plan - options changed in between: not this kernel)
        typedef void (*rkern_t)(const double*, int64_t, int, int, int, const double*, int64_t, double*, int64_t, int64_t, const double*, int, double,
                                double, double, const double*, const double*, const double*, int, int, int, int64_t, int, int);
        const int cls = p->u_h_cls;
#define BAND_RING_K(G) (cls == 1 ? k_band_inverse_ring<1, 2, G> : cls == 2 ? k_band_inverse_ring<2, 2, G> : k_band_inverse_ring<3, 2, G>)
        rkern_t rk = pl.G == 4 ? BAND_RING_K(4) : BAND_RING_K(0);
#undef BAND_RING_K
        const int64_t rows = chunk_rows(N, cus);
        const int64_t grid = (N + rows - 1) / rows;
        allow_lds((const void*)rk, pl.lds);
        hipLaunchKernelGGL(rk, dim3((unsigned)grid), dim3(BAND_CT), pl.lds, (hipStream_t)stream, U, (int64_t)p->u_p_off, k0, k1,
                           (int)p->h_ucomp[k0 * TTM_UC_LEN + TTM_UC_KC], Zsoa, ldz, Xsoa, ldx, N, tab_x, T, y_affine[0], y_affine[1], y_affine[2], tmin,
                           tmax, img, nb, pl.tab_slot, pl.R, rows, pl.w0, pl.W);
        if (kernel_name) *kernel_name = "k_band_inverse_ring";
        return 0;
    }
    int W = T, w0 = 0, Weven = 0, tab_slot = 0, Bc = 0, nblk = 0;
    size_t lds = 0;
    auto plan = [&]() {
        Weven = (W + 4 + 1) & ~1;
        tab_slot = BAND_RT_HDR + Weven + (((nb + 1 + 3) / 4 + 1) & ~1);
        const size_t fixed = (size_t)2 * Weven * 8;
        Bc = 0;
        if (fixed + (size_t)tab_slot * 8 > lds_per_cu) return;
        Bc = (int)((lds_per_cu - fixed) / ((size_t)tab_slot * 8));
        if (Bc > ncomp) Bc = ncomp;
        if (Bc > BAND_RT_KMAX) Bc = BAND_RT_KMAX;
        if (block > 0 && block < Bc) Bc = block;
        nblk = (ncomp + Bc - 1) / Bc;
        if (!stagger) Bc = (ncomp + nblk - 1) / nblk;                     // even blocks (staggered: full blocks + a short one)
        lds = fixed + (size_t)Bc * tab_slot * 8;
    };
    plan();
    // windowed tables when that saves a pass over the chunk (k_inverse_rt)
    if (window != 0 && (window > 0 || (Bc > 0 && nblk > 1))) {
        const int Bfull = Bc, nfull = nblk;
        W = window > 0 ? (window < 16 ? 16 : window) : (int)(wfrac * T);
        if (W >= T) W = T - 1;
        w0 = (T - W) / 2;
        plan();
        if (Bc == 0 || (window < 0 && !(Bfull > 0 && nblk < nfull))) { W = T; w0 = 0; plan(); }
    }
    if (Bc <= 0) return 1;
    typedef void (*kern_t)(const double*, int64_t, int, int, int, const double*, int64_t, double*, int64_t, int64_t, const double*, int, double,
                           double, double, const double*, const double*, const int*, int, int, int, int64_t, int, int);
    const int cls = p->u_h_cls;
    kern_t kern;
    kern = cls == 1 ? k_band_inverse<1, 2, BAND_RT_KMAX> : cls == 2 ? k_band_inverse<2, 2, BAND_RT_KMAX> : k_band_inverse<3, 2, BAND_RT_KMAX>;
    const int64_t rows = chunk_rows(N, cus);
    const int64_t grid = (N + rows - 1) / rows;
    allow_lds((const void*)kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BAND_CT), lds, (hipStream_t)stream, U, (int64_t)p->u_p_off, k0, k1,
                       (int)p->h_ucomp[k0 * TTM_UC_LEN + TTM_UC_KC], Zsoa, ldz, Xsoa, ldx, N, tab_x, T, y_affine[0], y_affine[1], y_affine[2], tmin,
                       tmax, bkt, nb, tab_slot, Bc, rows, w0, W);
    if (kernel_name) *kernel_name = "k_band_inverse";
    return 0;
}

}  // namespace ttm_band
