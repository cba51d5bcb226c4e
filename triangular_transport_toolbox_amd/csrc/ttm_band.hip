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

#include "ttm_band.h"
#include "ttm_band_etab.h"

namespace ttm_band {

typedef const __attribute__((address_space(4))) double* cdbl_p;      // uniform data: scalar loads
typedef const __attribute__((address_space(4))) int* cint_p;
struct __attribute__((aligned(16))) D2 { double x, y; };

__device__ const double g_band_etab[2 * TTM_BAND_ET_N] = { TTM_BAND_ETAB_VALUES };
// Taylor coefficients of exp(w): 1/7! .. 1/2! (scalar operands of the fma chain)
__device__ double g_band_taylor[6] = {1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5};

__host__ __device__ constexpr int cls_db(int cls) { return cls == 1 ? 3 : (cls == 2 ? 5 : 7); }
__host__ __device__ constexpr int cls_da(int cls) { return cls == 1 ? 1 : (cls == 2 ? 5 : 7); }
__host__ __device__ constexpr int cls_gs(int cls) { return cls == 1 ? 8 : (cls == 2 ? 16 : 24); }
__host__ __device__ constexpr int cls_gp(int cls) { return cls_db(cls) + 1 + cls_da(cls); }
__host__ __device__ constexpr int rec_stride(int cls, int lag) { return (TTM_P_HDR + lag * cls_gp(cls) + 7) / 8 * 8; }

#define BAND_ET_DOUBLES (2 * TTM_BAND_ET_N)          /* 12 816 bytes: a multiple of 16 */
#define BAND_CT 1024                                 /* threads per workgroup */
#define BAND_NS 4                                    /* rows per thread: pairs (2t, 2t+1) of the two halves of a tile */

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
template <int DB, int DA, int LAG>
__device__ __forceinline__ void band_push(cdbl_p g, double start, double x, double E, double (&pend)[LAG]) {
    constexpr int GP = DB + 1 + DA;
#pragma unroll
    for (int l = 0; l < LAG; ++l) {
        cdbl_p c = g + l * GP;
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
    D2 xa[NP], xb[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) xa[q] = *(const D2*)(xcol + roff[q]);
    cdbl_p rec = P + (int64_t)(kb + LAG) * PS;
    // one step = one column for the four rows of the thread: xc holds the column (requested a step ahead), xn takes the
    // next one - requested before anything else is done.  Steps are issued in pairs with the two register sets exchanged.
    auto step = [&](int j, const D2 (&xc)[NP], D2 (&xn)[NP]) {
        {
            const char* xnext = j + 1 < ke ? xcol + ldxb : xcol;              // (past the block: a harmless re-read)
#pragma unroll
            for (int q = 0; q < NP; ++q) xn[q] = *(const D2*)(xnext + roff[q]);
        }
        __builtin_amdgcn_sched_barrier(0);                    // (the scheduler would sink the loads to the end of the step)
        // ---- uniform data of the step ------------------------------------------------------------------------
        const double start = rec[0], sp_a = rec[2], sp_b = rec[3], sp_ds = rec[4];
        cint_p ri = (cint_p)rec;
        const int nI = ri[10];
        const double* tab = tabs + (ri[11] - tab0);
        // ---- per row pair (two rows' chains interleave; all four would need 96 registers for the coefficients) -----------
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            double zv[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = 2 * q + h;
                const double x = h ? xc[q].y : xc[q].x;
                const double m = band_spline(tab, nI, sp_a, sp_b, sp_ds, x);
                const double E = band_expq(etab, x, kt);
                zv[h] = pend[e][0] + m;
                band_push<DB, DA, LAG>(rec + TTM_P_HDR, start, x, E, pend[e]);
            }
            const unsigned int n = tbase + (unsigned int)(q * HALF);
            const D2 o = {zv[0], zv[1]};
            char* zp = zcol + (size_t)(n * 8u);
            if (FULL || n + 1 < c1_32) *(D2*)zp = o;
            else if (n < c1_32) *(double*)zp = o.x;
            __builtin_amdgcn_sched_barrier(0);
        }
        rec += PS; xcol += ldxb; zcol += ldzb;
    };
    int j = kb;
    for (; j + 1 < ke; j += 2) {
        step(j, xa, xb);
        step(j + 1, xb, xa);
    }
    if (j < ke) step(j, xa, xb);
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
    for (int i = tid; i < TTM_BAND_ET_N; i += CT) *(D2*)(etab + 2 * i) = *(const D2*)(g_band_etab + 2 * i);
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
            for (int i = 2 * tid; i < n; i += 2 * CT) *(D2*)(tabs + i) = *(const D2*)(U_ + tab0 + i);
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

bool usable(const ttm_program* p, int k0, int k1) {
    return p && p->u_enabled && p->u_p_lag == 2 && p->u_h_cls >= 1 && p->u_h_cls <= 3 && p->h_ucomp && k0 >= 0 && k1 <= p->D && k0 < k1 &&
           p->u_p_stride == rec_stride(p->u_h_cls, p->u_p_lag);
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
            double* logdet, const double* sigma, double* sumsq, int cus, size_t lds_per_cu, void* stream, const char** kernel_name) {
    (void)sigma;
    if (!usable(p, k0, k1) || !Zsoa || logdet || sumsq || N >= ((int64_t)1 << 28)) return 1;
    const bool aligned = ((uintptr_t)Xsoa % 16 == 0) && (ldx % 2 == 0) && ldx >= ((N + 1) & ~(int64_t)1) && ((uintptr_t)Zsoa % 16 == 0) &&
                         (ldz % 2 == 0) && ((uintptr_t)U % 16 == 0);
    if (!aligned) return 1;
    const size_t fixed = (size_t)BAND_ET_DOUBLES * 8;
    if (lds_per_cu <= fixed) return 1;
    int nblk = 0;
    const int Bc = plan_blocks(p, k0, k1, lds_per_cu - fixed, &nblk);
    if (Bc <= 0) return 1;
    size_t lds = 0;
    for (int kb = k0; kb < k1; kb += Bc) {
        size_t s = 0;
        for (int k = kb; k < kb + Bc && k < k1; ++k) s += (size_t)p->h_ucomp[k * TTM_UC_LEN + TTM_UC_NI] * TTM_U_TSTRIDE * 8;
        lds = s > lds ? s : lds;
    }
    lds += fixed;
    typedef void (*kern_t)(const double*, int64_t, int, int, int, const double*, int64_t, int64_t, double*, int64_t, int64_t, int);
    kern_t kern = p->u_h_cls == 1 ? k_band_forward<1, 2> : p->u_h_cls == 2 ? k_band_forward<2, 2> : k_band_forward<3, 2>;
    int64_t rows = (N + cus - 1) / cus;
    rows = (rows + 1) & ~(int64_t)1;
    const int64_t grid = (N + rows - 1) / rows;
    allow_lds((const void*)kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BAND_CT), lds, (hipStream_t)stream, U, (int64_t)p->u_p_off, k0, k1,
                       (int)p->h_ucomp[k0 * TTM_UC_LEN + TTM_UC_KC], Xsoa, ldx, N, Zsoa, ldz, rows, Bc);
    if (kernel_name) *kernel_name = "k_band_forward";
    return 0;
}

int inverse(const ttm_program*, const double*, int, int, const double*, int64_t, double*, int64_t, int64_t, const double*, int, const double*,
            const double*, const double*, const int32_t*, int, int, size_t, int, int, void*, const char**) {
    return 1;
}

}  // namespace ttm_band
