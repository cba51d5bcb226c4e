// ttm_band_image.h - the resident-table IMAGE of a component's inverse table: the bytes k_band_inverse keeps in LDS per
// component (search parameters, the windowed table, the 16-bit bucket index), laid out once per coefficient vector by the
// kernel that builds the table (k_table_build_index, csrc/ttm_kernels.hip) so that the lookup kernel can copy them into LDS
// by DMA (k_band_inverse_ring, csrc/ttm_band.hip) instead of assembling them per workgroup and block: a block switch of
// k_band_inverse is a string of dependent round trips (header loads, two rounds of table loads, the per-bucket scan:
// 5.9 us of a 143 us launch at C5), the DMA of a group of images is issued a group of columns ahead and costs a barrier.
//
// image of one component, `tab_slot` doubles (a whole number of 16-byte units):
//   [wl, wh, bucket scale, bucket bias, int32 {entries per bucket at most, 0}, int32 {degenerate, 0} |
//    xs: the W table entries of the window [w0, w0 + W) + sentinels (+inf) up to Weven | bucket index: nb + 1 uint16, zero padded]
// exactly what the block prologue of k_band_inverse leaves in a table slot (the same values, the same bits).
#pragma once

#include <math.h>
#include <stddef.h>
#include <stdint.h>

#define BAND_RT_HDR 6
#define BAND_RING_KMAX 24                             /* slots of the ring, at most */

struct BandRingPlan {
    int W, w0, Weven, tab_slot;                       // window [w0, w0 + W) of every table, doubles per image
    int R, G;                                         // ring of R slots, refilled G at a time (G = 0: every component resident, no refill)
    size_t lds;                                       // dynamic LDS of the launch
};

// The plan is a function of the table geometry, the number of components and the LDS of a CU alone: the kernel that writes
// the images and the kernel that reads them compute it separately and agree.  false: no ring kernel for this geometry.
static inline bool band_ring_plan(int T, int nb, int ncomp, size_t lds_per_cu, int window, int block, double wfrac, BandRingPlan* pl) {
    if (T < 64 || T > 4096 || nb + 1 != 1024 || ncomp < 1) return false;
    auto fit = [&](int W) {                           // slots that fit next to the {E_i, y_i} pairs of the window
        const int Weven = (W + 4 + 1) & ~1;
        const int tab_slot = BAND_RT_HDR + Weven + (((nb + 1 + 3) / 4 + 1) & ~1);
        const size_t fixed = (size_t)2 * Weven * 8;
        if (fixed + (size_t)tab_slot * 8 > lds_per_cu) return 0;
        int B = (int)((lds_per_cu - fixed) / ((size_t)tab_slot * 8));
        if (B > BAND_RING_KMAX) B = BAND_RING_KMAX;
        if (block > 0 && block < B) B = block;
        return B;
    };
    int W = T, w0 = 0;
    int B = fit(W);
    if (window > 0 || (window != 0 && B < ncomp)) {   // windowed tables when whole ones would need refills (or on request)
        int Ww = window > 0 ? (window < 16 ? 16 : window) : (int)(wfrac * T);
        if (Ww >= T) Ww = T - 1;
        Ww &= ~1;                                     // (even: the doubles per image then determine the window - what a lookup checks)
        const int Bw = fit(Ww);
        if (window > 0 || Bw > B) { W = Ww; w0 = (T - Ww) / 2; B = Bw; }
    }
    if (B <= 0) return false;
    pl->W = W; pl->w0 = w0;
    pl->Weven = (W + 4 + 1) & ~1;
    pl->tab_slot = BAND_RT_HDR + pl->Weven + (((nb + 1 + 3) / 4 + 1) & ~1);
    if (pl->tab_slot % 2) return false;               // (16-byte units)
    if (ncomp <= B) { pl->R = ncomp; pl->G = 0; }
    else {
        const int G = 4;                              // (refills of 8 measured 2 % slower at C5: OPTLOG round 4, item 22)
        const int R = B / G * G;
        if (R < 3 * G) return false;                  // (a refill is certified one group after it is issued and used two groups on)
        pl->R = R; pl->G = G;
    }
    pl->lds = (size_t)2 * pl->Weven * 8 + (size_t)pl->R * pl->tab_slot * 8;
    return true;
}

#if defined(__HIPCC__)
__device__ __forceinline__ void band_bucket_params(double lo, double hi, int nb, double& scale, double& bias) {
    scale = (double)nb / (hi - lo);                                           // (k_table_index: the same IEEE division)
    if (!(scale > 0.0 && scale < 1.0e300)) scale = 0.0;
    bias = -lo * scale;
}

// One workgroup writes the image of its component: xs (LDS, the T sorted table entries), bks (LDS, nb + 1 bucket starts as
// k_table_index computes them), red (LDS, one int, any value).  Every thread of the workgroup calls it.
__device__ inline void band_image_write(const double* xs, const int* bks, int* red, int T, int nb, double lo, double hi, int w0, int W,
                                        int Weven, int tab_slot, double* __restrict__ img) {
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) *red = 0;
    __syncthreads();
    // entries per bucket, at most, over the buckets that reach into the window
    int per = 0;
    for (int b = tid; b < nb; b += nt) {
        const int b0 = bks[b] & 0xffff, b1 = bks[b + 1] & 0xffff;
        if (b1 > w0 && b0 < w0 + W) per = b1 - b0 > per ? b1 - b0 : per;
    }
    if (per > 0) atomicMax(red, per);
    __syncthreads();
    per = *red;
    const int pm = per > 1 ? per : 1;
    const bool deg = pm + 3 >= W || per > 4;          // (more entries in a bucket than the resident search compares: every row by the outlier path)
    auto xw = [&](int i) { return i < W ? xs[w0 + i] : (double)INFINITY; };
    if (tid == 0) {
        double scale, bias;
        band_bucket_params(lo, hi, nb, scale, bias);
        img[0] = deg ? xw(1) : xw(pm);
        img[1] = deg ? xw(1) : xw(W - 2);
        img[2] = scale; img[3] = bias;
        int* ih = (int*)(img + 4);
        ih[0] = per; ih[1] = 0; ih[2] = deg ? 1 : 0; ih[3] = 0;
    }
    for (int i = tid; i < Weven; i += nt) img[BAND_RT_HDR + i] = xw(i);
    // bucket index: four uint16 per double, zero behind entry nb
    const int nbd = tab_slot - BAND_RT_HDR - Weven;
    for (int i = tid; i < nbd; i += nt) {
        unsigned long long pk = 0;
        for (int j = 0; j < 4; ++j) {
            const int q = 4 * i + j;
            const unsigned long long v = q <= nb ? (deg ? (unsigned)(w0 + 1) : (unsigned)bks[q]) & 0xffffu : 0u;
            pk |= v << (16 * j);
        }
        img[BAND_RT_HDR + Weven + i] = __longlong_as_double((long long)pk);
    }
}
#endif
