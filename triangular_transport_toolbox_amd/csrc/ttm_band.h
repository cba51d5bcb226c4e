// ttm_band.h - internal interface of csrc/ttm_band.hip (the kernels of banded U-form maps) to csrc/ttm_kernels.hip.
//
// A BANDED map (BASELINE config 5, Markov-type maps): the columns of the components are consecutive and every
// nonmonotone group of component k reads a column kc-1 .. kc-LAG (LAG <= TTM_P_LAG_MAX), every component has a
// special-term spline and no polynomial terms of its own variable.  Such a map is evaluated in PUSH form: walking the
// columns in order, the value x_c of a column is used at once for everything that depends on it -
//
//     S at column c          = pend[0] + G_c(x_c)                      (G_c: the component's spline)
//     pend[l], l = 0..LAG-2  = pend[l+1] + f_{c+l+1, c}(x_c)           (contribution to the component l+1 columns on)
//     pend[LAG-1]            = c0_{c+LAG} + f_{c+LAG, c}(x_c)
//
// so a row carries LAG running sums instead of LAG cached columns with their exp(-x^2/4), and one step is straight-line
// code.  The push records ("P section" of the U section, include/ttm.h) hold, per column, the coefficients of the
// groups that READ the column; ttm_fold builds them behind the hot records.
#pragma once

#include <stdint.h>

#include "../../include/ttm.h"

namespace ttm_band {

// build the P section from the H section (device, on `stream`); also writes the local-coordinate offset of every spline
// column into its padding slot.  U: the U section of the folded-coefficient buffer.
int build_records(const ttm_program* p, double* U, void* stream);

// doubles per push record of degree class `cls` with `lag` groups (what u_p_stride must be)
int record_stride(int cls, int lag);

// can [k0, k1) of this program run through the band kernels?
bool usable(const ttm_program* p, int k0, int k1);

// forward map of the components [k0, k1): Z and / or the fused log-determinant and sum of squares (`block` > 0: at most that
// many components' splines resident at a time)
int forward(const ttm_program* p, const double* U, int k0, int k1, const double* Xsoa, int64_t ldx, int64_t N, double* Zsoa,
            int64_t ldz, double* logdet, const double* sigma, double* sumsq, int cus, size_t lds_per_cu, int block, void* stream, const char** kernel_name);

// table inverse (resident windowed tables, as k_inverse_rt) in push form
// img: the resident-table images of the components (image_plan; written by k_table_build_index) or nullptr
int inverse(const ttm_program* p, const double* U, int k0, int k1, const double* Zsoa, int64_t ldz, double* Xsoa, int64_t ldx,
            int64_t N, const double* tab_x, int T, const double* y_affine, const double* tmin, const double* tmax, const int32_t* bkt,
            int nb, const double* img, int img_doubles, int cus, size_t lds_per_cu, int window, int block, void* stream, const char** kernel_name);

// forward map (+ log-determinant / sum of squares) and the table inverse of the image, in ONE launch, for maps of a few components
// (k_band_few_roundtrip): Z (nullable) = S(X), Xr = S^-1(S(X)) - conditioning columns are read from X.  Without `force` only the
// shapes the one launch is faster for (reach <= 2 columns, no density terms); 1: declined
int roundtrip(const ttm_program* p, const double* U, int k0, int k1, const double* Xsoa, int64_t ldx, int64_t N, double* Zsoa, int64_t ldz,
              double* Xr, int64_t ldr, double* logdet, const double* sigma, double* sumsq, const double* tab_x, int T, const double* y_affine,
              const double* tmin, const double* tmax, const int32_t* bkt, int nb, int cus, size_t lds_per_cu, bool force, void* stream,
              const char** kernel_name);

// does the table inverse of [k0, k1) take resident-table images (csrc/ttm_band_image.h) for this table geometry?  The window
// [w0, w0 + W) of every table and the doubles per image (the table-building kernel writes them, `inverse` reads them)
bool image_plan(const ttm_program* p, int k0, int k1, int T, int nb, size_t lds_per_cu, int window, int block, int* w0, int* W, int* tab_slot);

}  // namespace ttm_band
