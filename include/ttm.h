/*
 * ttm.h - C ABI of the MI355X-native triangular-transport-map engine (libttm.so).
 *
 * The reference (MaxRamgraber/Triangular-Transport-Toolbox, transport_map.py,
 * "TM" below) has no FFI / plugin boundary: its hot path is NumPy code behind
 * the Python class `transport_map`.  This header is the boundary the build adds
 * beneath that class.  Every entry point names the reference routine whose
 * per-sample arithmetic it replaces (file:line into /root/reference).
 * `triangular_transport_toolbox_amd/_capi.py` binds it with ctypes; the same
 * stub is what a maintainer of the reference would add.
 *
 * Conventions
 *  - plain C types only; all array arguments are DEVICE pointers unless the
 *    name starts with h_ (host); the caller owns every buffer (PyTorch-ROCm is
 *    the allocator in the Python host, any hipMalloc'ed memory works);
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); no entry
 *    point synchronises, allocates or frees: all are graph-capturable;
 *  - every function returns 0 on success, a negative TTM_E_* code otherwise;
 *    ttm_last_error_string() describes the last failure of the calling thread;
 *    no exception crosses the boundary;
 *  - sample matrices are column-major on the device ("SoA": column j of an
 *    N-sample ensemble starts at X + j*ldx, ldx >= N), fp64 throughout;
 *  - handles do not exist: a map is described by a `ttm_program` value (plain
 *    struct of sizes, small host tables and device table pointers) compiled by
 *    the host from the reference's `monotone` / `nonmonotone` lists.
 */
#ifndef TTM_H
#define TTM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TTM_VERSION 100

/* error codes */
#define TTM_OK            0
#define TTM_E_ARG        -1   /* invalid argument (shape, range, null pointer) */
#define TTM_E_HIP        -2   /* a HIP runtime call failed */
#define TTM_E_LIMIT      -3   /* program does not fit the kernel's LDS budget */
#define TTM_E_UNSUPPORTED -4

/* factor kinds (term tables) */
#define TTM_KIND_POLY 1      /* P_n(x), polynomial family of the program            (TM:1099-1144) */
#define TTM_KIND_HF   2      /* a_n P_n(x) exp(-x^2/4), normalised Hermite function (TM:1102-1150) */
#define TTM_KIND_LET  3      /* left edge term                                      (TM:917-937)   */
#define TTM_KIND_RET  4      /* right edge term                                     (TM:943-963)   */
#define TTM_KIND_RBF  5      /* radial basis function                               (TM:969-989)   */
#define TTM_KIND_IRBF 6      /* integrated radial basis function                    (TM:995-1016)  */

/* polynomial families (TM:274-304) */
#define TTM_FAM_HERMITE_E 0
#define TTM_FAM_POWER     1
#define TTM_FAM_HERMITE   2
#define TTM_FAM_CHEBYSHEV 3
#define TTM_FAM_LAGUERRE  4
#define TTM_FAM_LEGENDRE  5

/* monotonicity (TM:260-263) */
#define TTM_MONO_INTEGRATED 0
#define TTM_MONO_SEPARABLE  1

/* rectifiers (TM:4981-5018) */
#define TTM_RECT_EXPONENTIAL 0
#define TTM_RECT_SOFTPLUS    1
#define TTM_RECT_SQUARED     2
#define TTM_RECT_EXPNEG      3
#define TTM_RECT_ELU         4

/* int32 layout of one component block inside ttm_program.itab
 * (block k spans itab[h_comp_off[k] .. h_comp_off[k+1]) ):
 *   header  : TTM_HDR_LEN int32
 *   terms   : 4 int32 each  {f0, nf, b, ci}         nonmonotone terms, then monotone terms
 *   factors : 4 int32 each  {var, kind, order, p0}   factors on columns other than kc
 *   bfuns   : 4 int32 each  {kind, order, p0, 0}     distinct functions of x_kc alone, ordered
 *                                                    [HF by order][POLY by order][special terms]
 *   groups  : 4 int32 each  {var, P, off, has_hf}    univariate polynomial/HF nonmonotone terms,
 *                                                    grouped per variable: folded[off .. off+P) are the
 *                                                    summed coefficients of P_1..P_P, the next P those of
 *                                                    a_n P_n exp(-x^2/4)
 *   gen     : 1 int32 each                           indices of the nonmonotone terms NOT covered by
 *                                                    groups / the constant (cross terms, special terms)
 *   mnt     : 1 int32 each                           indices of monotone terms with factors on other columns
 * and, in the separate table ttm_program.ftab (block k at ftab[h_ftab_off[k] ..), read once per launch:
 *   fslots  : 2 int32 each  {src0, nsrc}             recipe of each folded coefficient
 *   fsrc    : 2 int32 each  {ci, p0}                 coefficient index within [nonmon | mon] and an
 *                                                    optional dpar multiplier (-1: none); ci = -1: the
 *                                                    source is the constant dpar[p0] itself
 * f0 indexes the component's factor list, b its bfun list (-1: term has no x_kc factor), ci the
 * coefficient inside coeffs_nonmon[k] / coeffs_mon[k], p0 the component's slice of dpar
 * (HF: a_n ; special term: centre, scale, 1/(sqrt2 scale), scale sqrt(2/pi), 1/(sqrt(2 pi) scale)).
 * Folded coefficients (recomputed from the coefficient vector when a kernel stages the program):
 *   [0] sum of constant nonmonotone coefficients | group arrays | wB[0..nB] = summed coefficients of the
 *   monotone terms that are functions of x_kc alone, per B function (slot nB: terms without x_kc). */
#define TTM_HDR_LEN      32
#define TTM_HDR_KC        0   /* column of x_k in the sample matrix (k + skip_dimensions) */
#define TTM_HDR_N_NM      1
#define TTM_HDR_OFF_NM    2
#define TTM_HDR_N_MON     3
#define TTM_HDR_OFF_MON   4
#define TTM_HDR_OFF_FAC   5
#define TTM_HDR_NB        6
#define TTM_HDR_OFF_B     7
#define TTM_HDR_NB_HF     8
#define TTM_HDR_NB_POLY   9
#define TTM_HDR_NB_ST    10
#define TTM_HDR_MAXP_HF  11
#define TTM_HDR_MAXP_POLY 12
#define TTM_HDR_FLAGS    13   /* bit0: every monotone term is a function of x_kc alone */
#define TTM_HDR_N_DPAR   14   /* doubles of dpar owned by this component                      */
#define TTM_HDR_LEN_BLK  15   /* int32 length of this component block (header included)      */
#define TTM_HDR_N_GRP    16
#define TTM_HDR_OFF_GRP  17
#define TTM_HDR_N_GEN    18
#define TTM_HDR_OFF_GEN  19
#define TTM_HDR_N_MNT    20
#define TTM_HDR_OFF_MNT  21
#define TTM_HDR_N_FOLD   22   /* doubles of folded coefficients of this component             */
#define TTM_HDR_OFF_FSLOT 23
#define TTM_HDR_OFF_FSRC 24
#define TTM_HDR_OFF_WB   25   /* offset of wB inside the folded array                          */
#define TTM_HDR_N_XGRP   26   /* cross groups: weights of B functions as folded series in ONE conditioning variable */
#define TTM_HDR_OFF_XGRP 27   /* records of 8 int32: {var, P, fold offset, has_hf, b, 0, 0, 0}                      */
#define TTM_HDR_OFF_XPROG 28  /* offset (within the block) of the component's X program; 0: it has none              */
#define TTM_HDR_X_NROW   29   /* X program: row columns in front of the q columns (3 + U values + computed products)  */
#define TTM_HDR_X_NSUM   30   /* X program: sums of an objective + gradient evaluation (<= TTM_X_SUM_MAX)              */
/* ---- "X program" of an integrated-rectifier component (csrc/ttm_xprog.h; round 5) -------------------------------------
 * Present (flag bit 4 of h_complex) when the functions of x_kc in the monotone list are polynomials / Hermite functions only
 * (flag bit 2) and EVERY factor on another column, of every nonmonotone and monotone term, is a polynomial or a Hermite
 * function of the map's family of an order within the component's order class, at most TTM_X_MAXF factors per term.
 * The kernels then evaluate each distinct factor value ("U value": P_n(x_j) or a_n P_n(x_j) exp(-x_j^2/4)) ONCE per sample,
 * each distinct product of them ("A value") once, and get the monomial form of g and the nonmonotone sum from matrices
 * that are functions of the coefficients alone - part of the FOLD (computed by the fold recipe like every folded sum):
 *     h_j = sum_a A_a HC[a][j],   a_j = sum_a A_a HP[a][j]   (g(t) = exp(-t^2/4) sum_j h_j t^j + sum_j a_j t^j),
 *     Psi_nonmon c_nonmon = sum_a A'_a CN[a].
 * A sample owns a ROW of doubles: [0] = 1, [1] = S_k, [2] = 1/2 S_k^2 - log(r + delta), [3 ..] the U values, then the
 * computed products (X_NROW columns so far), then - objective kernel only - the q columns
 *     qH_j = S x_k/2 mh_j - r'/(r + delta) E(x_k) x_k^j (j <= PH),   qP_j = S x_k/2 mp_j - r'/(r + delta) x_k^j (j <= PP)
 * (mh, mp: the node moments of csrc/ttm_dense.h), so that EVERY sum of an evaluation is a product of two row entries summed
 * over the samples: sum 0 = [0] [2] (J); 1 + a = A'_a [1] (gradient of the nonmonotone coefficients sharing product a);
 * 1 + NA_NM + a NQ + j = A_a q_j - the gradient of monotone coefficient i is sum_j CB[b(i)][j] of the sums of its product,
 * CB = the basis-conversion row of its B function (a_n folded in), applied once per evaluation by the finishing workgroup.
 * Layout (int32): header of 8 {NVAR, NU, NPROD, NA_NM, NA_MON, offset of the fold's X section within the component's fold,
 * offset of the U rows within the component's dpar slice, PH | PP << 8 (the component's own order class)}, then
 *   vars  : NVAR x 4 {column, plain-polynomial U values on it, Hermite-function U values on it, 0}
 *   prods : NPROD x 4 {c1, c2, c3, 0}: row columns of the factors (0 = the constant 1); product p lives in column 3 + NU + p
 *   anm   : NA_NM row columns (padded to a multiple of 4): the distinct products of the nonmonotone terms
 *   amon  : NA_MON row columns (padded): the distinct products A of the monotone terms
 *   terms : N_NM + N_MON x 4 {index into anm / amon, B function b (nB: none; -1 for a nonmonotone term), 0, 0}
 * dpar, at the U-row offset: (TTM_I_PMAX + 1) monomial coefficients per U value (a_n folded in), U values ordered by column,
 * plain polynomials before Hermite functions, by order.  Fold X section: CN[NA_NM] | HC[NA_MON][TTM_I_PMAX + 1] |
 * HP[NA_MON][TTM_I_PMAX + 1].                                                                                          */
#define TTM_X_MAXF        3
#define TTM_X_NU_MAX     40
#define TTM_X_SUM_MAX   192
#define TTM_XR_ONE        0
#define TTM_XR_S          1
#define TTM_XR_J          2
#define TTM_XR_U          3
#define TTM_XH_LEN        8
#define TTM_ST_NPAR       5   /* dpar doubles per special term                                 */
/* fast-path descriptor of a component (components whose terms are all univariate: no table walking,
 * every record is at a known offset so the scalar loads can be issued ahead of use) */
#define TTM_FDESC_LEN    16
#define TTM_FD_KC         0
#define TTM_FD_N_GRP      1
#define TTM_FD_N_ST       2
#define TTM_FD_MAXP_HF    3
#define TTM_FD_MAXP_POLY  4
#define TTM_FD_COMPLEX    5   /* 1: has cross / generic terms -> generic interpreter              */
#define TTM_FD_FINT_OFF   6   /* offset into fints                                             */
#define TTM_FD_FOLD_OFF   7   /* offset of the component's folded coefficients                 */
#define TTM_FD_STREAM     8   /* offset (within the component's fold) of the stream section:
                                 wHF[maxP_hf] | wPoly[maxP_poly] | per special term {w, centre,
                                 1/(sqrt2 scale), scale sqrt(2/pi), 1/(sqrt(2 pi) scale)}     */
#define TTM_FD_NB         9
#define TTM_FD_OFF_WB    10
#define TTM_FD_ST8       11   /* offset (within the component's fold) of the unified special-term section written
                                 by the second stage of ttm_fold: 8 doubles {g0 = w_none + sum A0, 7 pad}, then
                                 one 8-double record per special term {centre, 1/(sqrt2 scale), A1, B0, B1, G,
                                 DG, DT}: value += A1 erf + d (B0 + B1 erf) + G gauss, d/dx += (B0 + B1 erf) +
                                 (DG + DT t) gauss with d = x - centre, t = d/(sqrt2 scale), gauss = exp(-t^2);
                                 records that need the Gaussian for their VALUE (LET/RET/RBF) come first    */
#define TTM_FD_N_STA     12   /* how many of the N_ST records need the Gaussian for the value            */
#define TTM_FD_KC_SLOT   13   /* planned column cache: slot that keeps x_kc for later components (-1: none) */
#define TTM_FD_PLAN_OFF  14   /* offset into fints of the TTM_PLAN_WAYS entry-state words of the planned
                                 cache (what the cache holds when a sweep reaches this component: column |
                                 TTM_PLAN_E if exp(-x^2/4) is held too; -1: slot empty)                   */
/* planned column cache (fast kernels): the access sequence of a sweep over the components is static, so
 * hit / miss / replacement (Belady) decisions are made when the program is compiled; a group record's
 * flag word says where its column lives */
#define TTM_PLAN_WAYS     4   /* at most; ttm_program.plan_ways are used (slot w: x at 2w, exp(-x^2/4) at 2w+1) */
#define TTM_PLAN_HF       1   /* group has Hermite-function terms (needs exp(-x^2/4))           */
#define TTM_PLAN_XHIT     2   /* column value is in the slot                                   */
#define TTM_PLAN_EHIT     4   /* exp(-x^2/4) is in the slot too                                */
#define TTM_PLAN_SLOT(f) (((f) >> 8) & 255)     /* 255: bypass (do not keep)                   */
#define TTM_PLAN_E       (1 << 30)

/* ---- univariate form ("U-form") of a separable map whose terms are all univariate -------------------------
 * (csrc/ttm_uform.h).  Static structure in two int32 tables compiled by the host, coefficients in a section
 * appended to the folded-coefficient buffer by ttm_fold:
 *   ucomp : TTM_UC_LEN int32 per component
 *   ugrp  : TTM_UG_LEN int32 per group; a component's nonmonotone groups first, then (TTM_UCF_OWN) one group for
 *           the polynomial / Hermite-function terms of its monotone list
 *   U section (doubles, at fold + u_base): per component, at DBL_OFF: {c0, -t_lo/h, 1/h, 2/h} and TTM_U_GSTRIDE doubles per
 *           group {B[0..11] monomial coefficients of the exp(-x^2/4) part, A[0..11] of the plain part}; at TAB_OFF the
 *           spline of the summed special terms: NI columns of TTM_U_TSTRIDE doubles (12 coefficients in the local
 *           coordinate s in [-1,1] of the interval + padding), column 0 / NI-1 = the linear tails; then 2 doubles per
 *           component {fit error of the value, of the derivative} at u_err_off.                                   */
#define TTM_U_PMAX       10   /* largest polynomial order a group may have (round 5: 7 -> 10, example_03.py:103)          */
#define TTM_U_GHALF      12   /* doubles per coefficient set of a group in the U section: B[0..11] then A[0..11]          */
#define TTM_U_GSTRIDE    24   /* doubles per group in the U section                                                       */
#define TTM_U_DEG        11   /* degree of the spline pieces                                    */
#define TTM_U_TSTRIDE    14   /* doubles per spline column (16-byte reads of 16 consecutive columns hit disjoint banks) */
#define TTM_U_NI_MAX    128   /* most columns a spline may have (LDS budget of the staged table) */
#define TTM_UC_LEN        8
#define TTM_UC_KC         0
#define TTM_UC_KC_SLOT    1   /* planned column cache slot for x_kc (-1: none)                  */
#define TTM_UC_N_GRP      2   /* nonmonotone groups                                             */
#define TTM_UC_GRP_OFF    3   /* first group record                                             */
#define TTM_UC_NI         4   /* spline columns (0: the component has no special terms)         */
#define TTM_UC_TAB_OFF    5   /* doubles, relative to the U section                             */
#define TTM_UC_DBL_OFF    6
#define TTM_UC_FLAGS      7
#define TTM_UCF_OWN       1   /* group record N_GRP holds the monotone polynomial / HF terms    */
#define TTM_UCF_PUT_E     2   /* exp(-x_kc^2/4) is stored with x_kc (a later group reads it): the U-form kernels
                                 compute it eagerly, so every Hermite-function reader of a swept column is a hit */
/* ucomp is followed by TTM_PLAN_WAYS entry-state words per component (column | TTM_PLAN_E, -1 = empty): the cache
 * contents when a sweep reaches that component, under the eager semantics above                                  */
#define TTM_UC_STATE(D, k) ((D) * TTM_UC_LEN + (k) * TTM_PLAN_WAYS)
#define TTM_UG_LEN        8
#define TTM_UG_VAR        0
#define TTM_UG_FLAGS      1   /* TTM_PLAN_* | degree of B << 16 | TTM_UGF_POLY | degree of A << 24 */
#define TTM_UG_SRCA       2   /* offset in the component's folded block of the plain coefficients of P_1.. (-1: none) */
#define TTM_UG_PA         3
#define TTM_UG_SRCB       4   /* same for the Hermite-function coefficients                     */
#define TTM_UG_PB         5
#define TTM_UG_DEGB(f) (((f) >> 16) & 15)
#define TTM_UG_DEGA(f) (((f) >> 24) & 15)
#define TTM_UGF_POLY     (1 << 20)   /* group has plain polynomial terms                        */

/* "hot" records of a U-form map (H section, appended to the U section by ttm_fold when u_h_cls > 0): one
 * fixed-stride record per component holding everything a sweep step needs, so that the kernel issues all scalar
 * loads of a step at once instead of chasing ucomp -> ugrp -> U offsets:
 *   header, TTM_H_HDR doubles: int32 {put slot (2 x way, -1: none), flags (bit 0: store exp(-x_kc^2/4) too)},
 *           int32 {NI, kc}, c0, -t_lo/h, 1/h, 2/h, int32 {TAB_OFF, number of groups}, constant of the
 *           nonmonotone part alone (c0 also carries the monotone constants)
 *   u_h_ng group records of GS doubles: int32 {slot of the column (2 x way), 1} ({0, 0} for the padding records
 *           beyond the component's own groups), B[0..DB], A[0..DA], zero padding;
 *           (DB, DA, GS) = (3,1,8) / (5,5,16) / (7,7,24) / (10,10,24) for u_h_cls = 1 / 2 / 3 / 4 (class 4: maps of at most
 *           TTM_P_FEW_D components only - their records exist as the source of the push records).
 * Available (u_h_cls > 0) when every group of a full sweep reads a column the sweep itself produced (all cache
 * hits), no component has polynomial / Hermite-function terms in its monotone list and none has more than
 * TTM_H_NG_MAX nonmonotone groups.                                                                              */
#define TTM_H_HDR         8
#define TTM_H_NG_MAX      5   /* (hot kernels: 2 or 4 group records; 5: maps of a few components whose records only feed the push records) */

/* "push" records of a BANDED U-form map (P section, behind the H section; written by ttm_fold when u_p_lag > 0).
 * Banded: the components' columns are consecutive (kc_k = kc_0 + k), every nonmonotone group of component k reads a
 * column kc_k - 1 .. kc_k - u_p_lag, the monotone part of every component is a special-term spline (maps of a few components:
 * and / or one linear term of its own variable) and hot records exist (u_h_cls > 0).
 * u_p_lag is 2, or 3 / 5 (= TTM_P_LAG_MAX: a group four or five columns back - the smoother's block map of example_07.py:368-408)
 * for a map of at most TTM_P_FEW_D components whose groups do not all hit the planned
 * column cache (a group three columns back; conditioning columns in front of the first component) or that have linear own
 * terms; in the latter case the
 * hot records exist ONLY as the source of the push records (the kernels that sweep hot records do not take such a map).
 * The band kernels (csrc/ttm_band.hip) walk the columns and add what a column contributes to the components that
 * read it as soon as the column is known, so the records are indexed by COLUMN: record r (0 <= r < D + u_p_lag)
 * belongs to column kc_0 - u_p_lag + r, i.e. to component k = r - u_p_lag when that is >= 0 (the first u_p_lag records
 * stand for the columns in front of the first component: conditioning columns, or nothing - all coefficients zero).
 * u_p_stride doubles per record:
 *   [0] constant the running sum of the component u_p_lag columns on starts from, forward map: c0 (nonmonotone +
 *       monotone constants) + the constant terms of its groups' plain polynomials; 0 when there is no such component
 *   [1] the same for the inverse (nonmonotone constant + the groups' constant terms: the offset TM:4039 subtracts)
 *   [2] 1 - t_lo/h   [3] 1/h   [4] 2/h      spline geometry of the record's own component: column = trunc(x [3] + [2])
 *   [5] int32 {NI, TAB_OFF}   [6] int32 {k (-1: none), 0}   [7] slope of the component's own linear term (maps of a few
 *       components whose monotone list holds [k] next to, or instead of, special terms: NI = 0 then; its constant is in [0])
 *   [8 + l GP ...], l = 0..u_p_lag-1: the group of component k + l + 1 that reads this column (zeros if none):
 *       B[0..DB] (Hermite-function part, monomial coefficients), A[1..DA] (plain part without its constant term);
 *       GP = DB + 1 + DA, (DB, DA) as in the hot records.
 * ttm_fold also writes, into padding slot 12 of every spline column c of such a map, the offset s0 of its local
 * coordinate: s = x [4] + s0.                                                                                     */
#define TTM_P_HDR         8
#define TTM_P_LAG_MAX     5
#define TTM_P_FEW_D       4   /* components of a map the lag-3 kernels take, at most */

typedef struct ttm_program {
    /* device tables */
    const int32_t* itab;        /* all component blocks, back to back              */
    const int32_t* ftab;        /* fold recipes of all components, back to back    */
    const int32_t* fdesc;       /* fast-path descriptors, TTM_FDESC_LEN int32 per component */
    const int32_t* fints;       /* fast-path int stream per component: groups {var, P, alpha offset, TTM_PLAN_* flags},
                                   special-term kinds (bfun order), order of the unified records (indices into
                                   the bfun order), planned-cache entry state                              */
    const double*  dpar;        /* HF constants and special-term (centre, scale)   */
    const double*  quad_x;      /* Gauss-Legendre nodes   (TM:199-225), length Q   */
    const double*  quad_w;      /* Gauss-Legendre weights,               length Q   */
    /* host tables, length D+1 each (prefix offsets per component) */
    const int32_t* h_comp_off;  /* into itab                                       */
    const int32_t* h_dpar_off;  /* into dpar                                       */
    const int32_t* h_coef_off;  /* into the coefficient vector [nonmon_k | mon_k]  */
    const int32_t* h_nslots;    /* length D: per-sample scratch doubles the map kernels need
                                   (nB+1 for components with monotone cross terms, else 0) */
    const int32_t* h_n_nm;      /* length D: number of nonmonotone terms           */
    const int32_t* h_fold_off;  /* length D+1: prefix offsets of folded coefficients */
    const int32_t* h_ftab_off;  /* length D+1: prefix offsets into ftab              */
    const int32_t* h_nb1;       /* length D: nB+1 (distinct x_k functions + 1)      */
    const int32_t* h_complex;   /* length D: bit 0 = component needs the generic interpreter (cross / generic terms),
                                   bit 1 = integrated component with a dense B set (orders 1..P, no special terms),
                                   bit 2 = integrated component whose functions of x_k are polynomials / Hermite functions
                                   only (any orders): the monomial-form kernels of csrc/ttm_int.hip apply,
                                   bit 3 = the component has special terms (LET / RET / RBF / iRBF) somewhere,
                                   bit 4 = the component has an X program (TTM_HDR_OFF_XPROG), bits 16-23 = TTM_HDR_X_NROW, bits 24-25 = TTM_HDR_X_NSUM / 64 rounded up,
                                   bits 8-11 / 12-15 = largest Hermite-function / plain polynomial order among the
                                   functions of x_k (saturating at 15)                                          */
    /* device copy of the five prefix tables, 5 x (D+1) int32:
       [comp_off | dpar_off | coef_off | fold_off | ftab_off]                        */
    const int32_t* d_offsets;
    int32_t D;                  /* number of map components (len(monotone))        */
    int32_t d_cols;             /* columns of the sample matrix (skip + D)         */
    int32_t family;             /* TTM_FAM_*                                       */
    int32_t monotonicity;       /* TTM_MONO_*                                      */
    int32_t rectifier;          /* TTM_RECT_*                                      */
    int32_t Q;                  /* quadrature order                                */
    int32_t plan_ways;          /* ways of the planned column cache (1..TTM_PLAN_WAYS) the fints plan was made for */
    int32_t u_enabled;          /* 1: the map has a U-form (separable, all terms univariate, orders <= TTM_U_PMAX, splines
                                   within TTM_U_NI_MAX columns): ttm_fold also writes the U section and ttm_forward /
                                   ttm_inverse_table run the U-form kernels                                       */
    double  delta;              /* TM:34, added to the rectifier / to dS           */
    /* U-form tables (valid when u_enabled) */
    const int32_t* ucomp;       /* device, TTM_UC_LEN x D                            */
    const int32_t* ugrp;        /* device, TTM_UG_LEN per group                      */
    const double*  umono;       /* device, (TTM_U_PMAX+1)^2: row n = monomial coefficients of P_n */
    const double*  ugeo;        /* device, 2 doubles per component {t_lo, h} of its spline        */
    const int32_t* h_ucomp;     /* host copy of ucomp (launch planning)              */
    const int32_t* h_ugrp;      /* host copy of ugrp                                 */
    int64_t u_size;             /* doubles of the U section (H section included)     */
    int64_t u_err_off;          /* offset (within the U section) of the 2 x D fit errors */
    int64_t u_h_off;            /* offset (within the U section) of the hot records  */
    int32_t u_h_cls;            /* 0: no hot records; 1..3: degree class             */
    int32_t u_h_ng;             /* group records per component                       */
    int64_t u_p_off;            /* offset (within the U section) of the push records */
    int32_t u_p_lag;            /* 0: not a banded map; else groups per push record: 2, or 3 (see above) */
    int32_t u_p_stride;         /* doubles per push record                           */
} ttm_program;

const char* ttm_last_error_string(void);
/* sets the calling thread's error text: entry points that run tasks on worker threads of their own (the *_batch
 * optimisers) republish a worker's failure on the thread that made the call */
int ttm_set_error_string(const char* text);
int  ttm_version(void);
/* name of the (last) kernel the most recent launching entry point of this thread dispatched, e.g. "k_inverse_hl":
 * lets a benchmark name the kernel its timings belong to (which variant runs is decided inside the library) */
const char* ttm_last_kernel(void);
/* Launch-planning options (tests and tuning runs): which kernel variant an entry point picks is normally decided from
 * the program and the ensemble size; an option pins one aspect of that choice, e.g. ttm_set_option("no_uform", 1),
 * ("forward_ns", 2), ("rt_off", 1) - the list is TTM_OPTIONS in csrc/ttm_kernels.hip.  Defaults come from the environment
 * variables TTM_<NAME>, read once when the library is first used (never on the launch path); ttm_reset_options() goes
 * back to them.  Options change speed and kernel choice only, never results beyond the documented tolerances.       */
int ttm_set_option(const char* name, int32_t value);
int ttm_reset_options(void);
/* sizeof(ttm_program) as the library was compiled: bindings check their mirror of the struct against it */
int64_t ttm_program_sizeof(void);
/* number of visible HIP devices; 0 with an error string when there is none */
int  ttm_device_count(int* count);

/* ---- K0/K1: layout change + (de)standardisation ----------------------------
 * TM:750-787 standardize(); TM:2413-2417, 3701-3704, 3721-3724 inline affine maps.
 * ttm_colstats: mean and ddof-0 standard deviation of every column of a
 *   row-major N x d host-layout matrix resident on the device.
 *   work: >= ttm_colstats_work_size(N, d) doubles.
 *   Up to 8 columns and 131 072 rows: ONE launch (per-workgroup mean and squared deviations, combined in workgroup
 *   order by the last workgroup; option `colstats_one` = 0: the four launches of the general shape).
 * ttm_colstats_cols: the same moments of a COLUMN-major matrix Xcols[j*ld + n] that is already on the device (reset of
 *   device-resident samples: no row-major copy); shapes outside the one-launch range: TTM_E_UNSUPPORTED (export, then
 *   ttm_colstats).  Same sums, same bits, as ttm_colstats of the row-major copy.
 * ttm_standardize_cols: Xs[j*ldx + n] = (Xcols[j*ld + n] - mean[j]) / std[j].
 * ttm_import:  Xsoa[j*ldx + n] = (Xrow[n*d + j] - mean[j]) / std[j]   (mean/std NULL: plain transpose)
 * ttm_export:  Xrow[n*dout + j] = Xsoa[(j0+j)*ldx + n] * std[j0+j] + mean[j0+j]          */
int64_t ttm_colstats_work_size(int64_t N, int32_t d);
int ttm_colstats(const double* Xrow, int64_t N, int32_t d, double* mean, double* std,
                 double* work, void* stream);
int ttm_colstats_cols(const double* Xcols, int64_t ld, int64_t N, int32_t d, double* mean, double* std,
                      double* work, void* stream);
int ttm_standardize_cols(const double* Xcols, int64_t ld, int64_t N, int32_t d, const double* mean,
                         const double* std, double* Xs, int64_t ldx, void* stream);
int ttm_import(const double* Xrow, int64_t N, int32_t d, const double* mean, const double* std,
               double* Xsoa, int64_t ldx, void* stream);
int ttm_export(const double* Xsoa, int64_t ldx, int64_t N, int32_t j0, int32_t dout,
               const double* mean, const double* std, double* Xrow, void* stream);

/* ---- K9: exact order statistics of one column (radix select) -----------------------
 * np.quantile (method 'linear') of TM:775-778 (quantile standardisation) and TM:2266-2296 (special-term
 * centres) interpolates between two order statistics; this entry point returns them exactly:
 * out[j] = ranks[j]-th smallest value of col[0..N) (0-based).  ranks: device int64[nr], nr <= 16.
 * work: device scratch, ttm_select_work_size(nr) bytes, no initial state needed.  Eight histogram passes over the column, no
 * host sync: columns of up to 131 072 rows in ONE launch (k_select_coop: the keys stay in registers, the <= 64 workgroups meet
 * at a grid barrier per pass - every wait bounded, workgroup 0 finishes alone if its partners do not arrive -; option
 * `select_coop` = 0 switches it off), longer columns and sharded ones as 17 launches. */
int64_t ttm_select_work_size(int32_t nr);
int ttm_order_statistics(const double* col, int64_t N, const int64_t* ranks, int32_t nr, double* out,
                         void* work, void* stream);
/* The same select over a column whose rows are sharded over the ranks of a communicator (SURVEY.md section 8e:
 * special-term placement of a sample-sharded ensemble): N = local rows, ranks = GLOBAL 0-based ranks; every pass
 * all-reduces its nr x 256 bin counts (ttm_allreduce_i32), every rank ends with the same exact values; no element of
 * the column moves.  comm NULL: ttm_order_statistics.                                                              */
struct ttm_comm;
int ttm_order_statistics_dist(const double* col, int64_t N, const int64_t* ranks, int32_t nr, double* out,
                              void* work, struct ttm_comm* comm, void* stream);

/* ---- folded coefficients -------------------------------------------------------
 * The map kernels evaluate the nonmonotone part per variable with summed ("folded") coefficients
 * and the monotone part through per-function weights (layout: "Folded coefficients" above).
 * ttm_fold computes them for all components from the coefficient vector
 * coef = [nonmon_0 | mon_0 | nonmon_1 | ...]; fold has ttm_fold_size(p) doubles (the folded coefficients
 * plus 8 doubles of read-ahead padding; the kernels load coefficients four at a time).  Call it whenever
 * the coefficients change, before ttm_forward / ttm_inverse_*.                                    */
int64_t ttm_fold_size(const ttm_program* p);
/* offset (doubles) of the U section inside the fold buffer; -1 when the program has no U-form.  The 2 x D fit errors
 * {value, derivative} of the special-term splines are at fold + ttm_uform_offset(p) + p->u_err_off after ttm_fold: a
 * host that wants the guarantee reads them back (once per special-term placement is enough in practice: the error is
 * governed by the interval width, which the host chooses from the scales) and clears u_enabled if they exceed its
 * tolerance - the direct kernels then run instead.                                                               */
int64_t ttm_uform_offset(const ttm_program* p);
int ttm_fold(const ttm_program* p, const double* coef, double* fold, void* stream);
/* ttm_fold_staged: ttm_fold for a U-form map whose packed coefficient vector is still in page-locked HOST memory the device
 * can read (h_coef: hipHostMalloc / torch pinned memory): the fold kernel copies the vector into `coef` (device, written) itself,
 * so a new coefficient vector costs no host-to-device copy in the stream; h_err (page-locked host, nullable) also receives
 * the 2 D fit errors {value, derivative} of the special-term splines - readable behind an event on `stream`, no device-to-host
 * copy.  TTM_E_UNSUPPORTED when the map has no U-form (copy the coefficients and call ttm_fold).                      */
int ttm_fold_staged(const ttm_program* p, const double* h_coef, double* coef, double* fold, double* h_err, void* stream);

/* ---- K2/K3: forward map ------------------------------------------------------
 * TM:2391-2437 map(), TM:2439-2567 s(), TM:4238-4258 GaussQuadrature (fused),
 * TM:4981-5018 rectifier.evaluate; log-determinant part of TM:2618-2641 / 2686-2709.
 * Z[(k-k0)*ldz + n] = S_k(x_n) for k in [k0,k1).  Z may be NULL (derivative only).
 * logdet (nullable, length N): sum_k log( (dS_k/dx_k) / sigma[k-k0] )  (sigma NULL: no division)
 * sumsq  (nullable, length N): sum_k S_k(x_n)^2, the Mahalanobis part of TM:2681-2684.            */
int ttm_forward(const ttm_program* p, const double* coef, const double* fold, const double* Xsoa, int64_t ldx,
                int64_t N, int32_t k0, int32_t k1, double* Zsoa, int64_t ldz,
                double* logdet, const double* sigma, double* sumsq, void* stream);

/* ---- basis matrices (inspection / tests) -------------------------------------
 * TM:1498-1575 fun_mon / fun_nonmon, TM:2047-2120 der_fun_mon for component k.
 * which: 0 = Psi_nonmon, 1 = Psi_mon, 2 = dPsi_mon/dx_k.  out[i*ldo + n].                        */
int ttm_basis(const ttm_program* p, int32_t k, int32_t which, const double* Xsoa, int64_t ldx,
              int64_t N, double* out, int64_t ldo, void* stream);

/* ---- K4: table ("alternate") inverse, separable maps ---------------------------
 * TM:3987-4084 vectorized_root_search_alternate + the k-loop of TM:3639-3796.
 * ttm_inverse_table_build: out[(k-k0)*T + i] = Psi_mon_k(0,..,pts[i],..,0) . c_mon_k   (TM:4047-4058)
 * ttm_inverse_table: for k in [k0,k1) sequentially: offset = Psi_nonmon_k(x) . c_nonmon_k,
 *   target = clip(-offset + Z_k, tmin_k, tmax_k) (if truncate), x_k = lerp on (tab_x, tab_y) with
 *   scipy.interpolate.interp1d semantics (searchsorted-left, clip to [1,T-1], slope form).
 *   tab_x: (k1-k0) x T, non-decreasing (the host applies interp1d's stable sort); tab_y: the matching
 *   abscissae, row k at tab_y + (k-k0)*ldy (ldy = 0: one shared row, e.g. the unpermuted linspace).
 *   h_y_affine (host, nullable): {y0, step, y_last} when ldy = 0 and tab_y[i] == i*step + y0 (i < T-1),
 *   tab_y[T-1] == y_last bit for bit (np.linspace): the abscissae are then computed instead of gathered.
 *   bkt (int32, (k1-k0) x (nb+1)): search accelerator as ttm_inverse_table_index writes it: bkt[b] = number of entries
 *   whose bucket - clamp((int)fma(x, nb / (tmax - tmin), -tmin nb / (tmax - tmin)), 0, nb-1), monotone in x - is below b;
 *   the kernels compare only the entries of the target's own bucket (k_inverse_rt) or bisect inside the buckets around
 *   it (generic kernel, which also accepts bkt[b] = searchsorted_left(tab_x, tmin + b (tmax-tmin)/nb)) - same index as
 *   the full search.
 *   Xsoa holds the conditioning columns on entry and receives column kc of every component.       */
int ttm_inverse_table_build(const ttm_program* p, const double* coef, const double* fold, int32_t k0, int32_t k1,
                            const double* pts, int32_t T, double* out, void* stream);
/* ttm_inverse_table_index: for ncomp tables (rows of tab_x, length T) compute on the device what the lookup
 * needs: tmin/tmax (np.min / np.max of the row, TM:4075-4076), the bucket index bkt (ncomp x (nb+1)) and
 * unsorted[i] = 1 when row i is not non-decreasing (then interp1d's stable sort must be applied first -
 * the host does that rare case - and tmin/tmax/bkt of that row are not valid).                          */
int ttm_inverse_table_index(const double* tab_x, int32_t ncomp, int32_t T, int32_t nb, double* tmin, double* tmax,
                            int32_t* bkt, int32_t* unsorted, void* stream);
/* ttm_inverse_table_build_index: ttm_inverse_table_build followed by ttm_inverse_table_index of its output as ONE launch
 * (one workgroup per component; T <= 2048): what a new coefficient vector costs before its first table lookup
 * (TM:4047-4058 rebuilds the table in every inverse_map).  Same bits as the two calls.                      */
int ttm_inverse_table_build_index(const ttm_program* p, const double* coef, const double* fold, int32_t k0, int32_t k1,
                                  const double* pts, int32_t T, int32_t nb, double* out, double* tmin, double* tmax,
                                  int32_t* bkt, int32_t* unsorted, int32_t* h_unsorted, double* img, void* stream);
/* (h_unsorted: page-locked host memory, nullable - the sortedness flags also land there, readable behind an event on `stream`)
 * img (device, 16-byte aligned, nullable): (k1-k0) x ttm_inverse_table_image_doubles(p, k0, k1, T, nb) doubles that receive
 * the RESIDENT-TABLE IMAGE of every component - search parameters, the window of the table the lookup kernel of banded maps
 * keeps in LDS, its 16-bit bucket index (csrc/ttm_band_image.h) - laid out once here so that ttm_inverse_table copies them
 * into LDS by DMA instead of assembling them per workgroup.  A pure function of the table: passing it changes no result.
 * ttm_inverse_table_image_doubles returns 0 when the map / table geometry has no such kernel (img must then be NULL).       */
int64_t ttm_inverse_table_image_doubles(const ttm_program* p, int32_t k0, int32_t k1, int32_t T, int32_t nb);
/* ---- forward map + table inverse of the image in ONE launch (maps of at most four components, N >= 65 536) -----------------
 * What BASELINE configs[1] times as a step - map(X), then inverse_map of the result (TM:2554-2558, 2646-2712; optionally the
 * log-determinant and the sum of squares of the pullback density, TM:3663-3789) - as one pass over the ensemble: Zsoa (nullable)
 * = S(X), Xr = S^-1(S(X)) with the conditioning columns (if any) read from Xsoa; the columns of a tile are read once and z stays in
 * registers (k_band_few_roundtrip).  The same bits as ttm_forward followed by ttm_inverse_table on the default tables.  tab_x ..
 * nb: the sorted default tables of all components as for ttm_inverse_table (h_y_affine required).  TTM_E_UNSUPPORTED: not for
 * this map / these options, or a shape the one launch is not faster for (a sweep that reaches three columns back, the density
 * terms: option `roundtrip_fused` = 1 runs them anyway): make the two calls.                                                  */
int ttm_roundtrip(const ttm_program* p, const double* coef, const double* fold, const double* Xsoa, int64_t ldx, int64_t N,
                  double* Zsoa, int64_t ldz, double* Xr, int64_t ldr, double* logdet, const double* sigma, double* sumsq,
                  const double* tab_x, int32_t T, const double* h_y_affine, const double* tmin, const double* tmax,
                  const int32_t* bkt, int32_t nb, void* stream);

/* ttm_setup_staged: ttm_fold_staged and ttm_inverse_table_build_index of ALL components (k0 = 0, k1 = D) as ONE launch whose two
 * kinds of workgroups run side by side (the tables need only the folded coefficients: the table workgroups fold for themselves into
 * fold2, a scratch of ttm_fold_size doubles, zero-filled once per layout like fold, that nobody else reads).  Every output as the
 * two calls write it, bit for bit.  TTM_E_UNSUPPORTED: not for this map (the caller makes the two calls instead).                  */
int ttm_setup_staged(const ttm_program* p, const double* h_coef, double* coef, double* fold, double* fold2, double* h_err,
                     const double* pts, int32_t T, int32_t nb, double* out, double* tmin, double* tmax, int32_t* bkt, int32_t* unsorted,
                     int32_t* h_unsorted, double* img, void* stream);
int ttm_inverse_table(const ttm_program* p, const double* coef, const double* fold, int32_t k0, int32_t k1,
                      const double* Zsoa, int64_t ldz, double* Xsoa, int64_t ldx, int64_t N,
                      const double* tab_x, const double* tab_y, int64_t ldy, int32_t T,
                      const double* h_y_affine,
                      const double* tmin, const double* tmax, const int32_t* bkt, int32_t nb,
                      int32_t truncate, const double* img, int64_t img_doubles, void* stream);
/* (img: what ttm_inverse_table_build_index wrote for these tables and this [k0, k1), or NULL; img_doubles: doubles per
 * component it was laid out with - an image of another layout than this launch plans for is ignored, not misread) */

/* ---- K5: bisection inverse (both monotonicity modes) -----------------------------
 * TM:3798-3985 vectorized_root_search_bisection, exact bracket / window-shift / midpoint
 * sequence per sample (start_distance 2, threshold 1e-9, max_iterations 100).
 * iters (int32, length k1-k0, zero-initialised by the caller): receives the maximum midpoint-
 *   iteration count over the processed samples (atomic max).
 * cap (nullable, int32, length k1-k0): when given, every sample stops after cap[k-k0] midpoint
 *   iterations - used to replay global sample 0 under the reference's `np.sum(indices) > 0`
 *   loop guard (TM:3952).                                                                          */
int ttm_inverse_bisect(const ttm_program* p, const double* coef, const double* fold, int32_t k0, int32_t k1,
                       const double* Zsoa, int64_t ldz, double* Xsoa, int64_t ldx, int64_t N,
                       int32_t* iters, const int32_t* cap, void* stream);

/* ---- K5 (second mode): safeguarded Newton inverse - NOT the reference's method ------------
 * Same start, window doubling and stopping rule (|S - z| <= 1e-9, at most 100 trial points) as
 * ttm_inverse_bisect; inside the bracket the trial points are Newton steps with the analytic
 * dS/dx_k (midpoint when a step leaves the bracket).  Converges to the same root in ~6 instead of
 * ~33 evaluations of S; the last trial point is not the reference's last midpoint (difference
 * <= 1e-9 / (dS/dx)).  Opt-in of the host class (root_finder='newton'); the default stays
 * ttm_inverse_bisect.  iters: as above (maximum number of Newton / midpoint trial points).        */
int ttm_inverse_newton(const ttm_program* p, const double* coef, const double* fold, int32_t k0, int32_t k1,
                       const double* Zsoa, int64_t ldz, double* Xsoa, int64_t ldx, int64_t N,
                       int32_t* iters, void* stream);

/* ---- K6/K7: objective + gradient reductions for optimize() -------------------------
 * integrated: TM:3300-3376 objective_function, TM:3435-3569 objective_function_jacobian
 *   out[0] = sum_n ( S^2/2 - log(r(g)+delta) ), out[1..] = sum_n d/dc of the same, [nonmon | mon]
 * separable : inner loop of TM:2978-3018 fun_mon_objective
 *   out[0] = sum_n log dS_n, out[1+i] = sum_n dPsi_{n,i}/dS_n, dS = dPsi.c + delta*rowsum(dPsi)
 * (regularisation, 1/N and the A-matrix terms are O(m) host arithmetic).
 * coef_k: device vector [nonmon_k | mon_k] for this component (the trial point); its folded
 * coefficients are computed internally into `work`.
 * work: >= ttm_reduce_work_size(nout) doubles.  out: device, nout doubles.                        */
int64_t ttm_reduce_work_size(int32_t nout);
int ttm_objective(const ttm_program* p, int32_t k, const double* coef_k, const double* Xsoa,
                  int64_t ldx, int64_t N, double* work, double* out, void* stream);
/* The same reduction for an optimiser that lives on the host (TM:3108-3114, 3252-3257 drive SciPy with one
 * objective / gradient evaluation per call, so the per-call latency is what optimize() costs):
 * h_coef_k is a HOST vector (<= 128 coefficients) that travels as kernel arguments - no host-to-device copy; the
 * finishing sum runs in the workgroup that draws the last ticket of `counter` (device uint32[16], zero before the first
 * call; left zero) in the order of the three-launch path, so both give the same bits; `out` may be pinned host
 * memory (device-accessible), which also saves the device-to-host copy.  Two launches instead of three + two copies. */
int ttm_objective_host(const ttm_program* p, int32_t k, const double* h_coef_k, const double* Xsoa,
                       int64_t ldx, int64_t N, double* work, uint32_t* counter, double* out, void* stream);

/* Separable objective from a cached derivative basis: while one component is optimised its dPsi_mon (N x m_mon;
 * the reference precalculates and keeps it, TM:789-821, 2978-3018) does not change, so the host computes it once with
 * ttm_basis(which = 2) (m rows of N doubles, row stride ldp) and every evaluation is one streaming launch:
 *   out[0] = sum_n log dS_n, out[1+i] = sum_n dPsi_{n,i}/dS_n, dS = dPsi.c + delta*rowsum(dPsi), m <= 16.
 * h_coef_mon: HOST vector of the m trial coefficients (kernel arguments).  work: >= ttm_reduce_work_size(1+m) doubles;
 * counter / out as ttm_objective_host.                                                                             */
int ttm_objective_sep_cached(const double* dPsi, int64_t ldp, int64_t N, int32_t m, const double* h_coef_mon,
                             double delta, double* work, uint32_t* counter, double* out, void* stream);
/* The two host-loop reductions with a completion mark: the finishing workgroup writes *flag = mark (flag: pinned host
 * memory, nullable) after its results have reached memory, so a host loop polls ONE location per evaluation and needs
 * neither hipStreamSynchronize nor a separate ttm_signal launch (ttm_optimize_separable / _integrated do that).      */
int ttm_objective_host_marked(const ttm_program* p, int32_t k, const double* h_coef_k, const double* Xsoa,
                              int64_t ldx, int64_t N, double* work, uint32_t* counter, double* out, double* flag,
                              double mark, void* stream);
/* The same evaluation with SELF-VALIDATING sums for grids of up to 128 workgroups (N <= 131 072) - no ticket, no fence, no
 * mark: ttm_sentinel_fill arms the rows of partial sums in `work` once (every slot a signalling-NaN pattern no arithmetic
 * produces; TTM_E_UNSUPPORTED: the grid is larger - measured slower than the second launch there - or option sep_sentinel = 0:
 * take the marked call); ttm_objective_sep_cached_sent evaluates, workgroup 0 polls the slots of the others, adds the rows in
 * the order of the ticket finish (the same bits), leaves them armed again, and writes the 1 + m sums to out_host -
 * page-locked host memory whose slots the CALLER has armed with the same pattern (0x7FF4DEADBEEF0001) and polls until none
 * holds it; 0x7FF4DEADBEEF0002 in a slot: the device gave up waiting (csrc/ttm_optim.cpp: arm_values / poll_values).   */
int ttm_sentinel_fill(double* work, int32_t m, int64_t N, void* stream);
int ttm_objective_sep_cached_sent(const double* dPsi, int64_t ldp, int64_t N, int32_t m, const double* h_coef_mon,
                                  double delta, double* work, double* out_host, void* stream);
int ttm_objective_sep_direct_sent(const double* xk, int64_t N, int32_t m, const int32_t* kinds, const double* pars,
                                  const double* h_coef_mon, double delta, double* work, double* out_host, void* stream);
/* The evaluation SERVER of an optimiser loop over grids of up to 128 workgroups (k_objective_sep_server): ONE launch for the whole
 * loop - resident workgroups poll a mailbox in fine-grained DEVICE memory that the host writes through the PCIe BAR (posted
 * writes: 2.6-3.7 us per request against 6.2 us for a launch + completion round trip, tools/micro/mailbox.cpp) and answer every
 * request the way ttm_objective_sep_cached_sent does (the same bits).  ttm_mailbox_acquire: a 256-byte mailbox of the library's
 * pool (NULL: none - launch per evaluation); the host writes the m trial coefficients to box + 8, then (sfence) the word
 * gen << 32 | request number to box, request numbers 1, 2, ...; gen << 32 | 0xffffffff ends the server; a server that is not
 * asked anything for 0.2 s leaves by itself.  work / out_host as ttm_objective_sep_cached_sent (rows armed by ttm_sentinel_fill).
 * TTM_E_UNSUPPORTED: larger grids, option sep_server = 0.                                                                */
void* ttm_mailbox_acquire(void);
void ttm_mailbox_release(void* box);
int ttm_objective_sep_server_start(const double* dPsi, int64_t ldp, int64_t N, int32_t m, double delta, double* work,
                                   double* out_host, const void* box, uint32_t gen, void* stream);
int ttm_objective_sep_cached_marked(const double* dPsi, int64_t ldp, int64_t N, int32_t m, const double* h_coef_mon,
                                    double delta, double* work, uint32_t* counter, double* out, double* flag,
                                    double mark, void* stream);
/* The separable reduction with the derivative basis recomputed from the x_k column (xk: device, N doubles) instead of
 * read from a cached N x m matrix - for components whose monotone terms are all plain special terms of x_k: kinds
 * (device, m int32: TTM_KIND_LET / RET / RBF / IRBF) and pars (device, 5 m doubles: {centre, scale, 1/(sqrt2 scale),
 * scale sqrt(2/pi), 1/(sqrt(2 pi) scale)} per term, as in the program's constant array), both in coefficient order.
 * Same sums, bit for bit, as ttm_objective_sep_cached on the basis ttm_basis(which = 2) writes; 8 bytes of HBM per row
 * instead of 8 m.                                                                                                    */
int ttm_objective_sep_direct_marked(const double* xk, int64_t N, int32_t m, const int32_t* kinds, const double* pars,
                                    const double* h_coef_mon, double delta, double* work, uint32_t* counter,
                                    double* out, double* flag, double mark, void* stream);

/* ---- K8: Gram matrix of [Psi_nonmon | Psi_mon] -------------------------------------
 * replaces the N x m passes of TM:2966-2975 (QR projection) and TM:3031-3050 (L2 normal
 * equations): out[i*m + j] = sum_n Psi_{n,i} Psi_{n,j}, m = n_nonmon + n_mon (full symmetric).
 * work: >= ttm_reduce_work_size(m*m) doubles.                                                      */
int ttm_gram(const ttm_program* p, int32_t k, const double* Xsoa, int64_t ldx, int64_t N,
             double* work, double* out, void* stream);
/* The Gram matrices of nk <= 8 components (host list ks) in one launch + one reduction launch; out: the m_k x m_k blocks one
 * behind the other, in the order of ks.  The same sums, the same bits, as ttm_gram component by component (work: as for
 * the largest of them).  TTM_E_UNSUPPORTED: not for these components (more than 16 basis functions, no matrix-core path, more
 * partial sums than `work` holds): call ttm_gram for each.                                                                */
int ttm_gram_many(const ttm_program* p, const int32_t* ks, int32_t nk, const double* Xsoa, int64_t ldx, int64_t N,
                  double* work, double* out, void* stream);

/* ---- column utilities of the device-resident ensemble filter -----------------------------------------------
 * example_06.py:252-328 (the caller of the hot path in Examples C) keeps the ensemble in host NumPy; entf.Filter keeps
 * it column-major on the device and needs three small kernels around the map calls:
 * ttm_lorenz63_rk4: nt RK4 steps of length dt of the Lorenz-63 system (example_06.py:28-76) on the 3 x N ensemble E;
 * ttm_perturb: out[n] = in[n] + sd * noise[n]; noise NULL: standard normal deviates of a counter-based generator
 *   (Philox-4x32-10 + Box-Muller, a function of (seed, stream_id, row0 + n) only);
 * ttm_map_columns: out[j][n] = in[src[j]][n] * scale[j] + shift[j] for j < ncols <= 16 (src[j] < 0: the constant
 *   shift[j]; scale / shift NULL: 1 / 0; src, scale, shift: host) - column gather, (de)standardisation, permutation. */
int ttm_lorenz63_rk4(double* E, int64_t ld, int64_t N, double dt, int32_t nt, void* stream);
int ttm_perturb(const double* in, const double* noise, double sd, uint64_t seed, uint32_t stream_id, int64_t row0, int64_t N,
                double* out, void* stream);
int ttm_map_columns(const double* in, int64_t ldi, const int32_t* src, const double* scale, const double* shift,
                    int32_t ncols, int64_t N, double* out, int64_t ldo, void* stream);

/* ---- optimisers -----------------------------------------------------------------------------------------
 * TM:3108-3114 (scipy.optimize.minimize, method 'L-BFGS-B': maxcor 10, ftol 2.22e-9, gtol 1e-5, maxls 20) as a host
 * C++ loop (csrc/ttm_lbfgsb.h: L-BFGS-B 3.0 restated, dense for the few dozen variables of a map component).
 * ttm_lbfgsb_minimize: generic front end - fun(n, x, &f, g, user) returns 0 or an error code; lb / ub may be NULL or
 *   hold -inf / +inf for missing bounds; result (nullable, 5 doubles): {f, projected-gradient norm, iterations,
 *   evaluations, status: 0 = projected gradient <= gtol, 1 = relative reduction <= ftol, 2 = iteration limit,
 *   3 = abnormal termination in the line search}.
 * ttm_optimize_separable: the reduced separable problem of one component (TM:2978-3018) minimised without leaving the
 *   library: J(c) = c'Ac/2 - sum_n log dS_n / Ntotal + c.b with dS = dPsi.c + delta rowsum(dPsi) from the cached
 *   derivative basis (ttm_objective_sep_cached).  A (m x m, row-major), b, lb, ub, x (start / result): host.
 *   sums_host: >= 2 + m doubles, pinned host memory the device can write (the last one is the completion mark); sums_dev (device, >= 1 + m doubles) and
 *   comm: non-NULL when the samples are sharded over ranks - the local sums are then combined by ONE
 *   ttm_allreduce_f64 of the fused [objective | gradient] buffer per evaluation; Ntotal = samples of all ranks.
 *   One completion poll per evaluation, nothing else crosses the host boundary.
 * TM:3252-3257 (scipy.optimize.minimize, method 'BFGS': gtol 1e-5 on max |g|, 200 n iterations, More-Thuente line
 * search with the Nocedal-Wright bracketing search as its fallback) as a host C++ loop (csrc/ttm_bfgs.h).
 * ttm_bfgs_minimize: generic front end; result (nullable, 5 doubles): {f, max |g|, iterations, evaluations, status:
 *   0 = converged, 1 = iteration limit, 2 = no acceptable step (precision loss), 3 = NaN}.
 * ttm_optimize_integrated: component k of an integrated-rectifier map minimised without leaving the library:
 *   J(c) = sums[0] / Ntotal + penalty(c), grad = sums[1..m] / Ntotal + penalty'(c) with the sums of
 *   ttm_objective_host (TM:3300-3380, 3435-3573) over the local samples, m = n_nonmon + n_mon <= 128 coefficients
 *   [nonmonotone | monotone] in x (host, start / result); regularization 0 none, 1 l1: sum lambda_i |c_i|, 2 l2:
 *   sum lambda_i c_i^2 (TM:3382-3431, 3575-3633; lambda: host, m doubles, NULL for 0).  work / counter as
 *   ttm_objective_host; sums_host / sums_dev / comm / Ntotal as ttm_optimize_separable.                                */
/* hipStreamSynchronize(stream) for bindings that hold the raw stream handle only */
int ttm_stream_synchronize(void* stream);
/* ttm_signal: *flag = value, ordered behind everything already enqueued on the stream.  With flag in pinned host memory
 * a host loop polls it instead of paying hipStreamSynchronize per objective evaluation (ttm_optimize_separable does). */
int ttm_signal(double* flag, double value, void* stream);
typedef int32_t (*ttm_objective_cb)(int32_t n, const double* x, double* f, double* g, void* user);
int ttm_lbfgsb_minimize(int32_t n, double* x, const double* lb, const double* ub, ttm_objective_cb fun, void* user,
                        int32_t maxiter, double* result);
struct ttm_comm;
int ttm_optimize_separable(const double* dPsi, int64_t ldp, int64_t N, int32_t m, const double* A, const double* b,
                           double Ntotal, double delta, const double* lb, const double* ub, double* x, double* work,
                           uint32_t* counter, double* sums_dev, double* sums_host, struct ttm_comm* comm, void* stream,
                           int32_t maxiter, double* result);
/* ttm_optimize_separable_batch: ntasks independent component problems of one ensemble (same N, Ntotal, delta; the
 * reference's fork pool over components, TM:2789-2845) on nthreads host threads, each with a HIP stream of its own
 * (created once per process and device), so that the reductions of different components overlap on the device and
 * their completion polls on the host.  Every task carries its own dPsi / A / b / bounds / x / work / counter /
 * sums_host (as ttm_optimize_separable; nothing may be shared between tasks) and receives rc and result.  The call
 * first waits for `stream` (the producer of the dPsi bases); it returns when every task is done, with the first
 * non-zero rc.  nthreads = 1 runs the tasks in order on `stream` itself.  No communicator: ranks that share the
 * samples of an ensemble call ttm_optimize_separable per component, in the same order on every rank.              */
typedef struct ttm_sep_task {
    const double* dPsi;          /* device, m rows of N doubles, row stride ldp; NULL: see xk below */
    int64_t ldp;
    int32_t m, rc;
    const double *A, *b, *lb, *ub;   /* host */
    double* x;                   /* host, m: start / result */
    double* work;                /* device, >= ttm_reduce_work_size(1 + m) doubles */
    uint32_t* counter;           /* device uint32[16], zero */
    double* sums_host;           /* pinned host, >= 2 + m doubles */
    double result[5];
    /* dPsi == NULL: the derivative basis is recomputed per evaluation (ttm_objective_sep_direct_marked) from */
    const double* xk;            /* device, the component's x_k column (N doubles) */
    const int32_t* kinds;        /* device, m */
    const double* pars;          /* device, 5 m */
    int32_t armed;               /* in: 1 = the rows of partial sums in `work` are armed (ttm_sentinel_fill) for this m and N - a
                                    previous call on the same `work` left them so; out: 1 when this call leaves them armed   */
} ttm_sep_task;
/* ttm_separable_reduce_l2 (host arithmetic only): the reduced separable problem with L2 regularisation (TM:3021-3050,
 * 3148-3169) from the (n + m) x (n + m) Gram matrix G of [Psi_nonmon | Psi_mon] (row-major, host): A (m x m) and sol (n x m,
 * c_nonmon = -sol c_mon), by Cholesky on the diagonally equilibrated normal equations with one refinement step.
 * TTM_E_UNSUPPORTED when a matrix is not positive definite (the caller then solves it its own way).                  */
int ttm_separable_reduce_l2(const double* G, int32_t n, int32_t m, double lam, double* A, double* sol);
int ttm_optimize_separable_batch(ttm_sep_task* tasks, int32_t ntasks, int64_t N, double Ntotal, double delta,
                                 int32_t nthreads, void* stream, int32_t maxiter);
int ttm_bfgs_minimize(int32_t n, double* x, ttm_objective_cb fun, void* user, int32_t maxiter, double* result);
/* ttm_optimize_integrated_batch: the same for the components of an integrated-rectifier map (one task per component
 * k; fields as the arguments of ttm_optimize_integrated; work >= ttm_reduce_work_size(1 + m) doubles per task).   */
typedef struct ttm_int_task {
    int32_t k, m, regularization, rc;
    const double* lambda;        /* host, m doubles (NULL without regularization) */
    double* x;                   /* host, m: start / result [nonmonotone | monotone] */
    double* work;                /* device */
    uint32_t* counter;           /* device uint32[16], zero */
    double* sums_host;           /* pinned host, >= 2 + m doubles */
    double result[5];
} ttm_int_task;
int ttm_optimize_integrated_batch(const ttm_program* p, ttm_int_task* tasks, int32_t ntasks, const double* Xsoa,
                                  int64_t ldx, int64_t N, double Ntotal, int32_t nthreads, void* stream, int32_t maxiter);
int ttm_optimize_integrated(const ttm_program* p, int32_t k, int32_t m, const double* Xsoa, int64_t ldx, int64_t N,
                            double Ntotal, int32_t regularization, const double* lambda, double* x, double* work,
                            uint32_t* counter, double* sums_dev, double* sums_host, struct ttm_comm* comm, void* stream,
                            int32_t maxiter, double* result);

/* ---- C1: the one collective of the path (RCCL over xGMI) ---------------------------------------------
 * Replaces nothing in the reference (its only parallelism is the fork pool of TM:2789-2845, whose "collective" is
 * pool.map collecting per-component results, TM:2829-2845); it is what SURVEY.md section 8e needs when samples
 * (objective / gradient sums, Gram matrices, column moments, bisection caps) or components (summed objective,
 * coefficient exchange) are spread over the GPUs of a node.  One communicator per process (= per GPU):
 *   ttm_comm_unique_id : rank 0 draws the 128-byte rendezvous id; the host distributes it to the other ranks
 *                        (any side channel: the Python class uses torch.distributed's store / broadcast);
 *   ttm_comm_create    : every rank, on its current HIP device; collective.  Returns within TTM_COMM_TIMEOUT_S
 *                        seconds (environment, default 120) whatever the other ranks do: a rank missing from the
 *                        rendezvous is an error (TTM_E_HIP), not a hang;
 *   ttm_comm_size      : rank and number of ranks of a communicator (what a benchmark reports as its RCCL width);
 *   ttm_allreduce_f64 / _i32 : in place on a device buffer, op TTM_OP_SUM / TTM_OP_MAX, enqueued on `stream`
 *                        (no host synchronisation; results are valid in stream order);
 *   ttm_comm_destroy.
 * RCCL is bound at run time; without it the functions return TTM_E_UNSUPPORTED (ttm_comm_last_error says why).   */
#define TTM_OP_SUM 0
#define TTM_OP_MAX 1
typedef struct ttm_comm ttm_comm;
const char* ttm_comm_last_error(void);
int ttm_comm_unique_id(void* id128);
int ttm_comm_create(const void* id128, int32_t rank, int32_t nranks, ttm_comm** out);
int ttm_comm_destroy(ttm_comm* comm);
int ttm_comm_size(const ttm_comm* comm, int32_t* rank, int32_t* nranks);
int ttm_allreduce_f64(ttm_comm* comm, double* buf, int64_t count, int32_t op, void* stream);
int ttm_allreduce_i32(ttm_comm* comm, int32_t* buf, int64_t count, int32_t op, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TTM_H */
