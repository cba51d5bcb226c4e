// ttm_int.h - internal interface of csrc/ttm_int.hip (the kernels of integrated-rectifier maps whose components all have a
// dense B set, csrc/ttm_dense.h) to csrc/ttm_kernels.hip, which owns the C ABI, the argument checks and the launch
// bookkeeping.  A launch geometry is handed in; the name of the kernel that ran comes back for ttm_last_kernel().
#pragma once

#include <stddef.h>
#include <stdint.h>

#include "../../include/ttm.h"

struct DevProg;

namespace ttm_int {

// can the components [k0, k1) of this program run through the dense integrated kernels?
bool usable(const ttm_program* p, int k0, int k1);

// use_x != 0: through the X programs (csrc/ttm_xprog.h) when every component of the range has one - "k_int_forward" (the kernel
// that walks the term tables per sample reports "k_int_forward<walk>"), "k_int_root_x<...>" 
// forward map (TM:2391-2567) with the fused log-determinant / sum of squares of ttm_forward; chunk: components per
// workgroup (grid.y = chunks; the fused outputs need all components in one)
int forward(const ttm_program* p, const DevProg& P, int k0, int k1, const double* coef, const double* fold, const double* Xsoa,
            int64_t ldx, int64_t N, double* Zsoa, int64_t ldz, double* logdet, const double* sigma, double* sumsq, int grid,
            int chunk, int bd, size_t lds, int use_x, void* stream, const char** kernel_name);

// root search of ttm_inverse_bisect (newton = 0: the reference's bisection sequence, TM:3842-3976) / ttm_inverse_newton
int root(const ttm_program* p, const DevProg& P, int k0, int k1, const double* coef, const double* fold, const double* Zsoa,
         int64_t ldz, double* Xsoa, int64_t ldx, int64_t N, int32_t* iters, const int32_t* cap, int newton, int grid, int bd,
         size_t lds, int use_x, void* stream, const char** kernel_name);

// objective + gradient partial sums of one component (the k_objective launch of ttm_objective / ttm_objective_host_marked)
int objective(const ttm_program* p, const DevProg& P, int k, const double* coef_k, const double* fold_k, const double* Xsoa,
              int64_t ldx, int64_t N, int nscr, int nacc, double* partial, unsigned int* counter, double* out, double* flag,
              double mark, int grid, int bd, size_t lds, void* stream, const char** kernel_name);

// the same sums through the component's X program (csrc/ttm_xprog.h: every factor value once per sample, a sum per lane): one launch
// per evaluation, coefficients from the host (h_coef_k: kernel argument) or from device memory (d_coef_k); finished by the
// workgroup that draws the last ticket (out != nullptr) - grids of up to TTM_RED_BLOCKS workgroups
bool has_xprog(const ttm_program* p, int k);
size_t objective_x_lds(const ttm_program* p, int k, int bd);
int objective_x(const ttm_program* p, const DevProg& P, int k, const double* h_coef_k, const double* d_coef_k, int ncoef, const double* Xsoa,
                int64_t ldx, int64_t N, double* partial, unsigned int* counter, double* out, double* flag, double mark,
                int grid, int bd, void* stream, const char** kernel_name);

}  // namespace ttm_int
