// Microbenchmark (tuning aid, not part of the product): issue cost of the VALU instructions the map kernels are made
// of, per wave-instruction per SIMD, at 1 / 2 / 4 waves per SIMD with 8 independent chains per wave.
//   hipcc -O3 --offload-arch=gfx950 -o valu_costs valu_costs.hip && ./valu_costs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CH 8
enum Op { FMA64, ADD64, MUL64, MAX64, FLOOR64, RNDNE64, LDEXP64, RCP64, CVT_RT, FRACT64, CMPSEL64, ADD32, MAD24, MED3I, FMA64_ADD32, FMA64_2ADD32,
          FMA32, CVT_I32_F64, FMA64_CVT };

template <int OP>
__global__ __launch_bounds__(256) void k_op(double* out, int iters, double a, double b) {
    double acc[CH];
    int ia[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { acc[c] = threadIdx.x * 1e-3 + c + 1.5; ia[c] = threadIdx.x + c; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (OP == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(a), "v"(b));
            if (OP == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[c]) : "v"(b));
            if (OP == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(acc[c]) : "v"(a));
            if (OP == MAX64) asm volatile("v_max_f64 %0, %0, %1" : "+v"(acc[c]) : "v"(b));
            if (OP == FLOOR64) asm volatile("v_floor_f64 %0, %0" : "+v"(acc[c]));
            if (OP == RNDNE64) asm volatile("v_rndne_f64 %0, %0" : "+v"(acc[c]));
            if (OP == LDEXP64) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(acc[c]) : "v"(ia[c]));
            if (OP == RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(acc[c]));
            if (OP == FRACT64) asm volatile("v_fract_f64 %0, %0" : "+v"(acc[c]));
            if (OP == CVT_RT) asm volatile("v_cvt_i32_f64 %1, %0\n v_cvt_f64_i32 %0, %1" : "+v"(acc[c]), "+v"(ia[c]));
            if (OP == CVT_I32_F64) asm volatile("v_cvt_i32_f64 %1, %0" : "+v"(acc[c]), "+v"(ia[c]));
            if (OP == CMPSEL64) asm volatile("v_cmp_lt_f64 vcc, %0, %2\n v_cndmask_b32 %1, %1, %3, vcc" : "+v"(acc[c]), "+v"(ia[c]) : "v"(b), "v"(ia[(c + 1) % CH]) : "vcc");
            if (OP == ADD32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[c]) : "v"(ia[(c + 1) % CH]));
            if (OP == MAD24) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(ia[c]) : "v"(ia[(c + 1) % CH]));
            if (OP == MED3I) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(ia[c]) : "v"(ia[(c + 1) % CH]), "v"(ia[(c + 2) % CH]));
            if (OP == FMA32) { float f = __int_as_float(ia[c]); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f) : "v"(1.0f)); ia[c] = __float_as_int(f); }
            if (OP == FMA64_ADD32) asm volatile("v_fma_f64 %0, %0, %2, %3\n v_add_u32 %1, %1, %4" : "+v"(acc[c]), "+v"(ia[c]) : "v"(a), "v"(b), "v"(ia[(c + 1) % CH]));
            if (OP == FMA64_2ADD32) asm volatile("v_fma_f64 %0, %0, %2, %3\n v_add_u32 %1, %1, %4\n v_add_u32 %1, %1, %4" : "+v"(acc[c]), "+v"(ia[c]) : "v"(a), "v"(b), "v"(ia[(c + 1) % CH]));
            if (OP == FMA64_CVT) asm volatile("v_fma_f64 %0, %0, %2, %3\n v_cvt_i32_f64 %1, %0" : "+v"(acc[c]), "+v"(ia[c]) : "v"(a), "v"(b));
        }
    }
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += acc[c] + ia[c];
    if (s == 12345.678) out[0] = s;
}

template <int OP>
void run(const char* name, int ninst, int wg_per_cu, int iters) {
    double* out;
    hipMalloc(&out, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    for (int r = 0; r < 3; ++r) k_op<OP><<<grid, 256>>>(out, iters, 0.999999, 1e-9);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        k_op<OP><<<grid, 256>>>(out, iters, 0.999999, 1e-9);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double groups = (double)grid * 4 * iters * CH;              // wave-level op groups
    const double cyc = best * 1e-3 * 2.4e9 / (groups / 1024.0);
    printf("%-16s %d waves/SIMD: %8.3f ms  %6.2f cycles@2.4GHz per group of %d instruction(s) per SIMD\n", name, wg_per_cu, best, cyc, ninst);
    hipFree(out);
}

#define RUN(OP, n) run<OP>(#OP, n, 1, 20000); run<OP>(#OP, n, 2, 20000); run<OP>(#OP, n, 4, 20000);
int main() {
    // keep the clock up first
    run<FMA64>("warm", 1, 4, 400000);
    RUN(FMA64, 1) RUN(ADD64, 1) RUN(MUL64, 1) RUN(MAX64, 1) RUN(FLOOR64, 1) RUN(RNDNE64, 1) RUN(LDEXP64, 1) RUN(RCP64, 1) RUN(FRACT64, 1)
    RUN(CVT_I32_F64, 1) RUN(CVT_RT, 2) RUN(CMPSEL64, 2) RUN(ADD32, 1) RUN(MAD24, 1) RUN(MED3I, 1) RUN(FMA32, 1) RUN(FMA64_ADD32, 2) RUN(FMA64_2ADD32, 3) RUN(FMA64_CVT, 2)
    return 0;
}
