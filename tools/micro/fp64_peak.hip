// Microbenchmark (tuning aid, not part of the product): sustained fp64 FMA issue rate of the chip, to turn the
// VALU instruction counts of the map kernels into a time floor.  hipcc -O3 --offload-arch=gfx950 fp64_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int CH>
__global__ __launch_bounds__(256) void k_fma(double* out, int iters, double a, double b) {
    double acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = threadIdx.x * 1e-3 + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_fma(acc[c], a, b);
    }
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += acc[c];
    if (s == 12345.678) out[0] = s;
}

template <int CH>
void run(int wg_per_cu, int iters) {
    double* out;
    hipMalloc(&out, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    k_fma<CH><<<grid, 256>>>(out, iters, 0.999999, 1e-9);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        k_fma<CH><<<grid, 256>>>(out, iters, 0.999999, 1e-9);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double fma_wave = (double)grid * 4 * iters * CH;            // wave-instructions
    const double tf = fma_wave * 64 * 2 / (best * 1e-3) / 1e12;
    // cycles per wave-instruction per SIMD if the clock were 2.4 GHz
    const double cyc = best * 1e-3 * 2.4e9 / (fma_wave / 1024.0);
    printf("chains %d, %d waves/SIMD: %.3f ms, %.1f TFLOP/s fp64, %.2f cycles@2.4GHz per wave-FMA per SIMD\n", CH, wg_per_cu, best, tf, cyc);
    hipFree(out);
}

int main() {
    run<1>(1, 200000); run<1>(2, 200000); run<1>(4, 200000); run<1>(8, 100000);
    run<2>(1, 100000); run<2>(2, 100000);
    run<4>(1, 100000); run<4>(2, 50000); run<4>(4, 50000);
    run<8>(1, 50000); run<8>(2, 50000);
    return 0;
}
