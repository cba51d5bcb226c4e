// Is device memory host-writable on this platform (fine-grained VRAM over the PCIe BAR), and what does a host -> device -> host
// round trip cost through it?  A persistent workgroup (or NWG of them) polls a mailbox word in DEVICE memory that the host writes
// directly (posted PCIe writes), and answers into page-locked host memory.  Compare: one kernel launch + completion poll per
// round trip.   hipcc -O2 --offload-arch=gfx950 tools/micro/mailbox.cpp -o /tmp/mailbox && /tmp/mailbox [nwg]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <immintrin.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void server(volatile unsigned long long* box, unsigned long long* answer, unsigned int* arrive, int rounds) {
    // every workgroup polls the mailbox; the last to arrive of a round (ticket) answers; exits after `rounds` or on a quit word
    __shared__ unsigned long long seen;
    for (int r = 1; r <= rounds; ++r) {
        if (threadIdx.x == 0) {
            unsigned long long v;
            long spins = 0;
            do { v = __hip_atomic_load((unsigned long long*)box, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (v < (unsigned long long)r && v != ~0ull && ++spins < (1l << 26));
            seen = v;
        }
        __syncthreads();
        if (seen == ~0ull || seen < (unsigned long long)r) return;
        if (threadIdx.x == 0) {
            const unsigned int t = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == gridDim.x * (unsigned int)r - 1u) __hip_atomic_store(answer, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
    }
}
__global__ void echo(unsigned long long* answer, unsigned long long v) { if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(answer, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

int main(int argc, char** argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 1;
    const int rounds = 2000;
    unsigned long long *box = nullptr, *answer = nullptr;
    unsigned int* arrive = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&box, 4096, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 1;
    CHECK(hipHostMalloc((void**)&answer, 4096, hipHostMallocDefault));
    CHECK(hipMalloc((void**)&arrive, 64));
    CHECK(hipMemset(arrive, 0, 64));
    CHECK(hipMemset(box, 0, 4096));
    CHECK(hipDeviceSynchronize());
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, box) == hipSuccess) printf("box: type %d hostPointer %p devicePointer %p\n", (int)at.type, at.hostPointer, at.devicePointer);
    fflush(stdout);
    *answer = 0;
    // host write straight into device memory (a segfault here = not host-accessible)
    std::atomic<unsigned long long>* hb = reinterpret_cast<std::atomic<unsigned long long>*>(box);
    hb->store(0, std::memory_order_release);
    printf("host store into device memory: ok\n"); fflush(stdout);
    std::atomic<unsigned long long>* ha = reinterpret_cast<std::atomic<unsigned long long>*>(answer);
    hipLaunchKernelGGL(server, dim3(nwg), dim3(256), 0, 0, (volatile unsigned long long*)box, answer, arrive, rounds);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 1; r <= rounds; ++r) {
        hb->store((unsigned long long)r, std::memory_order_release);
        _mm_sfence();                                    // (write-combining mapping: push the store out)
        long spins = 0;
        while (ha->load(std::memory_order_acquire) != (unsigned long long)r && ++spins < (1l << 28)) {}
        if (spins >= (1l << 28)) { printf("no answer in round %d\n", r); hb->store(~0ull); break; }
    }
    auto t1 = std::chrono::steady_clock::now();
    CHECK(hipDeviceSynchronize());
    printf("mailbox round trip, %d polling workgroup(s): %.2f us\n", nwg, std::chrono::duration<double, std::micro>(t1 - t0).count() / rounds);
    // one launch + completion poll per round trip
    *answer = 0;
    t0 = std::chrono::steady_clock::now();
    for (int r = 1; r <= rounds; ++r) {
        hipLaunchKernelGGL(echo, dim3(nwg), dim3(256), 0, 0, answer, (unsigned long long)r);
        while (ha->load(std::memory_order_acquire) != (unsigned long long)r) {}
    }
    t1 = std::chrono::steady_clock::now();
    CHECK(hipDeviceSynchronize());
    printf("launch + poll round trip: %.2f us\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / rounds);
    return 0;
}
