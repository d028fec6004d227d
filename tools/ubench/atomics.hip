// Micro-benchmark: throughput of 64-bit atomic adds on a small array (100,000 counters, the gather loop's `counters`),
// by memory scope and by which XCDs issue them.  hipcc --offload-arch=gfx950 -O3 -o /tmp/atomics tools/ubench/atomics.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

template <int SCOPE>   // 0: agent (what atomicAdd does), 1: workgroup scope, 2: agent scope but only XCD (blockIdx % 8 == 0) works
__global__ __launch_bounds__(256) void k(unsigned long long* c, uint32_t n, uint32_t per_thread, uint32_t by_xcd) {
    const uint32_t xcd = blockIdx.x & 7u;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t i = 0; i < per_thread; ++i) {
        uint32_t t = (uint32_t)(mix(tid * 977 + i) % n);
        if (by_xcd) t = (t & ~7u) | xcd;                     // counters = x (mod 8) only from XCD x
        if (t >= n) t = xcd;
        if (SCOPE == 1) __hip_atomic_fetch_add(&c[t], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(&c[t], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main() {
    const uint32_t n = 100000, per = 16;
    unsigned long long* c;
    hipMalloc(&c, n * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {64, 256, 1024, 4096}) {
        for (int variant = 0; variant < 4; ++variant) {
            std::vector<unsigned long long> init(n, 1ull << 40);
            hipMemcpy(c, init.data(), n * 8, hipMemcpyHostToDevice);
            const uint32_t by_xcd = variant >= 2;
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (variant == 0 || variant == 2) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, c, n, per, by_xcd);
                else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, c, n, per, by_xcd);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            std::vector<unsigned long long> out(n);
            hipMemcpy(out.data(), c, n * 8, hipMemcpyDeviceToHost);
            unsigned long long dec = 0;
            for (uint32_t i = 0; i < n; ++i) dec += (1ull << 40) - out[i];
            const double total = (double)blocks * 256 * per;
            printf("blocks %5d  %-28s %8.1f us  %7.2f G atomics/s  lost updates %lld\n", blocks,
                   variant == 0 ? "agent scope" : variant == 1 ? "workgroup scope (unsafe)" : variant == 2 ? "agent scope, rows by XCD" : "workgroup scope, rows by XCD",
                   best * 1e3, total / best / 1e6, (long long)(5 * total) - (long long)dec);
        }
    }
    return 0;
}
