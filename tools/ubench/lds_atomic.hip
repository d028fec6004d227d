// What an LDS atomic costs on gfx950 (round 5: the abundance join, csrc/abund_pairs.hip, adds a u64 product and a u32 count per
// match to a 64 x 65 tile of accumulators in LDS; three formulations of that join landed within 20 % of one another).
//
// One 1,024-thread workgroup per CU; every lane issues N updates to a 4,160-slot accumulator tile: the same two instructions the
// join issues (ds_add_u64 + ds_add_u32), or one of them, or a plain read-add-write of the same cells; addresses consecutive over the
// lanes of a wave (no bank conflicts), strided by 65 slots (the join's rows), or random.  Prints lanes per clock and CU.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench/lds_atomic tools/ubench/lds_atomic.hip    (built here, run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int THREADS = 1024, SLOTS = 64 * 65, N = 2048;

// OP 0: u64 + u32 atomics   1: u64 atomic   2: u32 atomic   3: plain read-add-write of both (racy: cost only)   4: u32 atomic, 16 lanes active
// PAT 0: consecutive lanes -> consecutive slots   1: lane -> row (stride 65)   2: random slot
template <int OP, int PAT>
__global__ __launch_bounds__(THREADS) void kern(unsigned long long* out, uint32_t seed) {
    __shared__ unsigned long long s_p[SLOTS];
    __shared__ uint32_t s_c[SLOTS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < SLOTS; i += THREADS) { s_p[i] = 0; s_c[i] = 0; }
    __syncthreads();
    uint32_t x = seed ^ (uint32_t)(tid * 2654435761u) ^ (blockIdx.x * 40503u);
    const bool active = OP != 4 || (lane & 3) == 0;
    for (int i = 0; i < N; ++i) {
        x = x * 1664525u + 1013904223u;
        uint32_t slot;
        if (PAT == 0) slot = (uint32_t)((wave * 259 + i * 67) % (SLOTS - 64)) + (uint32_t)lane;
        else if (PAT == 1) slot = (uint32_t)lane * 65u + (uint32_t)((wave * 5 + i) & 63);
        else slot = (x >> 8) % SLOTS;
        if (!active) continue;
        if (OP == 0 || OP == 1) atomicAdd(&s_p[slot], (unsigned long long)x);
        if (OP == 0 || OP == 2 || OP == 4) atomicAdd(&s_c[slot], 1u);
        if (OP == 3) { s_p[slot] += x; s_c[slot] += 1u; }
    }
    __syncthreads();
    unsigned long long acc = 0;
    for (int i = tid; i < SLOTS; i += THREADS) acc += s_p[i] + s_c[i];
    if (acc == 0x123456789abcull) out[0] = acc;      // keep everything alive
}

template <int OP, int PAT>
static void run(const char* what, unsigned long long* d_out, int cus) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((kern<OP, PAT>), dim3(cus), dim3(THREADS), 0, 0, d_out, 1u);
    (void)hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((kern<OP, PAT>), dim3(cus), dim3(THREADS), 0, 0, d_out, 2u + r);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    const double lanes = (double)THREADS * N * (OP == 4 ? 0.25 : 1.0), clocks = ms * 1e-3 * 2.4e9;
    printf("%-58s %7.3f ms  %6.2f lane-updates / clock / CU  (%5.1f clocks per wave instruction pair or single)\n", what, ms, lanes / clocks,
           clocks / ((double)THREADS / 64 * N));
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    unsigned long long* d_out;
    (void)hipMalloc(&d_out, 64);
    printf("%d CUs, one workgroup of %d threads each, %d updates per lane; clocks at 2.4 GHz\n", cus, THREADS, N);
    run<0, 0>("ds_add_u64 + ds_add_u32, consecutive slots", d_out, cus);
    run<1, 0>("ds_add_u64, consecutive slots", d_out, cus);
    run<2, 0>("ds_add_u32, consecutive slots", d_out, cus);
    run<3, 0>("read-add-write u64 + u32, consecutive slots", d_out, cus);
    run<0, 1>("ds_add_u64 + ds_add_u32, lane -> row (stride 65)", d_out, cus);
    run<0, 2>("ds_add_u64 + ds_add_u32, random slots", d_out, cus);
    run<2, 2>("ds_add_u32, random slots", d_out, cus);
    run<3, 2>("read-add-write u64 + u32, random slots", d_out, cus);
    run<4, 2>("ds_add_u32, random slots, 16 of 64 lanes active", d_out, cus);
    return 0;
}
