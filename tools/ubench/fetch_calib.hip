// fetch_calib.hip -- what do rocprofv3's FETCH_SIZE / WRITE_SIZE count on gfx950, per access width?
//
// MI355X_MICROARCH.md calibrates ONE case: FETCH_SIZE reports half the bytes of a 16 B/lane coalesced streaming read.  The
// kernels of this library also read 8 B/lane (sorted u64 rows: overlap_wide_kernel, gather's row visits), 4 B/lane (query
// positions, posting lists), through `global_load ... lds` (the compare index, the overlap pass's slices), and with a
// lane stride (record lookups); VERDICT r03 item 4a asks for the ratio of each before a `traffic` figure is quoted.
//
// Every kernel below moves a KNOWN number of bytes of a buffer far larger than the 256 MiB Infinity Cache, once, with the
// access shape its name says.  Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and again with `--pmc WRITE_SIZE`
// (tools/prof_calib.sh); profiles/summarize.py lists the counter per kernel, tools/calib_table.py divides by the bytes.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/fetch_calib tools/ubench/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

namespace calib {

// grid-stride streaming read, W bytes per lane and step, consecutive lanes consecutive addresses; the sum keeps the loads alive
template <typename T>
__device__ __forceinline__ void read_body(const T* __restrict__ src, uint64_t n, uint32_t* sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const T v = src[i];
        const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
#pragma unroll
        for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc ^= w[k];
    }
    if (acc == 0x12345u) *sink = acc;
}
__global__ __launch_bounds__(256) void read_b32_kernel(const uint32_t* __restrict__ src, uint64_t n, uint32_t* sink) { read_body(src, n, sink); }
__global__ __launch_bounds__(256) void read_b64_kernel(const uint2* __restrict__ src, uint64_t n, uint32_t* sink) { read_body(src, n, sink); }
__global__ __launch_bounds__(256) void read_b128_kernel(const uint4* __restrict__ src, uint64_t n, uint32_t* sink) { read_body(src, n, sink); }

// the same stream through `global_load_dword ... lds` (one dword per lane straight into LDS, 256 B per wave-instruction)
__global__ __launch_bounds__(256) void read_lds_b32_kernel(const uint32_t* __restrict__ src, uint64_t n, uint32_t* sink) {
    __shared__ uint32_t lds[256];
    uint32_t acc = 0;
    const uint64_t per = (uint64_t)gridDim.x * 256;
    for (uint64_t base = (uint64_t)blockIdx.x * 256; base + 256 <= n; base += per) {
        const uint32_t* g = src + base + threadIdx.x;
        // M0 holds the LDS base of the wave's 256-byte piece; lanes land at consecutive dwords
        __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) uint32_t*)(lds + (threadIdx.x & ~63u)), 4, 0, 0);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        acc ^= lds[threadIdx.x];
        __syncthreads();
    }
    if (acc == 0x12345u) *sink = acc;
}

// one 8-byte load per lane, lanes `stride` elements apart (every lane its own cache line when stride >= 16): lookups
__global__ __launch_bounds__(256) void read_strided_b64_kernel(const uint64_t* __restrict__ src, uint64_t n_loads, uint64_t stride, uint32_t* sink) {
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_loads; i += (uint64_t)gridDim.x * 256) acc ^= src[i * stride];
    if (acc == 0x12345u) *sink = (uint32_t)acc;
}

// one wave reads ONE 512-byte row slice per step (lane x 8 B), rows 40 KB apart: the overlap pass's row visits
__global__ __launch_bounds__(256) void read_rowvisit_b64_kernel(const uint64_t* __restrict__ src, uint64_t n_rows, uint64_t row_words,
                                                               uint32_t visits, uint32_t* sink) {
    uint64_t acc = 0;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = (uint64_t)gridDim.x * 4;
    for (uint32_t v = 0; v < visits; ++v)
        for (uint64_t r = wave; r < n_rows; r += n_waves) acc ^= src[r * row_words + (uint64_t)v * 64 + lane];
    if (acc == 0x12345u) *sink = (uint32_t)acc;
}

// the overlap pass's visits as they are: a wave loads 512 B at a row's cursor and the cursor then moves on by `advance` bytes
// (what the visit consumed, ~45 hashes of the 64 loaded): the unconsumed tail is loaded again by the next visit
__global__ __launch_bounds__(256) void read_cursor_b64_kernel(const uint64_t* __restrict__ src, uint64_t n_rows, uint64_t row_words,
                                                             uint32_t visits, uint32_t advance_words, uint32_t* sink) {
    uint64_t acc = 0;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = (uint64_t)gridDim.x * 4;
    for (uint32_t v = 0; v < visits; ++v)
        for (uint64_t r = wave; r < n_rows; r += n_waves) acc ^= src[r * row_words + (uint64_t)v * advance_words + lane];
    if (acc == 0x12345u) *sink = (uint32_t)acc;
}

template <typename T>
__device__ __forceinline__ void write_body(T* __restrict__ dst, uint64_t n, uint32_t seed) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        T v;
        uint32_t* w = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
        for (unsigned k = 0; k < sizeof(T) / 4; ++k) w[k] = seed + (uint32_t)i + k;
        dst[i] = v;
    }
}
__global__ __launch_bounds__(256) void write_b32_kernel(uint32_t* __restrict__ dst, uint64_t n, uint32_t seed) { write_body(dst, n, seed); }
__global__ __launch_bounds__(256) void write_b64_kernel(uint2* __restrict__ dst, uint64_t n, uint32_t seed) { write_body(dst, n, seed); }
__global__ __launch_bounds__(256) void write_b128_kernel(uint4* __restrict__ dst, uint64_t n, uint32_t seed) { write_body(dst, n, seed); }

// 4-byte scatter: lane i writes element perm(i) (every store its own line): posting lists filled one entry at a time
__global__ __launch_bounds__(256) void write_scatter_b32_kernel(uint32_t* __restrict__ dst, uint64_t n_stores, uint64_t n_words, uint32_t seed) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_stores; i += (uint64_t)gridDim.x * 256)
        dst[(i * 0x9E3779B97F4A7C15ull >> 20) % n_words] = seed + (uint32_t)i;
}

}  // namespace calib

int main(int argc, char** argv) {
    const uint64_t bytes = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 2048ull) << 20;     // MiB; default 2 GiB (8 x the Infinity Cache)
    void *a = nullptr, *b = nullptr;
    uint32_t* sink = nullptr;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMalloc(&sink, 256));
    CHECK(hipMemset(a, 1, bytes));
    CHECK(hipMemset(b, 2, bytes));
    const int grid = 256 * 8;
    using namespace calib;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(read_b32_kernel, dim3(grid), dim3(256), 0, 0, (const uint32_t*)a, bytes / 4, sink);
        hipLaunchKernelGGL(read_b64_kernel, dim3(grid), dim3(256), 0, 0, (const uint2*)b, bytes / 8, sink);
        hipLaunchKernelGGL(read_b128_kernel, dim3(grid), dim3(256), 0, 0, (const uint4*)a, bytes / 16, sink);
        hipLaunchKernelGGL(read_lds_b32_kernel, dim3(grid), dim3(256), 0, 0, (const uint32_t*)b, bytes / 4, sink);
        // strided: bytes / 128 loads of 8 B, 128 B apart -> every load its own 128-byte line: 8 useful bytes of every line
        hipLaunchKernelGGL(read_strided_b64_kernel, dim3(grid), dim3(256), 0, 0, (const uint64_t*)a, bytes / 128, (uint64_t)16, sink);
        // row visits: rows of 40,000 bytes (5,000 u64), 4 visits of 512 B at the row's head
        hipLaunchKernelGGL(read_rowvisit_b64_kernel, dim3(grid), dim3(256), 0, 0, (const uint64_t*)b, bytes / 40000, (uint64_t)5000, 4u, sink);
        // 100 visits per 40 KB row, each 512 B wide, the cursor advancing 45 hashes: every row is walked end to end once
        hipLaunchKernelGGL(read_cursor_b64_kernel, dim3(grid), dim3(256), 0, 0, (const uint64_t*)a, bytes / 40000, (uint64_t)5000, 100u, 45u, sink);
        hipLaunchKernelGGL(write_b32_kernel, dim3(grid), dim3(256), 0, 0, (uint32_t*)a, bytes / 4, 7u);
        hipLaunchKernelGGL(write_b64_kernel, dim3(grid), dim3(256), 0, 0, (uint2*)b, bytes / 8, 7u);
        hipLaunchKernelGGL(write_b128_kernel, dim3(grid), dim3(256), 0, 0, (uint4*)a, bytes / 16, 7u);
        hipLaunchKernelGGL(write_scatter_b32_kernel, dim3(grid), dim3(256), 0, 0, (uint32_t*)b, bytes / 64, bytes / 4, 7u);
    }
    CHECK(hipDeviceSynchronize());
    // what each kernel moved, for tools/calib_table.py
    printf("{\"buffer_bytes\": %llu, \"kernels\": {"
           "\"read_b32_kernel\": {\"read\": %llu, \"shape\": \"4 B/lane coalesced stream\"}, "
           "\"read_b64_kernel\": {\"read\": %llu, \"shape\": \"8 B/lane coalesced stream\"}, "
           "\"read_b128_kernel\": {\"read\": %llu, \"shape\": \"16 B/lane coalesced stream\"}, "
           "\"read_lds_b32_kernel\": {\"read\": %llu, \"shape\": \"global_load_dword ... lds, 256 B per wave-instruction\"}, "
           "\"read_strided_b64_kernel\": {\"read\": %llu, \"lines\": %llu, \"shape\": \"8 B/lane, lanes 128 B apart (one line per lane)\"}, "
           "\"read_rowvisit_b64_kernel\": {\"read\": %llu, \"shape\": \"a wave reads 512 B of one row per step (8 B/lane), rows 40 KB apart, 4 steps per row\"}, "
           "\"read_cursor_b64_kernel\": {\"read\": %llu, \"shape\": \"the overlap pass's row visits: 512 B loaded at the row's cursor, cursor advances 45 hashes (360 B) per visit; known bytes = the part of the rows walked once\"}, "
           "\"write_b32_kernel\": {\"write\": %llu, \"shape\": \"4 B/lane coalesced stream\"}, "
           "\"write_b64_kernel\": {\"write\": %llu, \"shape\": \"8 B/lane coalesced stream\"}, "
           "\"write_b128_kernel\": {\"write\": %llu, \"shape\": \"16 B/lane coalesced stream\"}, "
           "\"write_scatter_b32_kernel\": {\"write\": %llu, \"lines\": %llu, \"shape\": \"4 B/lane scattered (one line per lane)\"}}}\n",
           (unsigned long long)bytes, (unsigned long long)bytes, (unsigned long long)bytes, (unsigned long long)bytes,
           (unsigned long long)(bytes / 4 / 256 / grid * 256 * grid * 4),
           (unsigned long long)(bytes / 128 * 8), (unsigned long long)(bytes / 128),
           (unsigned long long)(bytes / 40000 * 4 * 512),
           (unsigned long long)(bytes / 40000 * (99 * 45 + 64) * 8),
           (unsigned long long)bytes, (unsigned long long)bytes, (unsigned long long)bytes,
           (unsigned long long)(bytes / 64 * 4), (unsigned long long)(bytes / 64));
    return 0;
}
