// device_sort.hip -- sort + unique (+ multiplicities) of the kept hashes.
//
// GPU counterpart of inserting the kept hashes into the reference's sorted set
// (src/core/src/sketch/minhash.rs:313-383 Vec insert, :1237-1291 BTreeSet
// insert; abundance = number of insertions of the same hash).  The kept hashes
// are ~1/scaled of the k-mers, so this stage moves ~0.8 % of the bytes the
// sketch kernel reads; it uses rocPRIM's device radix sort (AMD's native
// primitive library, shipped with ROCm) restricted to the significant key bits
// (hashes <= max_hash < 2^54 at scaled = 1000 need 54 of 64 bits), followed by
// run-length encoding, which yields the unique keys and their counts at once.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include "device_api.hpp"

namespace smg {

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static size_t radix_temp_bytes(uint64_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_keys(nullptr, bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (size_t)n, 0u, 64u,
                                   (hipStream_t)0);
    return bytes;
}

static size_t rle_temp_bytes(uint64_t n) {
    size_t bytes = 0;
    (void)rocprim::run_length_encode(nullptr, bytes, (uint64_t*)nullptr, (unsigned int)n, (uint64_t*)nullptr,
                                     (uint64_t*)nullptr, (uint64_t*)nullptr, (hipStream_t)0);
    return bytes;
}

// layout of temp: [sorted keys: n u64][counts scratch: n u64][primitive temp]
size_t sort_unique_temp_bytes(uint64_t n) {
    if (n == 0) n = 1;
    const size_t prim = radix_temp_bytes(n) > rle_temp_bytes(n) ? radix_temp_bytes(n) : rle_temp_bytes(n);
    return align_up(n * 8, 256) * 2 + align_up(prim, 256) + 256;
}

hipError_t sort_unique(uint64_t* d_keys, uint64_t n, uint64_t* d_out, uint64_t* d_counts, uint64_t* d_n_out,
                       void* d_temp, size_t temp_bytes, int bits, hipStream_t stream) {
    if (n == 0) return hipMemsetAsync(d_n_out, 0, 8, stream);
    if (n > 0xffffffffull) return hipErrorInvalidValue;   // run_length_encode takes a 32-bit size
    if (temp_bytes < sort_unique_temp_bytes(n)) return hipErrorInvalidValue;
    char* base = (char*)d_temp;
    uint64_t* sorted = (uint64_t*)base;
    uint64_t* counts_scratch = (uint64_t*)(base + align_up(n * 8, 256));
    void* prim = base + 2 * align_up(n * 8, 256);
    size_t prim_bytes = temp_bytes - 2 * align_up(n * 8, 256);
    if (bits < 1) bits = 1;
    if (bits > 64) bits = 64;
    hipError_t e = rocprim::radix_sort_keys(prim, prim_bytes, d_keys, sorted, (size_t)n, 0u, (unsigned)bits, stream);
    if (e != hipSuccess) return e;
    prim_bytes = temp_bytes - 2 * align_up(n * 8, 256);
    e = rocprim::run_length_encode(prim, prim_bytes, sorted, (unsigned int)n, d_out,
                                   d_counts ? d_counts : counts_scratch, d_n_out, stream);
    return e;
}

// Many small hash lists sorted as ONE: list s (src[seg.src, seg.src + seg.n)) is copied to dst[seg.dst, ...) with its number in
// the bits above `hbits` (hashes <= max_hash need fewer than 64 bits), so that one radix sort + one run-length encode of all of
// dst orders every list by itself -- the per-list sorts of a batch of genomes were a thousand launches of a few microseconds.
namespace {
__global__ __launch_bounds__(256) void tag_gather_kernel(const uint64_t* __restrict__ src, const TagSegment* __restrict__ segs, uint64_t* __restrict__ dst, int hbits) {
    const TagSegment g = segs[blockIdx.x];
    const uint64_t tag = (uint64_t)blockIdx.x << hbits;
    for (uint64_t i = threadIdx.x; i < g.n; i += 256u) dst[g.dst + i] = src[g.src + i] | tag;
}
}  // namespace

hipError_t tag_gather_launch(const uint64_t* d_src, const TagSegment* d_segs, uint32_t n_segs, uint64_t* d_dst, int hbits, hipStream_t stream) {
    if (n_segs == 0) return hipSuccess;
    hipLaunchKernelGGL(tag_gather_kernel, dim3(n_segs), dim3(256), 0, stream, d_src, d_segs, d_dst, hbits);
    return hipGetLastError();
}

}  // namespace smg
