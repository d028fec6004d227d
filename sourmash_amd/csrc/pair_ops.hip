// pair_ops.hip -- single-pair and 1 x D set kernels on sorted u64 sketches.
//
// GPU counterparts of the reference's per-pair entry points
//   src/core/src/sketch/minhash.rs:539-558   count_common
//   src/core/src/sketch/minhash.rs:560-589   intersection (list)   + :1721-1763
//   src/core/src/sketch/minhash.rs:593-621   intersection_size, incl. the num (bottom-k) rule
//   src/core/src/sketch/minhash.rs:635-680   angular_similarity (integer sums; sqrt/acos stay on the host)
// and of the 1 x D loops of gather
//   src/sourmash/index/__init__.py:783-789   CounterGather.add   (overlap = |Q ∩ D_d| for every d)
//   src/sourmash/index/__init__.py:897-909   CounterGather.consume (c[d] -= |I ∩ D_d| for every d)
//
// A single pair is far too small to tile; one lane takes one hash of A and
// binary-searches B (both L2 resident), matches are flagged in place so an
// order-preserving compaction yields the sorted intersection.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_select.hpp>
#include "device_api.hpp"
#include "arena.hpp"
#include "pair_api.hpp"
#include "qindex.hpp"
#include "gather_api.hpp"

namespace smg {

__device__ __forceinline__ uint64_t lower_bound_dev(const uint64_t* __restrict__ a, uint64_t n, uint64_t x) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// flags[i] = 1 iff A[i] in B; matchB[i] = index in B.  sums[0] += matches,
// sums[1] += sum abundA[i]*abundB[j] over matches (if abundances given).
__global__ __launch_bounds__(256) void pair_match_kernel(const uint64_t* __restrict__ A, uint64_t na,
                                                         const uint64_t* __restrict__ B, uint64_t nb,
                                                         const uint64_t* __restrict__ abA,
                                                         const uint64_t* __restrict__ abB,
                                                         uint8_t* __restrict__ flags, unsigned long long* sums,
                                                         int invert) {
    unsigned long long cnt = 0, prod = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < na;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t x = A[i];
        const uint64_t j = lower_bound_dev(B, nb, x);
        const bool hit = j < nb && B[j] == x;
        if (flags) flags[i] = (hit != (invert != 0)) ? 1 : 0;
        if (hit) {
            ++cnt;
            if (abA && abB) prod += abA[i] * abB[j];
        }
    }
    // wave reduce, then one atomic per wave
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_down(cnt, off);
        prod += __shfl_down(prod, off);
    }
    if ((threadIdx.x & 63) == 0 && sums) {
        if (cnt) atomicAdd(&sums[0], cnt);
        if (prod) atomicAdd(&sums[1], prod);
    }
}

// The per-pair calls of the reference API (count_common / jaccard / similarity on two ~5,000-hash sketches) are latency, not
// work: one workgroup stages B in LDS (one coalesced trip), searches A's hashes there, reduces in LDS and publishes the
// count to a pinned host slot, sequence number last with system-scope release -- no memset, no read-back copy, and the
// host learns the result by polling that word instead of a stream synchronisation.
__global__ __launch_bounds__(1024) void pair_count_small_kernel(const uint64_t* __restrict__ A, uint32_t na,
                                                                const uint64_t* __restrict__ B, uint32_t nb,
                                                                unsigned long long* host_slot, unsigned long long seq) {
    extern __shared__ __attribute__((aligned(16))) uint64_t s_b[];
    __shared__ uint32_t s_part[16];
    for (uint32_t i = threadIdx.x; i < nb; i += 1024) s_b[i] = B[i];
    __syncthreads();
    uint32_t cnt = 0;
    for (uint32_t i = threadIdx.x; i < na; i += 1024) {
        const uint64_t x = A[i];
        uint32_t lo = 0, hi = nb;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_b[mid] < x) lo = mid + 1; else hi = mid;
        }
        cnt += lo < nb && s_b[lo] == x;
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long total = 0;
        for (int w = 0; w < 16; ++w) total += s_part[w];
        __hip_atomic_store(&host_slot[1], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host_slot[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t pair_count_small_launch(const uint64_t* A, uint64_t na, const uint64_t* B, uint64_t nb,
                                   unsigned long long* host_slot, unsigned long long seq, hipStream_t stream) {
    if (nb > PAIR_SMALL_MAX || na > 0xffffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pair_count_small_kernel, dim3(1), dim3(1024), (size_t)nb * 8, stream, A, (uint32_t)na, B, (uint32_t)nb,
                       host_slot, seq);
    return hipGetLastError();
}

// sums[2] += sum a^2 ; sums[3] += sum b^2
__global__ __launch_bounds__(256) void sumsq_kernel(const uint64_t* __restrict__ a, uint64_t n, unsigned long long* dst) {
    unsigned long long s = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        s += a[i] * a[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(dst, s);
}

// num (bottom-k) rule: count intersection hashes whose rank in A ∪ B is <= num
// (they survive the merge-and-truncate of minhash.rs:596-617).
__global__ __launch_bounds__(256) void num_rank_kernel(const uint64_t* __restrict__ I, uint64_t ni,
                                                       const uint64_t* __restrict__ A, uint64_t na,
                                                       const uint64_t* __restrict__ B, uint64_t nb, uint64_t num,
                                                       unsigned long long* dst) {
    unsigned long long cnt = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ni; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t x = I[i];
        const uint64_t ra = lower_bound_dev(A, na, x) + 1, rb = lower_bound_dev(B, nb, x) + 1;  // x is in both
        const uint64_t rank_union = ra + rb - (i + 1);
        if (rank_union <= num) ++cnt;
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(dst, cnt);
}

// ---- overlap[d] (op == 0: =, op == 1: -=) |Q ∩ D_d| for every row d of a CSR database ----------------------------------
// The query gets a first-level table (qindex.hpp) built on the stream in front of the pass, so a database element
// costs three load instructions instead of the log2(nq) dependent probes of a binary search.  Everything is
// stream-ordered: the table geometry is worked out on the device from Q[nq - 1].
struct QIndexHeader {
    QIndex qi;
    uint32_t buckets;
};
constexpr size_t QIH_BYTES = 64;                              // header, then the padded copy of Q, then the table

__global__ __launch_bounds__(256) void qindex_setup_kernel(const uint64_t* __restrict__ Q, uint64_t nq, uint8_t* scratch) {
    uint64_t* pad = reinterpret_cast<uint64_t*>(scratch + QIH_BYTES);
    const uint64_t qmax = Q[nq - 1];
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq + 4; i += (uint64_t)gridDim.x * blockDim.x)
        pad[i] = i < nq ? Q[i] : qmax;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        QIndexHeader* h = reinterpret_cast<QIndexHeader*>(scratch);
        h->qi.Q = pad;
        h->qi.nq = nq;
        h->qi.T = reinterpret_cast<const uint32_t*>(pad + nq + 4);
        h->qi.qmax = qmax;
        h->qi.rec = nullptr;                                        // this one-pass form keeps the two-level lookup
        qindex_geometry(nq, qmax, &h->qi.shift, &h->buckets);
    }
}

__global__ __launch_bounds__(256) void qindex_table_kernel(uint8_t* scratch) {
    const QIndexHeader* h = reinterpret_cast<const QIndexHeader*>(scratch);
    qindex_fill_bucket(h->qi.Q, h->qi.nq, h->qi.shift, h->buckets, const_cast<uint32_t*>(h->qi.T),
                       blockIdx.x * blockDim.x + threadIdx.x);
}

// one wave per row, 64 elements per step
__global__ __launch_bounds__(256) void overlap_vector_kernel(const QIndexHeader* __restrict__ hdr,
                                                             const uint64_t* __restrict__ hashes,
                                                             const uint64_t* __restrict__ offsets, uint64_t ndb,
                                                             unsigned long long* __restrict__ overlap, int op) {
    const QIndex qi = hdr->qi;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    for (uint64_t d = wave; d < ndb; d += n_waves) {
        const uint64_t lo = offsets[d], hi = offsets[d + 1];
        if (op == 1 && overlap[d] == 0) continue;          // dropped from the counter (index/__init__.py:908-909)
        unsigned long long cnt = 0;
        for (uint64_t i = lo + lane; i < hi; i += 64) cnt += q_find(qi, hashes[i]) != NONE32 ? 1 : 0;
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
        if (lane == 0) {
            if (op == 0) overlap[d] = cnt;
            else overlap[d] = cnt >= overlap[d] ? 0 : overlap[d] - cnt;
        }
    }
}

// packed arg-max with the reference tie-break (highest count, then lowest index):
// key = (count << 32) | ~index  (counts < 2^32), reduced with max.
__global__ __launch_bounds__(256) void argmax_kernel(const unsigned long long* __restrict__ overlap, uint64_t ndb,
                                                     uint64_t index_base, unsigned long long* __restrict__ best) {
    unsigned long long k = 0;
    for (uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; d < ndb; d += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long c = overlap[d];
        if (c) {
            const unsigned long long key = (c << 32) | (0xffffffffull & ~(unsigned long long)(index_base + d));
            k = key > k ? key : k;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_down(k, off);
        k = o > k ? o : k;
    }
    if ((threadIdx.x & 63) == 0 && k) atomicMax(best, k);
}

static unsigned grid_for(uint64_t n, unsigned per = 256, unsigned cap = 2048) {
    const uint64_t b = (n + per - 1) / per;
    return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

hipError_t pair_match_launch(const uint64_t* A, uint64_t na, const uint64_t* B, uint64_t nb, const uint64_t* abA,
                             const uint64_t* abB, uint8_t* flags, unsigned long long* sums, int invert,
                             hipStream_t stream) {
    if (na == 0 || nb == 0) {
        if (flags && na) return hipMemsetAsync(flags, invert ? 1 : 0, na, stream);
        return hipSuccess;
    }
    hipLaunchKernelGGL(pair_match_kernel, dim3(grid_for(na)), dim3(256), 0, stream, A, na, B, nb, abA, abB, flags, sums,
                       invert);
    return hipGetLastError();
}

hipError_t sumsq_launch(const uint64_t* a, uint64_t n, unsigned long long* dst, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n)), dim3(256), 0, stream, a, n, dst);
    return hipGetLastError();
}

hipError_t num_rank_launch(const uint64_t* I, uint64_t ni, const uint64_t* A, uint64_t na, const uint64_t* B,
                           uint64_t nb, uint64_t num, unsigned long long* dst, hipStream_t stream) {
    if (ni == 0) return hipSuccess;
    hipLaunchKernelGGL(num_rank_kernel, dim3(grid_for(ni)), dim3(256), 0, stream, I, ni, A, na, B, nb, num, dst);
    return hipGetLastError();
}

size_t select_temp_bytes(uint64_t n) {
    size_t bytes = 0;
    (void)rocprim::select(nullptr, bytes, (uint64_t*)nullptr, (uint8_t*)nullptr, (uint64_t*)nullptr,
                          (uint64_t*)nullptr, (size_t)(n ? n : 1), (hipStream_t)0);
    return bytes + 256;
}

hipError_t select_flagged(const uint64_t* in, const uint8_t* flags, uint64_t n, uint64_t* out, uint64_t* d_n_out,
                          void* temp, size_t temp_bytes, hipStream_t stream) {
    if (n == 0) return hipMemsetAsync(d_n_out, 0, 8, stream);
    return rocprim::select(temp, temp_bytes, in, flags, out, d_n_out, (size_t)n, stream);
}

hipError_t overlap_vector_launch(const uint64_t* Q, uint64_t nq, const uint64_t* hashes, const uint64_t* offsets,
                                 uint64_t ndb, unsigned long long* overlap, int op, hipStream_t stream) {
    if (ndb == 0) return hipSuccess;
    if (nq == 0) {                                        // nothing in common with anything
        return op == 0 ? hipMemsetAsync(overlap, 0, ndb * 8, stream) : hipSuccess;
    }
    if (nq >= NONE32) return hipErrorInvalidValue;
    static const bool no_ranges = [] { const char* e = getenv("SMG_OVERLAP"); return e && !strcmp(e, "rows"); }();
    if (!no_ranges && nq >= OVERLAP_RANGES_MIN_NQ && ndb >= OVERLAP_RANGES_MIN_ROWS && ndb < NONE32) {
        // overlap.hip: the query streams through LDS in ranges.  hipErrorNotSupported: neither streaming form has room for this
        // query's ranges (or the collection holds 2^32 hashes or more) -- the one-wave-per-row kernel below takes anything
        const hipError_t es = overlap_ranges_launch(Q, nq, hashes, offsets, ndb, overlap, op, stream);
        if (es != hipErrorNotSupported) return es;
    }
    // scratch: header, padded query, table of at most 2 * nq + 2 entries; allocated and released in stream order
    uint8_t* scratch = nullptr;
    const size_t bytes = QIH_BYTES + (nq + 4) * 8 + (2 * nq + 4) * 4;
    hipError_t e = arena_alloc((void**)&scratch, bytes, stream);
    if (e != hipSuccess) return e;
    const uint64_t copy_blocks = (nq + 4 + 255) / 256;
    hipLaunchKernelGGL(qindex_setup_kernel, dim3((unsigned)(copy_blocks < 1024 ? copy_blocks : 1024)), dim3(256), 0, stream, Q,
                       nq, scratch);
    hipLaunchKernelGGL(qindex_table_kernel, dim3((unsigned)((2 * nq + 2 + 255) / 256)), dim3(256), 0, stream, scratch);
    const uint64_t blocks = (ndb + 3) / 4;                // one wave per dataset, capped
    hipLaunchKernelGGL(overlap_vector_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, stream,
                       reinterpret_cast<const QIndexHeader*>(scratch), hashes, offsets, ndb, overlap, op);
    e = hipGetLastError();
    arena_free(scratch, stream);
    return e;
}

// one workgroup per destination row: dst[dst_off[i] ..) = src[src_off[rows[i]] ..), 8-byte coalesced copies
__global__ __launch_bounds__(256) void copy_rows_kernel(const uint64_t* __restrict__ src, const uint64_t* __restrict__ src_off,
                                                        const uint64_t* __restrict__ rows, uint64_t n_rows,
                                                        const uint64_t* __restrict__ dst_off, uint64_t* __restrict__ dst) {
    for (uint64_t i = blockIdx.x; i < n_rows; i += gridDim.x) {
        const uint64_t lo = src_off[rows[i]], len = src_off[rows[i] + 1] - lo, out = dst_off[i];
        for (uint64_t t = threadIdx.x; t < len; t += blockDim.x) dst[out + t] = src[lo + t];
    }
}

hipError_t copy_rows_launch(const uint64_t* src, const uint64_t* src_off, const uint64_t* rows, uint64_t n_rows,
                            const uint64_t* dst_off, uint64_t* dst, hipStream_t stream) {
    if (n_rows == 0) return hipSuccess;
    hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)(n_rows < 65536 ? n_rows : 65536)), dim3(256), 0, stream, src, src_off,
                       rows, n_rows, dst_off, dst);
    return hipGetLastError();
}

hipError_t argmax_launch(const unsigned long long* overlap, uint64_t ndb, uint64_t index_base,
                         unsigned long long* best, hipStream_t stream) {
    if (ndb == 0) return hipSuccess;
    hipLaunchKernelGGL(argmax_kernel, dim3(grid_for(ndb)), dim3(256), 0, stream, overlap, ndb, index_base, best);
    return hipGetLastError();
}

}  // namespace smg
