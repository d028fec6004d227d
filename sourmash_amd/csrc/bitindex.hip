// bitindex.hip -- dense path of the all-pairs compare: sketches as bit rows over the collection's
// own hash dictionary, intersections as popcount(AND).
//
// Same results as compare.hip (u32 |A ∩ B| for every pair; reference: minhash.rs:539-558 count_common),
// different cost model.  The merge walk costs ~(n_i + n_j) steps per pair whatever the data; when the
// collection is "dense" -- its U distinct hashes are at most a few hundred times the mean sketch size,
// as for related genomes or any collection drawn from a common pool (BASELINE configs C3/C4: U = 50,000,
// n = 5,000) -- a sketch is better stored as U bits, and a 64 x 64 tile of pairs costs U/32 x 4096
// (AND + popcount) in registers with every bitmap word fetched once per tile.  smgpu_bitindex_new
// decides from the measured U (dictionary = device radix sort + run-length encode of all hashes) and
// returns NULL for sparse collections, for which the merge kernel stays the right tool.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "device_api.hpp"

namespace smg {

constexpr int BT = 64;            // tile edge (sketches)
constexpr int BKC = 32;           // bitmap words staged per k-step
constexpr int BSTRIDE = BKC + 4;  // LDS row stride in words: 36*t mod 64 is a distinct multiple of 4 for t = 0..15

__device__ __forceinline__ uint64_t lower_bound_u64(const uint64_t* __restrict__ a, uint64_t n, uint64_t x) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// one wave per sketch: set bit rank(h) for every hash h of the row
__global__ __launch_bounds__(256) void bitmap_build_kernel(const uint64_t* __restrict__ hashes,
                                                           const uint64_t* __restrict__ offsets, uint32_t n,
                                                           const uint64_t* __restrict__ dict, uint64_t U,
                                                           uint32_t* __restrict__ bits, uint32_t words_per_row) {
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    for (uint64_t row = wave; row < n; row += n_waves) {
        const uint64_t lo = offsets[row], hi = offsets[row + 1];
        uint32_t* out = bits + row * (uint64_t)words_per_row;
        for (uint64_t i = lo + lane; i < hi; i += 64) {
            const uint64_t rank = lower_bound_u64(dict, U, hashes[i]);
            atomicOr(&out[rank >> 5], 1u << (rank & 31));
        }
    }
}

// popcount(x) + acc in one instruction
__device__ __forceinline__ uint32_t popc_acc(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

// common[local row][col] = popcount(bits[row] & bits[col]) for a 64 x 64 tile; lane (tr, tc) owns the
// 4 x 4 pairs (tr + 16 i, tc + 16 j).  Rows of the launch are the 16-row tiles rb_first,
// rb_first + rb_stride, ... (same dealing as compare.hip); four of them form one 64-row group.
__global__ __launch_bounds__(256) void bitmatrix_kernel(const uint32_t* __restrict__ bits, uint32_t words_per_row,
                                                        uint32_t n, uint32_t rb_first, uint32_t rb_stride,
                                                        uint32_t rb_count, uint32_t* __restrict__ common, uint32_t upper_only) {
    // upper_only: the caller mirrors the triangle afterwards (symmetrize_kernel), so a tile whose columns all lie left of
    // the group's first row is never read: half of the launch at world size 1
    if (upper_only && (blockIdx.x + 1) * BT <= (rb_first + blockIdx.y * 4 * rb_stride) * 16) return;
    __shared__ __attribute__((aligned(16))) uint32_t sA[BT * BSTRIDE];
    __shared__ __attribute__((aligned(16))) uint32_t sB[BT * BSTRIDE];
    const int tid = threadIdx.x;
    const int tr = tid >> 4, tc = tid & 15;
    const uint32_t col0 = blockIdx.x * BT;
    const uint32_t grp = blockIdx.y;                                 // 64-row group = 4 owned 16-row tiles

    auto global_row = [&](uint32_t local) -> uint32_t {              // local row in the group -> global row
        const uint32_t t = grp * 4 + (local >> 4);
        if (t >= rb_count) return 0xffffffffu;
        return (rb_first + t * rb_stride) * 16 + (local & 15);
    };

    uint32_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0;

    // staging assignment: 64 rows x 8 chunks of 4 words = 512 chunks per operand, 2 per thread.  The words of step k0 + BKC
    // are fetched into registers while step k0 is being computed from LDS, so the trip to L2 / HBM overlaps the popcounts
    // instead of sitting between two barriers.
    const uint32_t* srcA[2];
    const uint32_t* srcB[2];
    uint32_t dstS[2];
    uint32_t mA[2], mB[2];                                           // all ones, or zero for a row past the end
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int chunk = tid + q * 256;
        const int row = chunk >> 3, kc = (chunk & 7) * 4;
        const uint32_t gr = global_row((uint32_t)row);
        const uint32_t gc = col0 + (uint32_t)row;
        mA[q] = gr < n ? 0xffffffffu : 0u;                              // rows past the end read row 0 and are masked out
        mB[q] = gc < n ? 0xffffffffu : 0u;
        srcA[q] = bits + (uint64_t)(gr < n ? gr : 0u) * words_per_row + kc;
        srcB[q] = bits + (uint64_t)(gc < n ? gc : 0u) * words_per_row + kc;
        dstS[q] = (uint32_t)(row * BSTRIDE + kc);
    }
    uint4 ra[2], rb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        ra[q] = *reinterpret_cast<const uint4*>(srcA[q]);
        rb[q] = *reinterpret_cast<const uint4*>(srcB[q]);
    }
    for (uint32_t k0 = 0; k0 < words_per_row; k0 += BKC) {
        __syncthreads();                                                // the previous step's readers are done
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            *reinterpret_cast<uint4*>(&sA[dstS[q]]) = make_uint4(ra[q].x & mA[q], ra[q].y & mA[q], ra[q].z & mA[q], ra[q].w & mA[q]);
            *reinterpret_cast<uint4*>(&sB[dstS[q]]) = make_uint4(rb[q].x & mB[q], rb[q].y & mB[q], rb[q].z & mB[q], rb[q].w & mB[q]);
        }
        __syncthreads();
        if (k0 + BKC < words_per_row) {                                 // next step's words: in flight during the compute below
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                ra[q] = *reinterpret_cast<const uint4*>(srcA[q] + k0 + BKC);
                rb[q] = *reinterpret_cast<const uint4*>(srcB[q] + k0 + BKC);
            }
        }
#pragma unroll
        for (int k4 = 0; k4 < BKC; k4 += 4) {
            uint4 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const uint4*>(&sA[(tr + 16 * i) * BSTRIDE + k4]);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const uint4*>(&sB[(tc + 16 * j) * BSTRIDE + k4]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // one v_and + one v_bcnt per word, the v_bcnt taking the running sum as its second operand.  Spelled as
                    // an instruction: from `acc += popc(..) + popc(..) + ..` hipcc makes v_bcnt x, 0 plus a v_add3 per two
                    // words -- 2.55 instructions per word instead of 2.05, and v_add3 is a half-rate instruction like v_bcnt
                    // (profiles/r02_compare_pmc.txt, r01_ubench_valu.txt)
                    uint32_t s = acc[i][j];
                    s = popc_acc(a[i].x & b[j].x, s);
                    s = popc_acc(a[i].y & b[j].y, s);
                    s = popc_acc(a[i].z & b[j].z, s);
                    s = popc_acc(a[i].w & b[j].w, s);
                    acc[i][j] = s;
                }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t local = (uint32_t)(tr + 16 * i);
        const uint32_t gr = global_row(local);
        if (gr >= n) continue;
        const uint64_t out_row = (uint64_t)(grp * 64 + local) * n;     // owned tiles back to back
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t gc = col0 + (uint32_t)(tc + 16 * j);
            if (gc < n) common[out_row + gc] = acc[i][j];
        }
    }
}

hipError_t bitmap_build_launch(const uint64_t* d_hashes, const uint64_t* d_offsets, uint32_t n, const uint64_t* d_dict,
                               uint64_t U, uint32_t* d_bits, uint32_t words_per_row, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(d_bits, 0, (size_t)n * words_per_row * 4, stream);
    if (e != hipSuccess) return e;
    const uint64_t blocks = ((uint64_t)n + 3) / 4;
    hipLaunchKernelGGL(bitmap_build_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, stream,
                       d_hashes, d_offsets, n, d_dict, U, d_bits, words_per_row);
    return hipGetLastError();
}

hipError_t bitmatrix_launch(const uint32_t* d_bits, uint32_t words_per_row, uint32_t n, uint32_t rb_first,
                            uint32_t rb_stride, uint32_t rb_count, uint32_t* d_common, hipStream_t stream, bool upper_only) {
    if (n == 0 || rb_count == 0) return hipSuccess;
    dim3 grid((n + BT - 1) / BT, (rb_count + 3) / 4);
    hipLaunchKernelGGL(bitmatrix_kernel, grid, dim3(256), 0, stream, d_bits, words_per_row, n, rb_first, rb_stride,
                       rb_count, d_common, upper_only ? 1u : 0u);
    return hipGetLastError();
}

}  // namespace smg
