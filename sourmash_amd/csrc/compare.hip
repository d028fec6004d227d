// compare.hip -- all-vs-all sorted-u64 merge-intersection, tiled in LDS (gfx950).
//
// GPU counterpart of
//   src/sourmash/compare.py:14-64      compare_serial (the N(N-1)/2 Python loop)
//   src/core/src/sketch/minhash.rs:539-558  count_common
//   src/core/src/sketch/minhash.rs:915-953  Intersection (two-pointer walk)
//   src/core/src/sketch/minhash.rs:1765-1807 intersection_size (common, union)
//   src/core/src/sketch/minhash.rs:624-631  jaccard = common / max(1, union)
//
// Layout: CSR -- d_hashes holds every sketch's sorted unique u64 hashes back to
// back, d_offsets[n+1] the row starts.
//
// Kernel: one workgroup owns a 16 x 16 tile of (row sketch, column sketch)
// pairs, one lane per pair.  The 32 sketches of the tile are streamed through
// LDS in lock-step "slabs": every round each sketch contributes its next
// <= SEG (64) hashes (coalesced loads, one wave per sketch segment), the
// slab's upper bound `hi` is the smallest last-loaded hash among sketches that
// still have more to come, and every lane two-pointer-merges the parts of its
// row and column segments that are <= hi.  Sketches then advance by exactly
// what was consumed.  This is the reference's merge walk, cut at common hash
// boundaries so that 256 walks share each byte fetched: HBM/L2 sees every
// hash of a tile once per round instead of 16 times.  Works for any length
// mix (empty, 1-hash, 50k-hash rows) because the cut points are data driven.
//
// union = n_i + n_j - common (scaled sketches; equals the walk's union count),
// so Jaccard needs only the u32 common matrix and the row lengths.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "device_api.hpp"

namespace smg {

constexpr int CT = 16;            // tile edge (sketches)
constexpr int SEG_PAD = 2;        // u64 slack per staged segment: the walk may look one block past the end, and the
                                  // odd dword shift spreads the 16 column segments over distinct LDS banks
constexpr int CMP_BLOCK = CT * CT;
constexpr int CMP_ZMAX = 16;      // hash-range slices per tile (grid.z); a tile uses ceil(longest / slice_len) of them

__device__ __forceinline__ uint64_t lower_bound_row(const uint64_t* __restrict__ a, uint64_t lo, uint64_t hi, uint64_t x) {
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// One unit of work: slice z (of zt) of the tile (row tile index `ty` in launch order, column tile cb).
struct WorkItem { uint32_t ty, cb, z, zt; };

// Planning pass: one lane per tile decides how many hash-range slices the tile is cut into
// (ceil(longest sketch / slice_len), at most CMP_ZMAX) and appends its work items.  Tiles that get
// sliced (they contain an unusually long sketch) go to the `heavy` list, which the workers drain
// first: longest-processing-time-first scheduling keeps ragged collections from leaving a tail.
__global__ __launch_bounds__(256) void compare_plan_kernel(
    const uint64_t* __restrict__ offsets, uint32_t n, uint32_t row_lo, uint32_t row_hi, int symmetric,
    uint32_t rb_first, uint32_t rb_stride, uint32_t n_row_tiles, uint32_t n_col_tiles, uint32_t slice_len,
    WorkItem* __restrict__ heavy, WorkItem* __restrict__ light, unsigned int* __restrict__ counters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_row_tiles * n_col_tiles) return;
    const uint32_t ty = t / n_col_tiles, cb = t % n_col_tiles;
    const uint32_t rb = rb_first + ty * rb_stride;
    const uint32_t row0 = row_lo + rb * CT, col0 = cb * CT;
    if (symmetric && col0 + CT <= row0) return;              // strictly below the diagonal
    uint64_t best = 0;
    for (int i = 0; i < CT; ++i) {
        const uint32_t r = row0 + i, c = col0 + i;
        if (r < row_hi) { const uint64_t l = offsets[r + 1] - offsets[r]; best = l > best ? l : best; }
        if (c < n) { const uint64_t l = offsets[c + 1] - offsets[c]; best = l > best ? l : best; }
    }
    if (best == 0) return;                                    // nothing can intersect
    uint32_t zt = (uint32_t)((best + slice_len - 1) / slice_len);
    zt = zt < 1 ? 1 : (zt > (uint32_t)CMP_ZMAX ? (uint32_t)CMP_ZMAX : zt);
    WorkItem* list = zt > 1 ? heavy : light;
    const unsigned int base = atomicAdd(&counters[zt > 1 ? 0 : 1], zt);
    for (uint32_t z = 0; z < zt; ++z) list[base + z] = WorkItem{ty, cb, z, zt};
}

// Worker: persistent workgroups pull work items (heavy list first) with one atomic per item.
template <int SEG>
__global__ __launch_bounds__(CMP_BLOCK) void compare_tile_kernel(
    const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ offsets, uint32_t n,
    uint32_t row_lo, uint32_t row_hi, uint32_t* __restrict__ common, int symmetric,
    uint32_t rb_first, uint32_t rb_stride, const WorkItem* __restrict__ heavy,
    const WorkItem* __restrict__ light, unsigned int* __restrict__ counters) {
    // symmetric: 0 = every tile; 1 = all rows local: tiles on/above the diagonal, mirrored on write;
    // 2 = upper tiles only, no mirror (sharded launch; symmetrize_kernel fills the rest).
    // counters: [0] heavy items, [1] light items, [2] next heavy, [3] next light.
    // Output rows: the tiles this launch owns back to back (local row = ty * CT + r); pre-zeroed,
    // partial counts of the slices of a tile are combined with atomicAdd.
    constexpr int SEG_STRIDE = SEG + SEG_PAD;
    constexpr int PER_LANE = SEG / 64;                 // staged hashes per lane and sketch
    static_assert(SEG % 64 == 0, "a wave stages whole rounds of 64 hashes");
    __shared__ uint64_t s_seg[2 * CT][SEG_STRIDE];
    __shared__ uint64_t s_pos[2 * CT], s_end[2 * CT];
    __shared__ uint32_t s_take[2 * CT];
    __shared__ unsigned long long s_hi;
    __shared__ uint32_t s_live[2];   // [0] rows with data left, [1] columns with data left
    __shared__ uint64_t s_piv[2];
    __shared__ WorkItem s_item;
    __shared__ int s_have;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = tid / CT, c = tid % CT;

    for (;;) {
        // ---- fetch the next work item ----
        __syncthreads();                                   // previous item fully done (LDS reuse)
        if (tid == 0) {
            int have = 0;
            unsigned int i = atomicAdd(&counters[2], 1u);
            if (i < counters[0]) { s_item = heavy[i]; have = 1; }
            else {
                i = atomicAdd(&counters[3], 1u);
                if (i < counters[1]) { s_item = light[i]; have = 1; }
            }
            s_have = have;
        }
        __syncthreads();
        if (!s_have) return;
        const WorkItem it = s_item;
        const uint32_t rb = rb_first + it.ty * rb_stride;
        const uint32_t row0 = row_lo + rb * CT, col0 = it.cb * CT;

        if (tid < 2 * CT) {
            const uint32_t s = tid < CT ? row0 + tid : col0 + (tid - CT);
            const bool ok = tid < CT ? (s < row_hi) : (s < n);
            s_pos[tid] = ok ? offsets[s] : 0;
            s_end[tid] = ok ? offsets[s + 1] : 0;
        }
        __syncthreads();
        // ---- hash-range slice: the tile's longest sketch is cut into zt equal pieces (pivot values);
        //      slice z of EVERY sketch is its part inside [pivot_z, pivot_z+1) ----
        if (it.zt > 1) {
            if (tid == 0) {
                uint64_t best_len = 0, best_pos = 0;
                for (int i = 0; i < 2 * CT; ++i) {
                    const uint64_t len = s_end[i] - s_pos[i];
                    if (len > best_len) { best_len = len; best_pos = s_pos[i]; }
                }
                s_piv[0] = it.z == 0 ? 0ull : hashes[best_pos + (uint64_t)it.z * best_len / it.zt];
                s_piv[1] = it.z + 1 == it.zt ? ~0ull : hashes[best_pos + (uint64_t)(it.z + 1) * best_len / it.zt];
            }
            __syncthreads();
            if (tid < 2 * CT) {
                const uint64_t lo = s_pos[tid], hi = s_end[tid];
                const uint64_t a = it.z == 0 ? lo : lower_bound_row(hashes, lo, hi, s_piv[0]);
                const uint64_t b = it.z + 1 == it.zt ? hi : lower_bound_row(hashes, a, hi, s_piv[1]);
                s_pos[tid] = a;
                s_end[tid] = b;
            }
            __syncthreads();
        }
        uint32_t cnt = 0;

        for (;;) {
            if (tid == 0) { s_hi = ~0ull; s_live[0] = 0; s_live[1] = 0; }
            __syncthreads();
            // ---- stage the next <= SEG hashes of each of the 32 sketches: wave w takes sketches 8w..8w+7,
            //      lane l the hashes 2l, 2l+1 (one 16-byte load when aligned) ----
            uint64_t e[8][PER_LANE];
            uint32_t have[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int s = wave * 8 + i;
                const uint64_t pos = s_pos[s], end = s_end[s];
                const uint64_t left = end - pos;
                const uint32_t len = left < (uint64_t)SEG ? (uint32_t)left : (uint32_t)SEG;
                have[i] = len;
#pragma unroll
                for (int j = 0; j < PER_LANE; ++j) {
                    const uint32_t idx = (uint32_t)(lane + 64 * j);
                    const uint64_t v = idx < len ? hashes[pos + idx] : ~0ull;
                    e[i][j] = v;
                    s_seg[s][idx] = v;
                }
                if (lane == 0) {
                    if (left > (uint64_t)SEG) atomicMin(&s_hi, (unsigned long long)hashes[pos + SEG - 1]);
                    if (len) atomicOr(&s_live[s < CT ? 0 : 1], 1u);
                }
            }
            __syncthreads();
            if (s_live[0] == 0 || s_live[1] == 0) break;   // every row or every column exhausted
            const uint64_t hi = s_hi;
            // ---- how many staged hashes of each sketch are <= hi ----
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int s = wave * 8 + i;
                uint32_t take = 0;
#pragma unroll
                for (int j = 0; j < PER_LANE; ++j)
                    take += (uint32_t)__popcll(__ballot((uint32_t)(lane + 64 * j) < have[i] && e[i][j] <= hi));
                if (lane == 0) { s_take[s] = take; s_pos[s] += take; }
            }
            __syncthreads();
            // ---- the merge walk of minhash.rs:915-953 on the LDS-resident parts ----
            {
                const uint32_t na = s_take[r], nb = s_take[CT + c];
                const uint64_t* A = s_seg[r];
                const uint64_t* B = s_seg[CT + c];
                uint32_t ia = 0, ib = 0;
                while (ia < na && ib < nb) {
                    const uint64_t a = A[ia], b = B[ib];
                    const bool lt = a < b, gt = b < a;         // compiles to three compares feeding three add-with-carry
                    cnt += !(lt | gt);
                    ia += !gt;
                    ib += !lt;
                }
            }
            // next round's staging overwrites s_seg: the barrier at the top of the loop orders it
        }

        const uint32_t row = row0 + r, col = col0 + c;
        if (row < row_hi && col < n && cnt) {
            const uint64_t idx = (uint64_t)(it.ty * CT + r) * n + col;
            if (!symmetric) {
                atomicAdd(&common[idx], cnt);
            } else if (col >= row) {
                atomicAdd(&common[idx], cnt);
                if (symmetric == 1 && col != row) atomicAdd(&common[(uint64_t)(col - row_lo) * n + row], cnt);
            }
        }
    }
}

__global__ __launch_bounds__(256) void jaccard_from_counts_kernel(const uint32_t* __restrict__ common,
                                                                  const uint64_t* __restrict__ offsets, uint32_t n,
                                                                  uint32_t row_lo, uint32_t row_hi,
                                                                  double* __restrict__ out) {
    const uint64_t total = (uint64_t)(row_hi - row_lo) * n;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t row = row_lo + (uint32_t)(i / n), col = (uint32_t)(i % n);
        double v;
        if (row == col) {
            v = 1.0;                                              // compare.py:33 np.ones diagonal
        } else {
            const uint64_t ni = offsets[row + 1] - offsets[row], nj = offsets[col + 1] - offsets[col];
            const uint64_t cm = common[i];
            const uint64_t uni = ni + nj - cm;
            v = (double)cm / (double)(uni > 1 ? uni : 1);         // minhash.rs:624-631
        }
        out[i] = v;
    }
}

// Slice length: long enough that ordinary sketches (a few thousand hashes) are never cut when there
// are plenty of tiles, shorter when the tile count alone cannot fill 256 CUs x 4 workgroups.
static uint32_t pick_slice_len(uint64_t n_tiles) {
    if (n_tiles >= 2048) return 8192;
    if (n_tiles >= 512) return 2048;
    return 1024;
}

size_t compare_workspace_bytes(uint32_t n_row_tiles, uint32_t n_col_tiles) {
    const size_t tiles = (size_t)n_row_tiles * n_col_tiles;
    return 256 + tiles * sizeof(WorkItem) * (CMP_ZMAX + 1);        // heavy (x ZMAX) + light (x 1) lists
}

static hipError_t compare_launch(const uint64_t* d_hashes, const uint64_t* d_offsets, uint32_t n, uint32_t row_lo,
                                 uint32_t row_hi, int symmetric, uint32_t rb_first, uint32_t rb_stride,
                                 uint32_t n_row_tiles, uint32_t* d_common, hipStream_t stream) {
    const uint32_t n_col_tiles = (n + CT - 1) / CT;
    const size_t tiles = (size_t)n_row_tiles * n_col_tiles;
    hipError_t e = hipMemsetAsync(d_common, 0, (size_t)n_row_tiles * CT * n * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    void* ws = nullptr;
    e = hipMallocAsync(&ws, compare_workspace_bytes(n_row_tiles, n_col_tiles), stream);   // stream-ordered scratch
    if (e != hipSuccess) return e;
    unsigned int* counters = (unsigned int*)ws;
    WorkItem* heavy = (WorkItem*)((char*)ws + 256);
    WorkItem* light = heavy + tiles * CMP_ZMAX;
    e = hipMemsetAsync(counters, 0, 256, stream);
    if (e == hipSuccess) {
        const uint64_t work_tiles = tiles / (symmetric ? 2 : 1);
        hipLaunchKernelGGL(compare_plan_kernel, dim3((unsigned)((tiles + 255) / 256)), dim3(256), 0, stream, d_offsets, n,
                           row_lo, row_hi, symmetric, rb_first, rb_stride, n_row_tiles, n_col_tiles,
                           pick_slice_len(work_tiles), heavy, light, counters);
        // 64 hashes per sketch and round: 17 KiB of LDS per workgroup, 8 workgroups (= 8 waves per SIMD) per CU.
        // The walk is a dependent LDS round trip per step; twice the resident waves hide it better than longer
        // rounds amortise the barriers (measured: +33 % pairs/s over 128 hashes per round at 5,000-hash sketches).
        const uint64_t cap = 256ull * 8;
        const unsigned grid0 = (unsigned)(work_tiles + 1 < cap ? work_tiles + 1 : cap);
        hipLaunchKernelGGL((compare_tile_kernel<64>), dim3(grid0 < 1 ? 1 : grid0), dim3(CMP_BLOCK), 0, stream, d_hashes,
                           d_offsets, n, row_lo, row_hi, d_common, symmetric, rb_first, rb_stride, heavy, light, counters);
        e = hipGetLastError();
    }
    const hipError_t e2 = hipFreeAsync(ws, stream);
    return e != hipSuccess ? e : e2;
}

hipError_t compare_counts_launch(const uint64_t* d_hashes, const uint64_t* d_offsets, uint32_t n, uint32_t row_lo,
                                 uint32_t row_hi, uint32_t* d_common, hipStream_t stream) {
    if (row_hi <= row_lo || n == 0) return hipSuccess;
    const int symmetric = (row_lo == 0 && row_hi == n) ? 1 : 0;
    return compare_launch(d_hashes, d_offsets, n, row_lo, row_hi, symmetric, 0u, 1u, (row_hi - row_lo + CT - 1) / CT,
                          d_common, stream);
}

// Sharded form: this launch owns the row tiles rb_first, rb_first + rb_stride, ... (rb_count of them)
// of the full n x n problem and computes only their tiles on or above the diagonal.
hipError_t compare_blocks_launch(const uint64_t* d_hashes, const uint64_t* d_offsets, uint32_t n, uint32_t rb_first,
                                 uint32_t rb_stride, uint32_t rb_count, uint32_t* d_common, hipStream_t stream) {
    if (rb_count == 0 || n == 0) return hipSuccess;
    return compare_launch(d_hashes, d_offsets, n, 0u, n, 2, rb_first, rb_stride, rb_count, d_common, stream);
}

__global__ __launch_bounds__(256) void symmetrize_kernel(uint32_t* __restrict__ m, uint32_t n) {
    // m[j][i] = m[i][j] for i < j.  32 x 32 tiles through LDS so both the read and the write are coalesced.
    __shared__ uint32_t t[32][33];
    const uint32_t bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return;
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (uint32_t r = ty; r < 32; r += 8) {
        const uint32_t i = bi * 32 + r, j = bj * 32 + tx;
        t[r][tx] = (i < n && j < n) ? m[(uint64_t)i * n + j] : 0;
    }
    __syncthreads();
    for (uint32_t r = ty; r < 32; r += 8) {
        const uint32_t j = bj * 32 + r, i = bi * 32 + tx;        // writing element (j, i), j is the row
        if (i < n && j < n && i < j) m[(uint64_t)j * n + i] = t[tx][r];
    }
}

hipError_t symmetrize_launch(uint32_t* d_common, uint32_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    dim3 grid((n + 31) / 32, (n + 31) / 32);
    hipLaunchKernelGGL(symmetrize_kernel, grid, dim3(256), 0, stream, d_common, n);
    return hipGetLastError();
}

hipError_t jaccard_from_counts_launch(const uint32_t* d_common, const uint64_t* d_offsets, uint32_t n,
                                      uint32_t row_lo, uint32_t row_hi, double* d_out, hipStream_t stream) {
    if (row_hi <= row_lo || n == 0) return hipSuccess;
    const uint64_t total = (uint64_t)(row_hi - row_lo) * n;
    const uint64_t nb = (total + 255) / 256;
    hipLaunchKernelGGL(jaccard_from_counts_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, stream,
                       d_common, d_offsets, n, row_lo, row_hi, d_out);
    return hipGetLastError();
}

}  // namespace smg
