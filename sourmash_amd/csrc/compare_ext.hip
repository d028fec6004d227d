// compare_ext.hip -- all-pairs tiles for the two similarities the flat scaled kernels (compare.hip, bitindex.hip) do not give:
//
//   * bottom-k ("num") sketches: |A ∩ B ∩ bottom_num(A ∪ B)| and |bottom_num(A ∪ B)|
//       src/core/src/sketch/minhash.rs:593-621  intersection_size for num sketches: merge both into a sketch truncated to
//                                               `self.num`, intersect (A ∩ B) with that; union size = its length
//       src/core/src/sketch/minhash.rs:624-631  jaccard = common / max(1, size)
//   * abundance-tracking sketches: sum over the common hashes of abund_A * abund_B (u64), per row the sum of squares
//       src/core/src/sketch/minhash.rs:635-680  angular_similarity (the integer sums here; sqrt / acos on the host's libm)
//
// for every pair of a collection in one launch (src/sourmash/compare.py:14-64 walks the pairs in Python and calls
// similarity() per pair; round 3 of this library did the same with one 12 us GPU call per pair).
//
// Formulation.  A workgroup owns a 16 x 16 tile of (row sketch, column sketch) pairs, one lane per pair, and streams the 32
// sketches through LDS in lock-step rounds exactly like compare.hip's walk kernel: every round each sketch stages its next
// <= 64 hashes, the round's bound `hi` is the smallest last-staged hash among sketches with more to come, and every lane
// runs the reference's two-pointer walk (minhash.rs:915-953) over its row's and its column's staged hashes <= hi.
//   num:   one step of that walk consumes exactly ONE element of A ∪ B (the smaller head, or both heads when they are equal), in
//          ascending order -- so "the intersection with the merged sketch truncated to num" is simply "the equal heads met
//          during the first num steps".  A lane counts its steps across rounds and stops counting at num; elements a round
//          leaves over on one side (the other side's part <= hi is used up) are union elements too and are added to the step
//          count.  |merged| = min(num, |A| + |B| - common): when the walk was cut, |A ∪ B| >= num and the formula gives num
//          whatever was counted; when it was not, common is the full intersection.
//   abund: at equal heads the lane multiplies the two abundances (u64, wrapping like the reference's release build) -- as one
//          v_mad_u64_u32 when every abundance of the collection fits 32 bits (the caller checks), else the full 64 x 64 product.
// Only tiles on or above the diagonal run; a lane writes its pair's entry and the mirrored one.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "device_api.hpp"

namespace smg {

namespace {

constexpr int XT = 16;                 // tile edge (sketches)
constexpr int XSEG = 64;               // hashes staged per sketch and round
constexpr int XPAD = 2;                // u64 slack per staged segment (the 16 column streams of a half-wave hit distinct banks)
constexpr int XSTRIDE = XSEG + XPAD;

constexpr int XZMAX = 16;              // hash-range slices a tile of abundance sketches can be cut into (grid.z)
constexpr uint32_t XSLICE = 6144;      // hashes of the tile's longest sketch per slice: ordinary sketches (~5,000) are never cut
constexpr uint32_t XSLICE_SMALL = 1700;   // ... for small grids (see compare_abund_launch)

enum { X_NUM = 0, X_ABUND32 = 1, X_ABUND64 = 2 };

__device__ __forceinline__ uint64_t lower_bound_u64(const uint64_t* __restrict__ a, uint64_t lo, uint64_t hi, uint64_t x) {
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

template <int MODE>
__global__ __launch_bounds__(XT * XT) void compare_ext_kernel(
    const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ abunds, const uint64_t* __restrict__ offsets,
    const uint32_t* __restrict__ nums, uint32_t n, uint32_t* __restrict__ common, unsigned long long* __restrict__ prod,
    uint32_t slice_len) {
    // nums: per sketch, the `num` of the pair (i, j), i < j, is nums[i] (compare.py:39: siglist[i].similarity(siglist[j]);
    //       minhash.rs:596-604: self.num); common / prod: full n x n, diagonal left to the caller
    constexpr bool ABUND = MODE != X_NUM;
    using AbT = typename std::conditional<MODE == X_ABUND32, uint32_t, uint64_t>::type;
    // 64-bit abundances: a staged element is {hash, abundance} side by side, 16 bytes -- the walk's step reads each side's head with
    // ONE ds_read_b128 (measured at C3: 7.83 -> 6.86 ms).  32-bit abundances keep two arrays: the 16-byte elements cost two of the six
    // workgroups a CU's LDS holds, and with this latency-bound walk occupancy is worth more (5.86 ms against 6.49 interleaved).
    constexpr bool PACKED = MODE == X_ABUND64;
    struct alignas(16) Elem { uint64_t h; uint64_t a; };
    __shared__ uint64_t s_seg[PACKED ? 1 : 2 * XT][PACKED ? 1 : XSTRIDE];
    __shared__ uint32_t s_ab[MODE == X_ABUND32 ? 2 * XT : 1][MODE == X_ABUND32 ? XSTRIDE : 1];
    __shared__ Elem s_el[PACKED ? 2 * XT : 1][PACKED ? XSEG + 1 : 1];
    __shared__ uint64_t s_pos[2 * XT], s_end[2 * XT];
    __shared__ uint32_t s_take[2 * XT];
    __shared__ unsigned long long s_hi;
    __shared__ uint32_t s_live[2];
    __shared__ uint64_t s_piv[2];
    __shared__ uint32_t s_zt;

    const uint32_t rt = blockIdx.y, ct = blockIdx.x;
    if (ct < rt) return;                                       // below the diagonal: the mirror of a tile that runs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = tid / XT, c = tid % XT;
    const uint32_t row0 = rt * XT, col0 = ct * XT;
    const uint32_t row = row0 + r, col = col0 + c;

    if (tid < 2 * XT) {
        const uint32_t s = tid < XT ? row0 + tid : col0 + (tid - XT);
        const bool ok = s < n;
        s_pos[tid] = ok ? offsets[s] : 0;
        s_end[tid] = ok ? offsets[s + 1] : 0;
    }
    __syncthreads();
    // ---- abundance tiles holding an unusually long sketch are cut into hash-range slices (grid.z), like compare.hip's walk:
    //      one 50,000-hash row among 5,000-hash rows would otherwise keep its 16-sketch tile running ten times as long as
    //      the others and the whole launch waits for it.  Slice z of EVERY sketch of the tile is its part between two
    //      pivot values taken from the tile's longest sketch; sums over disjoint hash ranges add up (atomics below).
    //      Bottom-k walks count steps from the smallest hash on and cannot be cut -- they are short by construction. ----
    if (ABUND) {
        if (wave == 0) {                                       // (one wave decides: most workgroups with z > 0 leave right here)
            const uint64_t pos = lane < 2 * XT ? s_pos[lane] : 0, len = lane < 2 * XT ? s_end[lane] - pos : 0;
            uint64_t best_len = len;
#pragma unroll
            for (int o = 32; o; o >>= 1) { const uint64_t x = __shfl_xor(best_len, o); best_len = x > best_len ? x : best_len; }
            const unsigned long long holders = __ballot(len == best_len);
            const uint64_t best_pos = __shfl(pos, (int)__builtin_ctzll(holders));
            if (lane == 0) {
                uint32_t zt = (uint32_t)((best_len + slice_len - 1) / slice_len);
                zt = zt < 1 ? 1 : (zt > (uint32_t)XZMAX ? (uint32_t)XZMAX : zt);
                s_zt = zt;
                const uint32_t z = blockIdx.z;
                if (z < zt && zt > 1) {
                    s_piv[0] = z == 0 ? 0ull : hashes[best_pos + (uint64_t)z * best_len / zt];
                    s_piv[1] = z + 1 == zt ? ~0ull : hashes[best_pos + (uint64_t)(z + 1) * best_len / zt];
                }
            }
        }
        __syncthreads();
        const uint32_t zt = s_zt, z = blockIdx.z;
        if (z >= zt) return;
        if (zt > 1) {
            if (tid < 2 * XT) {
                const uint64_t lo = s_pos[tid], hi = s_end[tid];
                const uint64_t a = z == 0 ? lo : lower_bound_u64(hashes, lo, hi, s_piv[0]);
                const uint64_t b = z + 1 == zt ? hi : lower_bound_u64(hashes, a, hi, s_piv[1]);
                s_pos[tid] = a;
                s_end[tid] = b;
            }
            __syncthreads();
        }
    } else if (blockIdx.z) return;
    // a lane whose pair does not exist (past n, or below the diagonal of a diagonal tile) walks nothing
    const bool mine = row < n && col < n && col > row;
    const uint32_t cap = MODE == X_NUM && mine ? nums[row] : 0xffffffffu;
    uint32_t cnt = 0, steps = 0;
    unsigned long long acc = 0;

    for (;;) {
        if (tid == 0) { s_hi = ~0ull; s_live[0] = 0; s_live[1] = 0; }
        __syncthreads();
        // ---- stage the next <= 64 hashes (and abundances) of each of the 32 sketches: wave w takes sketches 8w .. 8w + 7 ----
        uint64_t e[8];
        uint32_t have[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int s = wave * 8 + i;
            const uint64_t pos = s_pos[s], end = s_end[s];
            const uint64_t left = end - pos;
            const uint32_t len = left < (uint64_t)XSEG ? (uint32_t)left : (uint32_t)XSEG;
            have[i] = len;
            const uint64_t v = (uint32_t)lane < len ? hashes[pos + lane] : ~0ull;
            e[i] = v;
            if (PACKED) s_el[s][lane] = Elem{v, (uint32_t)lane < len ? abunds[pos + lane] : 0ull};
            else s_seg[s][lane] = v;
            if (MODE == X_ABUND32) s_ab[s][lane] = (uint32_t)lane < len ? (uint32_t)abunds[pos + lane] : 0u;
            if (lane == 0) {
                if (left > (uint64_t)XSEG) atomicMin(&s_hi, (unsigned long long)hashes[pos + XSEG - 1]);
                if (len) atomicOr(&s_live[s < XT ? 0 : 1], 1u);
            }
        }
        __syncthreads();
        // every row or every column used up: no pair of the tile can meet another common hash, and in num mode further union
        // elements cannot change a count either
        if (s_live[0] == 0 || s_live[1] == 0) break;
        const uint64_t hi = s_hi;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int s = wave * 8 + i;
            const uint32_t take = (uint32_t)__popcll(__ballot((uint32_t)lane < have[i] && e[i] <= hi));
            if (lane == 0) { s_take[s] = take; s_pos[s] += take; }
        }
        __syncthreads();
        if (mine) {
            const uint32_t na = s_take[r], nb = s_take[XT + c];
            const uint64_t* A = s_seg[PACKED ? 0 : r];
            const uint64_t* B = s_seg[PACKED ? 0 : XT + c];
            uint32_t ia = 0, ib = 0;
            if (MODE == X_NUM) {
                while (ia < na && ib < nb && steps < cap) {
                    const uint64_t a = A[ia], b = B[ib];
                    const bool lt = a < b, gt = b < a;
                    cnt += !(lt | gt);
                    ia += !gt;
                    ib += !lt;
                    ++steps;
                }
                steps += (na - ia) + (nb - ib);                // what one side has left below the bound: union elements, none common
            } else {
                const Elem* EA = s_el[PACKED ? r : 0];
                const Elem* EB = s_el[PACKED ? XT + c : 0];
                const uint32_t* WA = s_ab[PACKED ? 0 : r];
                const uint32_t* WB = s_ab[PACKED ? 0 : XT + c];
                while (ia < na && ib < nb) {
                    uint64_t a, b;
                    AbT wa, wb;
                    if (PACKED) {
                        const Elem ea = EA[ia], eb = EB[ib];
                        a = ea.h; b = eb.h; wa = (AbT)ea.a; wb = (AbT)eb.a;
                    } else {
                        a = A[ia]; b = B[ib]; wa = (AbT)WA[ia]; wb = (AbT)WB[ib];
                    }
                    const bool lt = a < b, gt = b < a, eq = !(lt | gt);
                    // no branch: with 64 walks per wave some lane meets a common hash at almost every step anyway
                    cnt += eq;
                    acc += (unsigned long long)wa * (unsigned long long)(eq ? wb : (AbT)0);   // AbT = u32: select + one v_mad_u64_u32
                    ia += !gt;
                    ib += !lt;
                }
            }
        }
        // the next round's staging overwrites the segments: the barrier at the top of the loop orders it
    }
    if (mine) {
        if (!ABUND) {
            common[(uint64_t)row * n + col] = cnt;
            common[(uint64_t)col * n + row] = cnt;
        } else if (s_zt == 1) {
            common[(uint64_t)row * n + col] = cnt;
            common[(uint64_t)col * n + row] = cnt;
            prod[(uint64_t)row * n + col] = acc;
            prod[(uint64_t)col * n + row] = acc;
        } else if (cnt) {                                      // slices of one tile meet in the (zeroed) matrices
            atomicAdd(&common[(uint64_t)row * n + col], cnt);
            atomicAdd(&common[(uint64_t)col * n + row], cnt);
            atomicAdd(&prod[(uint64_t)row * n + col], acc);
            atomicAdd(&prod[(uint64_t)col * n + row], acc);
        }
    }
}

// per sketch: sum of squared abundances (u64, wrapping) and the diagonal of the matrices; a workgroup per sketch, four loads per
// thread in flight (a wave per sketch with one load at a time took 0.2 ms for 1,000 sketches of 5,000: 79 trips to memory in a row)
__global__ __launch_bounds__(256) void ext_rows_kernel(const uint64_t* __restrict__ abunds, const uint64_t* __restrict__ offsets,
                                                       uint32_t n, uint32_t* __restrict__ common, unsigned long long* __restrict__ prod,
                                                       unsigned long long* __restrict__ sumsq) {
    __shared__ unsigned long long s_part[4];
    const int tid = threadIdx.x;
    for (uint32_t s = blockIdx.x; s < n; s += gridDim.x) {
        const uint64_t lo = offsets[s], hi = offsets[s + 1];
        unsigned long long acc = 0;
        if (abunds) {
            uint64_t p = lo + (uint64_t)tid;
            for (; p + 768 < hi; p += 1024) {
                const unsigned long long a0 = abunds[p], a1 = abunds[p + 256], a2 = abunds[p + 512], a3 = abunds[p + 768];
                acc += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
            }
            for (; p < hi; p += 256) { const unsigned long long a = abunds[p]; acc += a * a; }
#pragma unroll
            for (int off = 32; off; off >>= 1) acc += __shfl_xor(acc, off);
            __syncthreads();                                     // (the previous sketch's sums are read)
            if ((tid & 63) == 0) s_part[tid >> 6] = acc;
            __syncthreads();
        }
        if (tid == 0) {
            common[(uint64_t)s * n + s] = (uint32_t)(hi - lo);
            if (abunds) {
                const unsigned long long t = s_part[0] + s_part[1] + s_part[2] + s_part[3];
                sumsq[s] = t;
                prod[(uint64_t)s * n + s] = t;
            }
        }
    }
}

// bottom-k Jaccard from the counts: size = min(num of the pair, n_i + n_j - common), jaccard = common / max(1, size) (one IEEE
// divide, minhash.rs:624-631); the diagonal is 1.0 (compare.py:33 np.ones)
__global__ __launch_bounds__(256) void num_jaccard_kernel(const uint32_t* __restrict__ common, const uint64_t* __restrict__ offsets,
                                                          const uint32_t* __restrict__ nums, uint32_t n, uint32_t* __restrict__ usize,
                                                          double* __restrict__ jaccard) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (idx >= (uint64_t)n * n) return;
    const uint32_t i = (uint32_t)(idx / n), j = (uint32_t)(idx % n);
    const uint32_t lo = i < j ? i : j;
    const uint64_t ni = offsets[i + 1] - offsets[i], nj = offsets[j + 1] - offsets[j];
    const uint64_t c = common[idx];
    uint64_t u = ni + nj - c;
    if (i == j) u = ni;
    if (u > nums[lo]) u = nums[lo];
    if (usize) usize[idx] = (uint32_t)u;
    if (jaccard) jaccard[idx] = i == j ? 1.0 : (double)c / (double)(u > 1 ? u : 1);
}

}  // namespace

hipError_t compare_num_launch(const uint64_t* d_hashes, const uint64_t* d_offsets, const uint32_t* d_nums, uint32_t n,
                              uint32_t* d_common, uint32_t* d_union, double* d_jaccard, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const uint32_t nt = (n + XT - 1) / XT;
    hipLaunchKernelGGL((compare_ext_kernel<X_NUM>), dim3(nt, nt), dim3(XT * XT), 0, stream, d_hashes, (const uint64_t*)nullptr, d_offsets,
                       d_nums, n, d_common, (unsigned long long*)nullptr, XSLICE);
    hipLaunchKernelGGL(ext_rows_kernel, dim3(n < 4096u ? n : 4096u), dim3(256), 0, stream, (const uint64_t*)nullptr, d_offsets, n, d_common,
                       (unsigned long long*)nullptr, (unsigned long long*)nullptr);
    if (d_union || d_jaccard)
        hipLaunchKernelGGL(num_jaccard_kernel, dim3((unsigned)(((uint64_t)n * n + 255) / 256)), dim3(256), 0, stream, d_common, d_offsets,
                           d_nums, n, d_union, d_jaccard);
    return hipGetLastError();
}

hipError_t compare_abund_launch(const uint64_t* d_hashes, const uint64_t* d_abunds, const uint64_t* d_offsets, uint32_t n,
                                bool narrow, uint32_t* d_common, unsigned long long* d_prod, unsigned long long* d_sumsq,
                                hipStream_t stream, uint64_t total_known) {
    if (n == 0) return hipSuccess;
    // Round 5: the sums come from joins of per-block lists sorted by hash (abund_pairs.hip: work grows with the MATCHES, not with
    // pairs x lengths).  SMG_COMPARE_ABUND=walk keeps the per-pair walk below (tests run both against the oracle); it also serves
    // collections of 2^32 elements or more.
    static const bool walk_only = [] { const char* e = getenv("SMG_COMPARE_ABUND"); return e && !strcmp(e, "walk"); }();
    if (!walk_only) {
        // the number of elements sizes the lists: callers that packed the collection pass it; the raw entry point without it reads
        // offsets[n] back, which BLOCKS the caller until the stream has drained (documented in the header)
        uint64_t total = total_known;
        if (total == 0) {
            hipError_t et = hipMemcpyAsync(&total, d_offsets + n, 8, hipMemcpyDeviceToHost, stream);
            if (et == hipSuccess) et = hipStreamSynchronize(stream);
            if (et != hipSuccess) return et;
        }
        const hipError_t ej = abund_pairs_launch(d_hashes, d_abunds, d_offsets, n, total, narrow, d_common, d_prod, stream);
        if (ej == hipSuccess) {
            hipLaunchKernelGGL(ext_rows_kernel, dim3(n < 4096u ? n : 4096u), dim3(256), 0, stream, d_abunds, d_offsets, n, d_common, d_prod, d_sumsq);
            return hipGetLastError();
        }
        if (ej != hipErrorNotSupported) return ej;
    }
    const uint32_t nt = (n + XT - 1) / XT;
    // (sliced tiles add their parts up: the matrices start from zero)
    hipError_t e = hipMemsetAsync(d_common, 0, (size_t)n * n * 4, stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(d_prod, 0, (size_t)n * n * 8, stream);
    if (e != hipSuccess) return e;
    // slice length: tiles of ordinary sketches are cut as well when the grid is small -- a 1,000-sketch collection is 2,016 tiles
    // on 1,536 resident workgroups (two rounds, the second a third full); three slices per tile fill the rounds (5.83 -> 5.47 ms at C3)
    const uint32_t tiles = nt * (nt + 1) / 2;
    uint32_t slice_len = tiles >= 16384 ? XSLICE : XSLICE_SMALL;
    if (narrow)
        hipLaunchKernelGGL((compare_ext_kernel<X_ABUND32>), dim3(nt, nt, XZMAX), dim3(XT * XT), 0, stream, d_hashes, d_abunds, d_offsets,
                           (const uint32_t*)nullptr, n, d_common, d_prod, slice_len);
    else
        hipLaunchKernelGGL((compare_ext_kernel<X_ABUND64>), dim3(nt, nt, XZMAX), dim3(XT * XT), 0, stream, d_hashes, d_abunds, d_offsets,
                           (const uint32_t*)nullptr, n, d_common, d_prod, slice_len);
    hipLaunchKernelGGL(ext_rows_kernel, dim3(n < 4096u ? n : 4096u), dim3(256), 0, stream, d_abunds, d_offsets, n, d_common, d_prod, d_sumsq);
    return hipGetLastError();
}

// ---- lists with several scaled values: every pair at ITS coarser scaled (minhash.rs:688-696, 777-798) --------------------------
// Downsampling a FracMinHash sketch to a coarser scaled keeps the hashes <= max_hash(scaled): a PREFIX of the sorted row.  The
// host works out, per scaled value of the list, how long that prefix is in every row that is as fine or finer (binary
// searches in the host copies), the rows' prefixes are gathered into a CSR of their own on the device (no host sketch objects,
// no second upload), the ordinary compare runs on it, and the entries of the pairs whose coarser scaled is this value go
// to their places in the one n x n matrix.
__global__ __launch_bounds__(256) void csr_prefix_gather_kernel(const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ src_start,
                                                                const uint64_t* __restrict__ new_off, uint32_t m, uint64_t* __restrict__ out) {
    const uint32_t r = blockIdx.x;
    if (r >= m) return;
    const uint64_t s = src_start[r], d = new_off[r], len = new_off[r + 1] - d;
    for (uint64_t i = threadIdx.x; i < len; i += 256) out[d + i] = hashes[s + i];
}

__global__ __launch_bounds__(256) void class_scatter_kernel(const uint32_t* __restrict__ sub, uint32_t m, const uint32_t* __restrict__ rows,
                                                            const uint32_t* __restrict__ class_of, uint32_t cls, uint32_t n,
                                                            uint32_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (uint64_t)m * m) return;
    const uint32_t a = (uint32_t)(i / m), b = (uint32_t)(i % m);
    const uint32_t ra = rows[a], rb = rows[b];
    const uint32_t ca = class_of[ra], cb = class_of[rb];
    if ((ca > cb ? ca : cb) == cls) out[(uint64_t)ra * n + rb] = sub[i];
}

hipError_t csr_prefix_gather_launch(const uint64_t* d_hashes, const uint64_t* d_src_start, const uint64_t* d_new_off, uint32_t m,
                                    uint64_t* d_out, hipStream_t stream) {
    if (m == 0) return hipSuccess;
    hipLaunchKernelGGL(csr_prefix_gather_kernel, dim3(m), dim3(256), 0, stream, d_hashes, d_src_start, d_new_off, m, d_out);
    return hipGetLastError();
}

hipError_t class_scatter_launch(const uint32_t* d_sub, uint32_t m, const uint32_t* d_rows, const uint32_t* d_class_of, uint32_t cls,
                                uint32_t n, uint32_t* d_out, hipStream_t stream) {
    if (m == 0) return hipSuccess;
    hipLaunchKernelGGL(class_scatter_kernel, dim3((unsigned)(((uint64_t)m * m + 255) / 256)), dim3(256), 0, stream, d_sub, m, d_rows,
                       d_class_of, cls, n, d_out);
    return hipGetLastError();
}

}  // namespace smg
