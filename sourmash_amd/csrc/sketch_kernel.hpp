// sketch_kernel.hpp -- the register-window DNA sketch kernel (template) and its launcher, shared by the translation units
// that instantiate it: sketch.hip (k = 1 .. 64), sketch_long.hip (k = 65 .. SK_FAST_MAX_K, two parts) and sketch_dense.hip (the
// per-position form of all of them, six parts), so that the fully unrolled instantiations compile side by side.  See sketch.hip
// for the design notes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include "kmer_core.hpp"

namespace smg {

constexpr int SK_BLOCK = 256;      // 4 waves, one per SIMD
// The longest k-mer the unrolled kernel is instantiated for.  Its window lives in registers: from k = 89 on an instantiation needs
// more than 256 of them and runs one wave per SIMD (k = 88: 131 Gbase/s, k = 96: 93), where the run-time-k kernel of
// sketch_words.hip -- 67 registers at any k -- is already faster (k = 96: 106, k = 128: 85 against 68; profiles/r05_long_k.json).
constexpr int SK_FAST_MAX_K = 88;
constexpr int sk_part_size(int first_k_minus_1) { return SK_FAST_MAX_K - first_k_minus_1 < 16 ? SK_FAST_MAX_K - first_k_minus_1 : 16; }
constexpr int SK_OUT_CAP = 2048;   // LDS staging entries for kept hashes (16 KiB)

// DENSE == false: append kept hashes (unordered) to out, count in *out_count.
// DENSE == true : out[i] = hash of the k-mer starting at i (out pre-zeroed by the
//                 caller; bad k-mers and hash 0 stay 0) -- kmerminhash_seq_to_hashes.
template <int K, int P, bool DENSE>
__global__ __launch_bounds__(SK_BLOCK) void sketch_dna_kernel(
    const uint8_t* __restrict__ seq, uint64_t len, uint64_t seed, uint64_t thr,
    uint64_t* __restrict__ out, unsigned long long* __restrict__ out_count, uint64_t out_cap,
    uint64_t n_tiles, uint32_t skip) {
    // seq is 16-byte aligned; its first `skip` (< 16) bytes precede the caller's buffer and are
    // treated as invalid.  len includes them.  DENSE positions are reported relative to seq + skip.
    using G = LaneGeom<K, P>;
    constexpr int TILE = SK_BLOCK * P;                       // start positions per tile
    constexpr int LANE_RD = ((G::NW + 3) / 4) * 4;           // dwords each lane reads (whole b128s)
    constexpr int IN_DW = (SK_BLOCK - 1) * (P / 4) + LANE_RD;  // dwords the tile needs in LDS
    constexpr int IN_CHUNKS = (IN_DW + 3) / 4;               // 16-byte chunks to stage
    static_assert(P % 4 == 0, "lane runs must start dword aligned");

    __shared__ __attribute__((aligned(16))) uint32_t s_in[IN_CHUNKS * 4];
    __shared__ uint64_t s_out[SK_OUT_CAP];
    __shared__ unsigned int s_cnt;
    __shared__ unsigned long long s_base;

    const int tid = threadIdx.x;
    if (tid == 0) s_cnt = 0;

    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t base = tile * (uint64_t)TILE;
        __syncthreads();   // previous tile's readers are done with s_in; s_cnt reset visible
        // ---- stage TILE + halo bytes: coalesced 16-byte loads, zero fill past the end ----
        for (int c = tid; c < IN_CHUNKS; c += SK_BLOCK) {
            const uint64_t off = base + (uint64_t)c * 16;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (off + 16 <= len) {
                v = *reinterpret_cast<const uint4*>(seq + off);
            } else if (off < len) {
                uint32_t w[4] = {0, 0, 0, 0};
                for (uint64_t b = off; b < len; ++b) w[(b - off) >> 2] |= (uint32_t)seq[b] << (8 * ((b - off) & 3));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            if (off == 0 && skip) {                      // blank the alignment prefix
                uint32_t w[4] = {v.x, v.y, v.z, v.w};
                for (uint32_t b = 0; b < skip; ++b) w[b >> 2] &= ~(0xffu << (8 * (b & 3)));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            *reinterpret_cast<uint4*>(&s_in[c * 4]) = v;
        }
        __syncthreads();
        // ---- each lane pulls its window into registers ----
        uint32_t raw[LANE_RD];
        const uint4* wp = reinterpret_cast<const uint4*>(&s_in[tid * (P / 4)]);
        static_assert((P / 4) % 4 == 0 || P == 8 || P == 4, "window must stay 16-byte aligned for P=16,32");
#pragma unroll
        for (int i = 0; i < LANE_RD / 4; ++i) {
            if constexpr ((P / 4) % 4 == 0) {
                const uint4 v = wp[i];
                raw[4 * i] = v.x; raw[4 * i + 1] = v.y; raw[4 * i + 2] = v.z; raw[4 * i + 3] = v.w;
            } else {
                const uint32_t* p32 = &s_in[tid * (P / 4) + 4 * i];
                raw[4 * i] = p32[0]; raw[4 * i + 1] = p32[1]; raw[4 * i + 2] = p32[2]; raw[4 * i + 3] = p32[3];
            }
        }
        process_lane<K, P, !DENSE>(raw, seed, thr, [&](int o, uint64_t h) {
            if constexpr (DENSE) {
                const uint64_t pos = base + (uint64_t)tid * P + (uint64_t)o - skip;   // valid k-mers never start in the prefix
                if (pos < out_cap) out[pos] = h;
                return;
            }
            const unsigned int idx = atomicAdd(&s_cnt, 1u);
            if (idx < (unsigned)SK_OUT_CAP) {
                s_out[idx] = h;
            } else {  // pathological density (e.g. scaled == 1): spill straight to HBM
                const unsigned long long g = atomicAdd(out_count, 1ull);
                if (g < out_cap) out[g] = h;
            }
        });
        if constexpr (DENSE) continue;
        // ---- flush the LDS buffer when it is at least half full ----
        __syncthreads();
        const unsigned int cnt = s_cnt;
        if (cnt >= (unsigned)SK_OUT_CAP / 2) {
            const unsigned int n = cnt < (unsigned)SK_OUT_CAP ? cnt : (unsigned)SK_OUT_CAP;
            if (tid == 0) s_base = atomicAdd(out_count, (unsigned long long)n);
            __syncthreads();
            const unsigned long long b = s_base;
            for (unsigned int i = tid; i < n; i += SK_BLOCK)
                if (b + i < out_cap) out[b + i] = s_out[i];
            __syncthreads();
            if (tid == 0) s_cnt = 0;
        }
    }
    if constexpr (DENSE) return;
    __syncthreads();
    const unsigned int cnt = s_cnt;
    if (cnt) {
        const unsigned int n = cnt < (unsigned)SK_OUT_CAP ? cnt : (unsigned)SK_OUT_CAP;
        if (tid == 0) s_base = atomicAdd(out_count, (unsigned long long)n);
        __syncthreads();
        const unsigned long long b = s_base;
        for (unsigned int i = tid; i < n; i += SK_BLOCK)
            if (b + i < out_cap) out[b + i] = s_out[i];
    }
}


typedef hipError_t (*sketch_launch_fn)(const uint8_t*, uint64_t, uint64_t, uint64_t, uint64_t*, unsigned long long*, uint64_t, bool,
                                       hipStream_t);
// the sparse (append kept hashes) form at P = 16 for one ksize
template <int K>
static hipError_t launch_sparse_k(const uint8_t* d_seq, uint64_t len, uint64_t seed, uint64_t thr, uint64_t* d_out,
                                  unsigned long long* d_count, uint64_t cap, bool, hipStream_t stream) {
    constexpr uint64_t TILE = (uint64_t)SK_BLOCK * 16;
    const uint32_t skip = (uint32_t)((uintptr_t)d_seq & 15);
    d_seq -= skip;
    len += skip;
    const uint64_t n_tiles = (len + TILE - 1) / TILE;
    if (n_tiles == 0) return hipSuccess;
    const uint64_t max_blocks = 256ull * 8;
    const unsigned grid = (unsigned)(n_tiles < max_blocks ? n_tiles : max_blocks);
    hipLaunchKernelGGL((sketch_dna_kernel<K, 16, false>), dim3(grid), dim3(SK_BLOCK), 0, stream, d_seq, len, seed, thr, d_out,
                       d_count, cap, n_tiles, skip);
    return hipGetLastError();
}
// launcher of ksize k0 + 1 + i for i in 0 .. n - 1
template <int K0, int... KS>
static sketch_launch_fn sparse_launcher_from(uint32_t k, std::integer_sequence<int, KS...>) {
    static const sketch_launch_fn table[] = {&launch_sparse_k<K0 + KS + 1>...};
    return table[k - K0 - 1];
}
// the per-position form (kmerminhash_seq_to_hashes: one hash per k-mer start, 0 for bad k-mers) at P = 16 for one ksize
template <int K>
static hipError_t launch_dense_k(const uint8_t* d_seq, uint64_t len, uint64_t seed, uint64_t thr, uint64_t* d_out,
                                 unsigned long long* d_count, uint64_t cap, bool, hipStream_t stream) {
    constexpr uint64_t TILE = (uint64_t)SK_BLOCK * 16;
    const uint32_t skip = (uint32_t)((uintptr_t)d_seq & 15);
    d_seq -= skip;
    len += skip;
    const uint64_t n_tiles = (len + TILE - 1) / TILE;
    if (n_tiles == 0) return hipSuccess;
    const uint64_t max_blocks = 256ull * 8;
    const unsigned grid = (unsigned)(n_tiles < max_blocks ? n_tiles : max_blocks);
    hipLaunchKernelGGL((sketch_dna_kernel<K, 16, true>), dim3(grid), dim3(SK_BLOCK), 0, stream, d_seq, len, seed, thr, d_out,
                       d_count, cap, n_tiles, skip);
    return hipGetLastError();
}
template <int K0, int... KS>
static sketch_launch_fn dense_launcher_from(uint32_t k, std::integer_sequence<int, KS...>) {
    static const sketch_launch_fn table[] = {&launch_dense_k<K0 + KS + 1>...};
    return table[k - K0 - 1];
}
// sketch_dense.hip, compiled as six parts of up to 16 ksizes each: k = 1 .. 16, ..., 81 .. 88
sketch_launch_fn dense_launcher_0(uint32_t k);
sketch_launch_fn dense_launcher_1(uint32_t k);
sketch_launch_fn dense_launcher_2(uint32_t k);
sketch_launch_fn dense_launcher_3(uint32_t k);
sketch_launch_fn dense_launcher_4(uint32_t k);
sketch_launch_fn dense_launcher_5(uint32_t k);
inline sketch_launch_fn dense_launcher(uint32_t k) {
    switch ((k - 1u) / 16u) {
    case 0: return dense_launcher_0(k);
    case 1: return dense_launcher_1(k);
    case 2: return dense_launcher_2(k);
    case 3: return dense_launcher_3(k);
    case 4: return dense_launcher_4(k);
    default: return dense_launcher_5(k);
    }
}
// sketch_long.hip, compiled as two parts: k = 65 .. 80, 81 .. 88
sketch_launch_fn sparse_launcher_long_0(uint32_t k);
sketch_launch_fn sparse_launcher_long_1(uint32_t k);
inline sketch_launch_fn sparse_launcher_long(uint32_t k) { return k <= 80u ? sparse_launcher_long_0(k) : sparse_launcher_long_1(k); }
static_assert(SK_FAST_MAX_K > 80 && SK_FAST_MAX_K <= 96, "the part tables above");

}  // namespace smg
