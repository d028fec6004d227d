// sketch.hip -- DNA k-mer sketching kernels for gfx950 (MI355X).
//
// GPU counterpart of the reference's per-record hot loop
//   src/core/src/signature.rs:38-58   SigsTrait::add_sequence
//   src/core/src/signature.rs:246-306 SeqToHashes::next (DNA branch)
//   src/core/src/sketch/minhash.rs:313-383 / 1237-1291 add_hash (keep rule)
// for a whole buffer at once: every k-mer start position is independent, so the
// sequence is cut into tiles of 256 lanes x P positions; the tile's bytes (+ a
// K-1 byte halo) are staged once through LDS with 16-byte coalesced loads, each
// lane pulls its P+K-1 byte window into registers with ds_read_b128 and runs
// smg::process_lane (kmer_core.hpp).  Kept hashes (about 1 in `scaled`) are
// appended to a per-workgroup LDS buffer and flushed to HBM with one global
// atomic per flush, so the single output counter sees a few thousand atomics
// per launch instead of one per kept hash.
//
// Roofline: the kernel reads 1 B/base and writes 8 B per kept hash; it is
// bound by VALU integer issue (12 64-bit multiplies per k-mer), not by HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <utility>
#include "kmer_core.hpp"
#include "device_api.hpp"

namespace smg {

constexpr int SK_BLOCK = 256;      // 4 waves, one per SIMD
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

// Any k (1..GENERIC_MAX_K): one lane per start position, bytes read from LDS.
// Slow path for k values without a specialised instantiation.
constexpr int GENERIC_MAX_K = 256;
constexpr int GENERIC_TILE = 4096;

__global__ __launch_bounds__(SK_BLOCK) void sketch_dna_generic_kernel(
    const uint8_t* __restrict__ seq, uint64_t len, uint32_t k, uint64_t seed, uint64_t thr,
    uint64_t* __restrict__ out, unsigned long long* __restrict__ out_count, uint64_t out_cap, uint64_t n_tiles,
    int dense) {
    __shared__ uint8_t s_in[GENERIC_TILE + GENERIC_MAX_K];
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t base = tile * (uint64_t)GENERIC_TILE;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < GENERIC_TILE + k - 1; i += SK_BLOCK)
            s_in[i] = (base + i < len) ? seq[base + i] : 0;
        __syncthreads();
        for (uint32_t p = threadIdx.x; p < GENERIC_TILE; p += SK_BLOCK) {
            uint8_t fwd[GENERIC_MAX_K], rev[GENERIC_MAX_K];
            bool ok = true;
            for (uint32_t j = 0; j < k; ++j) {
                const uint8_t u = s_in[p + j] & 0xdf;
                uint8_t c;
                switch (u) {
                case 'A': c = 'T'; break;
                case 'C': c = 'G'; break;
                case 'G': c = 'C'; break;
                case 'T': c = 'A'; break;
                default: c = 0; ok = false;
                }
                fwd[j] = u;
                rev[k - 1 - j] = c;
            }
            if (!ok) continue;
            bool use_rev = false;
            for (uint32_t j = 0; j < k; ++j) {
                if (fwd[j] != rev[j]) { use_rev = rev[j] < fwd[j]; break; }
            }
            const uint64_t h = mmh3_h1_bytes(use_rev ? rev : fwd, k, seed);
            if ((h - 1) < thr) {
                if (dense) {
                    if (base + p < out_cap) out[base + p] = h;
                } else {
                    const unsigned long long g = atomicAdd(out_count, 1ull);
                    if (g < out_cap) out[g] = h;
                }
            }
        }
    }
}

// position of the first byte outside ACGTacgt (for force == false), or ~0.
__global__ __launch_bounds__(SK_BLOCK) void first_invalid_kernel(const uint8_t* __restrict__ seq, uint64_t len,
                                                                unsigned long long* __restrict__ first) {
    unsigned long long best = ~0ull;
    for (uint64_t i = (uint64_t)blockIdx.x * SK_BLOCK + threadIdx.x; i < len; i += (uint64_t)gridDim.x * SK_BLOCK) {
        const uint8_t u = seq[i] & 0xdf;
        if (!(u == 'A' || u == 'C' || u == 'G' || u == 'T')) { best = i; break; }   // i only grows: first hit is this lane's min
    }
    if (best != ~0ull) atomicMin(first, best);
}

template <int K, int P>
static hipError_t launch_k(const uint8_t* d_seq, uint64_t len, uint64_t seed, uint64_t thr, uint64_t* d_out,
                           unsigned long long* d_count, uint64_t cap, bool dense, hipStream_t stream) {
    constexpr uint64_t TILE = (uint64_t)SK_BLOCK * P;
    const uint32_t skip = (uint32_t)((uintptr_t)d_seq & 15);     // realign: 16-byte loads need an aligned base
    d_seq -= skip;
    len += skip;
    const uint64_t n_tiles = (len + TILE - 1) / TILE;
    if (n_tiles == 0) return hipSuccess;
    const uint64_t max_blocks = 256ull * 8;   // 256 CUs x 8 resident workgroups
    const unsigned grid = (unsigned)(n_tiles < max_blocks ? n_tiles : max_blocks);
    if (dense)
        hipLaunchKernelGGL((sketch_dna_kernel<K, P, true>), dim3(grid), dim3(SK_BLOCK), 0, stream, d_seq, len, seed,
                           thr, d_out, d_count, cap, n_tiles, skip);
    else
        hipLaunchKernelGGL((sketch_dna_kernel<K, P, false>), dim3(grid), dim3(SK_BLOCK), 0, stream, d_seq, len, seed,
                           thr, d_out, d_count, cap, n_tiles, skip);
    return hipGetLastError();
}

// The register-window kernel is instantiated for EVERY ksize 1 .. 64 (the reference treats all k alike,
// signature.rs:246-306; tests/test_kmer_core_cpu.py checks each instantiation against the oracle on the host): the
// table below is what sketch_dna_launch dispatches through.  Longer k-mers take the byte-wise generic kernel.
constexpr int FAST_MAX_K = 64;
typedef hipError_t (*launch_fn)(const uint8_t*, uint64_t, uint64_t, uint64_t, uint64_t*, unsigned long long*, uint64_t, bool,
                                hipStream_t);
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
template <int... KS>
static launch_fn sparse_launcher(uint32_t k, std::integer_sequence<int, KS...>) {
    static const launch_fn table[] = {&launch_sparse_k<KS + 1>...};
    return table[k - 1];
}

static hipError_t sketch_any(const uint8_t* d_seq, uint64_t len, uint32_t k, uint64_t seed, uint64_t thr,
                             uint64_t* d_out, unsigned long long* d_count, uint64_t cap, bool dense,
                             hipStream_t stream) {
    if (len < k || k == 0) return hipSuccess;
    static const bool generic_only = [] { const char* e = getenv("SMG_SKETCH_GENERIC"); return e && *e == '1'; }();
    if (!dense && k <= (uint32_t)FAST_MAX_K && !generic_only)
        return sparse_launcher(k, std::make_integer_sequence<int, FAST_MAX_K>())(d_seq, len, seed, thr, d_out, d_count, cap, false, stream);
    if (!generic_only) switch (k) {                                    // per-position output (seq_to_hashes): the usual ksizes
    case 21: return launch_k<21, 16>(d_seq, len, seed, thr, d_out, d_count, cap, dense, stream);
    case 31: return launch_k<31, 16>(d_seq, len, seed, thr, d_out, d_count, cap, dense, stream);
    case 51: return launch_k<51, 16>(d_seq, len, seed, thr, d_out, d_count, cap, dense, stream);
    default: break;
    }
    if (k > (uint32_t)GENERIC_MAX_K) return hipErrorInvalidValue;
    const uint64_t n_tiles = (len + GENERIC_TILE - 1) / GENERIC_TILE;
    const uint64_t max_blocks = 256ull * 8;
    const unsigned grid = (unsigned)(n_tiles < max_blocks ? n_tiles : max_blocks);
    hipLaunchKernelGGL(sketch_dna_generic_kernel, dim3(grid), dim3(SK_BLOCK), 0, stream, d_seq, len, k, seed, thr,
                       d_out, d_count, cap, n_tiles, dense ? 1 : 0);
    return hipGetLastError();
}

hipError_t sketch_dna_launch(const uint8_t* d_seq, uint64_t len, uint32_t k, uint64_t seed, uint64_t thr,
                             uint64_t* d_out, unsigned long long* d_count, uint64_t cap, hipStream_t stream) {
    return sketch_any(d_seq, len, k, seed, thr, d_out, d_count, cap, false, stream);
}

hipError_t kmer_hashes_launch(const uint8_t* d_seq, uint64_t len, uint32_t k, uint64_t seed, uint64_t* d_out,
                              uint64_t n_kmers, hipStream_t stream) {
    return sketch_any(d_seq, len, k, seed, ~0ull, d_out, nullptr, n_kmers, true, stream);
}

hipError_t first_invalid_launch(const uint8_t* d_seq, uint64_t len, unsigned long long* d_first, hipStream_t stream) {
    if (len == 0) return hipSuccess;
    const uint64_t nb = (len + SK_BLOCK - 1) / SK_BLOCK;
    const unsigned grid = (unsigned)(nb < 2048 ? nb : 2048);
    hipLaunchKernelGGL(first_invalid_kernel, dim3(grid), dim3(SK_BLOCK), 0, stream, d_seq, len, d_first);
    return hipGetLastError();
}

}  // namespace smg
