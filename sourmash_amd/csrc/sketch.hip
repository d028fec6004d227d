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
#include "sketch_kernel.hpp"
#include "device_api.hpp"

namespace smg {

// Any k (1..GENERIC_MAX_K): one lane per start position, bytes read from LDS.
// The plain form of the walk: no default path takes it (SMG_SKETCH_GENERIC=1 selects it for the tests that compare the others with it).
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

// The register-window kernel is instantiated for EVERY ksize 1 .. SK_FAST_MAX_K = 88 (the reference treats all k alike,
// signature.rs:246-306; tests/test_kmer_core_cpu.py checks each instantiation against the oracle on the host): k = 1 .. 64 in
// this unit, 65 .. 88 in sketch_long.hip, the per-position form of all of them in sketch_dense.hip.  Longer k-mers take the
// run-time-k kernel of sketch_words.hip (sketch_kernel.hpp says why the line is drawn at 88).
constexpr int FAST_MAX_K = SK_FAST_MAX_K;
constexpr int FAST_HERE_K = 64;

static hipError_t sketch_any(const uint8_t* d_seq, uint64_t len, uint32_t k, uint64_t seed, uint64_t thr,
                             uint64_t* d_out, unsigned long long* d_count, uint64_t cap, bool dense,
                             hipStream_t stream) {
    if (len < k || k == 0) return hipSuccess;
    static const bool generic_only = [] { const char* e = getenv("SMG_SKETCH_GENERIC"); return e && *e == '1'; }();
    constexpr uint32_t words_from = (uint32_t)FAST_MAX_K + 1u;         // from this k on the run-time-k kernel is taken (it accepts any k >= 16)
    if (k >= words_from && !generic_only) return sketch_dna_words_launch(d_seq, len, k, seed, thr, d_out, d_count, cap, dense, stream);
    if (!dense && k <= (uint32_t)FAST_MAX_K && !generic_only) {
        const sketch_launch_fn f = k <= (uint32_t)FAST_HERE_K ? sparse_launcher_from<0>(k, std::make_integer_sequence<int, FAST_HERE_K>())
                                                             : sparse_launcher_long(k);
        return f(d_seq, len, seed, thr, d_out, d_count, cap, false, stream);
    }
    if (dense && k <= (uint32_t)FAST_MAX_K && !generic_only)          // per-position output (seq_to_hashes): every k <= 88 (sketch_dense.hip)
        return dense_launcher(k)(d_seq, len, seed, thr, d_out, d_count, cap, true, stream);
    // longer k-mers: 16 key bytes at a time from the staged stretch (sketch_words.hip); the byte loop below stays as the form the
    // others are tested against (SMG_SKETCH_GENERIC=1, k <= 256)
    if (!generic_only || k > (uint32_t)GENERIC_MAX_K) return sketch_dna_words_launch(d_seq, len, k, seed, thr, d_out, d_count, cap, dense, stream);
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
