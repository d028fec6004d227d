// sketch_words.hip -- DNA sketching for k-mers longer than the register-window kernel holds (k = 129 .. WORDS_MAX_K), round 5.
//
//   src/core/src/signature.rs:246-306  SeqToHashes::next (DNA) has no k cliff and no upper limit;
//   src/core/src/signature.rs:38-58    add_sequence: skip hash 0;  src/core/src/sketch/minhash.rs:319: keep h <= max_hash.
//
// A workgroup stages a stretch of WORDS_TILE start positions + k - 1 bytes in LDS -- upper-cased, and once more as its reverse
// complement; a bit per byte that is not ACGT, the count of such bytes in front of every 32-byte word -- then every lane hashes the
// k-mers at p = tid, tid + 256, ... through kmer_words.hpp: five dword reads + four byte-aligns per 16 key bytes of either strand,
// neighbouring lanes reading neighbouring bytes of the same dwords.  k is a run-time value: one kernel for every length, LDS
// sized at launch.
// Work per k-mer grows with k (MurmurHash3 is 4 64-bit multiplies per 16 bytes), so this path is slower than the unrolled
// kernels by what the run-time addressing costs, not by an order of magnitude (profiles/r05_long_k.json).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include "kmer_words.hpp"
#include "sketch_kernel.hpp"
#include "device_api.hpp"

namespace smg {

namespace {

constexpr int WORDS_TILE = 4096;             // start positions per stretch (16 per lane)
constexpr uint32_t WORDS_MAX_K = 60000;      // LDS: two copies of the stretch + masks = 2.25 (TILE + k) + the 16 KB of kept hashes <= 160 KB

struct WordsGeom {
    uint32_t n_chunks;       // 16-byte chunks of the stretch (TILE + k - 1 bytes, rounded up)
    uint32_t n_words;        // 32-byte words of the bad-byte mask
    uint32_t win_dwords;     // the stretch, its reverse complement, slack behind both
    size_t lds;
};

WordsGeom words_geometry(uint32_t k) {
    WordsGeom g;
    const uint32_t bytes = (uint32_t)WORDS_TILE + k - 1u;
    g.n_chunks = (bytes + 15u) / 16u;
    g.n_words = (g.n_chunks + 1u) / 2u + 1u;                 // + 1: upto(p + k) may name the word behind the last byte
    g.win_dwords = ww_layout(g.n_chunks).dwords();
    g.lds = (size_t)g.win_dwords * 4 + (size_t)g.n_words * 8 + 16;
    return g;
}

__global__ __launch_bounds__(SK_BLOCK) void sketch_dna_words_kernel(
    const uint8_t* __restrict__ seq, uint64_t len, uint32_t k, uint64_t seed, uint64_t thr, uint64_t* __restrict__ out,
    unsigned long long* __restrict__ out_count, uint64_t out_cap, uint64_t n_tiles, uint32_t skip, int dense, WordsGeom g) {
    // seq is 16-byte aligned; its first `skip` (< 16) bytes precede the caller's buffer and count as invalid; len includes them
    extern __shared__ __attribute__((aligned(16))) uint32_t ww_lds[];
    uint32_t* const s_win = ww_lds;                              // [win_dwords]
    uint32_t* const s_bits = s_win + g.win_dwords;               // [n_words]
    uint32_t* const s_before = s_bits + g.n_words;               // [n_words]
    __shared__ uint32_t s_wave[SK_BLOCK / 64];
    __shared__ uint64_t s_out[SK_OUT_CAP];                       // kept hashes wait here: one global atomic per flush, not per wave
    __shared__ unsigned int s_cnt;
    __shared__ unsigned long long s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const WwBad bad{s_bits, s_before};
    const WwLayout lay = ww_layout(g.n_chunks);
    if (tid == 0) s_cnt = 0;
    auto flush = [&](unsigned int at_least) {                    // (called by every thread, between barriers)
        __syncthreads();
        const unsigned int cnt = s_cnt;
        if (cnt < at_least || cnt == 0) return;
        const unsigned int n = cnt < (unsigned)SK_OUT_CAP ? cnt : (unsigned)SK_OUT_CAP;
        if (tid == 0) s_base = atomicAdd(out_count, (unsigned long long)n);
        __syncthreads();
        const unsigned long long b = s_base;
        for (unsigned int i = (unsigned)tid; i < n; i += SK_BLOCK)
            if (b + i < out_cap) out[b + i] = s_out[i];
        __syncthreads();
        if (tid == 0) s_cnt = 0;
    };

    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t base = tile * (uint64_t)WORDS_TILE;
        __syncthreads();                                         // the previous stretch's readers are done
        // ---- stage: 16-byte chunks, upper-cased, zero (= invalid) past the end; chunk c of the stretch is chunk n - 1 - c of the
        //      reverse complement, its dwords in reverse order (the slack behind the copies is read and masked away: left as it is) ----
        for (uint32_t c = (uint32_t)tid; c < g.n_chunks; c += SK_BLOCK) {
            const uint64_t off = base + (uint64_t)c * 16;
            uint32_t w[4] = {0, 0, 0, 0};
            if (off + 16 <= len) {
                const uint4 v = *reinterpret_cast<const uint4*>(seq + off);
                w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
            } else if (off < len) {
                for (uint64_t b = off; b < len; ++b) w[(b - off) >> 2] |= (uint32_t)seq[b] << (8 * ((b - off) & 3));
            }
            if (off == 0 && skip)                                // blank the alignment prefix
                for (uint32_t b = 0; b < skip; ++b) w[b >> 2] &= ~(0xffu << (8 * (b & 3)));
            uint32_t nib = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                w[i] &= 0xdfdfdfdfu;                             // upper-case (signature.rs:214)
                nib |= ww_bad4(w[i]) << (4 * i);
            }
            *reinterpret_cast<uint4*>(&s_win[c * 4u]) = make_uint4(w[0], w[1], w[2], w[3]);
            *reinterpret_cast<uint4*>(&s_win[lay.rc_off / 4u + (g.n_chunks - 1u - c) * 4u]) =
                make_uint4(ww_revcomp4(w[3]), ww_revcomp4(w[2]), ww_revcomp4(w[1]), ww_revcomp4(w[0]));
            reinterpret_cast<uint16_t*>(s_bits)[c] = (uint16_t)nib;
        }
        for (uint32_t c = g.n_chunks + (uint32_t)tid; c < g.n_words * 2u; c += SK_BLOCK) reinterpret_cast<uint16_t*>(s_bits)[c] = 0;
        __syncthreads();
        // ---- bad bytes in front of every mask word: every thread sums a run of words, the runs' sums are scanned ----
        const uint32_t per = (g.n_words + SK_BLOCK - 1u) / SK_BLOCK;
        const uint32_t w0 = (uint32_t)tid * per, w1 = w0 + per < g.n_words ? w0 + per : g.n_words;
        uint32_t mine = 0;
        for (uint32_t w = w0; w < w1; ++w) mine += (uint32_t)__popc(s_bits[w]);
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t front = incl - mine, total = 0;
#pragma unroll
        for (int v = 0; v < SK_BLOCK / 64; ++v) {
            const uint32_t s = s_wave[v];
            if (v < wave) front += s;
            total += s;
        }
        for (uint32_t w = w0; w < w1; ++w) { s_before[w] = front; front += (uint32_t)__popc(s_bits[w]); }
        __syncthreads();
        // ---- the k-mers ----
#pragma unroll 1
        for (uint32_t p = (uint32_t)tid; p < (uint32_t)WORDS_TILE; p += SK_BLOCK) {
            const bool ok = base + p + k <= len && (total == 0u || bad.clean(p, k));
            if (!any_lane(ok)) continue;
            const uint64_t h = ww_hash(s_win, lay, p, k, seed);
            const bool keep = ok && (h - 1) < thr;               // h != 0 (signature.rs:50) and h <= thr (minhash.rs:319)
            if (dense) {
                const uint64_t pos = base + p - skip;            // (a good k-mer never starts in the prefix)
                if (keep && pos < out_cap) out[pos] = h;
                continue;
            }
            if (keep) {
                const unsigned int idx = atomicAdd(&s_cnt, 1u);
                if (idx < (unsigned)SK_OUT_CAP) {
                    s_out[idx] = h;
                } else {                                         // pathological density (scaled == 1): straight to HBM
                    const unsigned long long at = atomicAdd(out_count, 1ull);
                    if (at < out_cap) out[at] = h;
                }
            }
        }
        if (!dense) flush((unsigned)SK_OUT_CAP / 2);             // (the next stretch's first barrier orders the reset of s_cnt)
    }
    if (!dense) flush(1u);
}

}  // namespace

// k = 129 .. WORDS_MAX_K; dense: out[i] = hash of the k-mer starting at i (pre-zeroed by the caller), else kept hashes appended
hipError_t sketch_dna_words_launch(const uint8_t* d_seq, uint64_t len, uint32_t k, uint64_t seed, uint64_t thr, uint64_t* d_out,
                                   unsigned long long* d_count, uint64_t cap, bool dense, hipStream_t stream) {
    if (k < 16u || k > WORDS_MAX_K) return hipErrorInvalidValue;
    const uint32_t skip = (uint32_t)((uintptr_t)d_seq & 15);
    d_seq -= skip;
    len += skip;
    const uint64_t n_tiles = (len + WORDS_TILE - 1) / WORDS_TILE;
    if (n_tiles == 0) return hipSuccess;
    if (k > sketch_dna_max_k()) return hipErrorInvalidValue;      // (the host entry points say so in words before they get here)
    const WordsGeom g = words_geometry(k);
    if (g.lds > 48 * 1024) {                                     // more dynamic LDS than a kernel gets unasked: allow what the longest k of THIS device needs, once
        static std::once_flag once;
        static hipError_t allowed = hipSuccess;
        std::call_once(once, [] {
            allowed = hipFuncSetAttribute((const void*)sketch_dna_words_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)words_geometry(sketch_dna_max_k()).lds);
        });
        if (allowed != hipSuccess) return allowed;
    }
    const uint64_t max_blocks = 256ull * 8;
    const unsigned grid = (unsigned)(n_tiles < max_blocks ? n_tiles : max_blocks);
    hipLaunchKernelGGL(sketch_dna_words_kernel, dim3(grid), dim3(SK_BLOCK), g.lds, stream, d_seq, len, k, seed, thr, d_out, d_count, cap,
                       n_tiles, skip, dense ? 1 : 0, g);
    return hipGetLastError();
}

// The longest k-mer the device at hand can take: the kernel's LDS (stretch, reverse complement, masks) + its static 16 KB of kept
// hashes must fit what the device gives a workgroup (160 KB on MI355X -> WORDS_MAX_K; a 64 KB part ~ 17,000).  Asked once.
uint32_t sketch_dna_max_k() {
    static const uint32_t max_k = [] {
        int dev = 0, lds = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || lds <= 0) {
            (void)hipGetLastError();
            lds = 64 * 1024;
        }
        const size_t room = (size_t)lds;
        auto fits = [&](uint32_t k) { return words_geometry(k).lds + sizeof(uint64_t) * SK_OUT_CAP + 64 <= room; };
        uint32_t lo = 16, hi = WORDS_MAX_K;                          // the largest k in [16, WORDS_MAX_K] that fits (lds grows with k)
        if (!fits(lo)) return 0u;
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo + 1) / 2;
            if (fits(mid)) lo = mid; else hi = mid - 1;
        }
        return lo;
    }();
    return max_k;
}

}  // namespace smg
