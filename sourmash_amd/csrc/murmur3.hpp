// MurmurHash3_x64_128 (low 64 bits), host + gfx950 device.
//
// Replaces: src/core/src/lib.rs:57-59 `_hash_murmur(kmer, seed) =
// murmurhash3_x64_128(kmer, seed).0` (crate murmurhash3 0.0.5, un-vendored; the
// algorithm is Appleby's public-domain MurmurHash3_x64_128 with the u64 seed
// loaded into both h1 and h2) and its C export `hash_murmur`
// (src/core/src/ffi/mod.rs:22-31, include/sourmash.h:133).
//
// Two forms:
//   * mmh3_h1_bytes(ptr, len, seed)      -- any length, host and device.
//   * mmh3_h1_words<K>(w[], seed)        -- K known at compile time, the key
//     already assembled as little-endian 32-bit words, zero padded.  This is
//     what the sketch kernel calls once per k-mer: 12 64-bit multiplies for
//     K = 31 (4 in the 16-byte block, 4 in the 15-byte tail, 4 in fmix64 x2).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SMG_HD __host__ __device__ __forceinline__
#else
#define SMG_HD inline
#endif

namespace smg {

constexpr uint64_t MMH3_C1 = 0x87c37b91114253d5ULL;
constexpr uint64_t MMH3_C2 = 0x4cf5ad432745937fULL;

// rotl64 with a compile-time count.  On the device it is spelled as two v_alignbit_b32 on the
// 32-bit halves: left to itself hipcc folds the rotate into the preceding constant multiply
// (k * c rotated = two separate wide multiplies, 9 instructions instead of 4 + 2).
template <int R>
SMG_HD uint64_t rotl64(uint64_t x) {
    static_assert(R > 0 && R < 64, "rotate count");
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    uint32_t nlo, nhi;
    if constexpr (R == 32) { nlo = hi; nhi = lo; }
    else if constexpr (R < 32) {
        nhi = __builtin_amdgcn_alignbit(hi, lo, 32 - R);
        nlo = __builtin_amdgcn_alignbit(lo, hi, 32 - R);
    } else {
        nhi = __builtin_amdgcn_alignbit(lo, hi, 64 - R);
        nlo = __builtin_amdgcn_alignbit(hi, lo, 64 - R);
    }
    uint64_t r = ((uint64_t)nhi << 32) | nlo;
    asm("" : "+v"(r));     // keep the pair a pair: otherwise a following 64-bit add is split in two
    return r;
#else
    return (x << R) | (x >> (64 - R));
#endif
}

// h * 5 + c.  Device: two v_lshl_add_u64 (hipcc would emit two v_mad_u64_u32 plus a move).
SMG_HD uint64_t mul5_add(uint64_t h, uint64_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t t;
    asm("v_lshl_add_u64 %0, %1, 2, %1" : "=v"(t) : "v"(h));
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(t) : "v"(t), "v"(c));
    return t;
#else
    return h * 5 + c;
#endif
}

SMG_HD uint64_t fmix64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

// fmix64 split around its last multiply.  fmix64(k) == fmix64_tail(fmix64_head(k)); the top dword of the result is
// the top dword of the last product (the closing xor-shift by 33 leaves it alone), so a scaled sketch can reject a
// k-mer from fmix64_tail_hi() of both halves without finishing either (kmer_core.hpp, process_lane).
SMG_HD uint64_t fmix64_head(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    return k;
}
SMG_HD uint64_t fmix64_tail(uint64_t k) {
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}
SMG_HD uint32_t fmix64_tail_hi(uint64_t k) {
    const uint32_t lo = (uint32_t)k, hi = (uint32_t)(k >> 32);
    const uint32_t clo = 0x1a85ec53u, chi = 0xc4ceb9feu;
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(lo, clo) + lo * chi + hi * clo;
#else
    return (uint32_t)(((uint64_t)lo * clo) >> 32) + lo * chi + hi * clo;
#endif
}

SMG_HD void mmh3_block(uint64_t& h1, uint64_t& h2, uint64_t k1, uint64_t k2) {
    k1 *= MMH3_C1; k1 = rotl64<31>(k1); k1 *= MMH3_C2; h1 ^= k1;
    h1 = rotl64<27>(h1); h1 += h2; h1 = mul5_add(h1, 0x52dce729);
    k2 *= MMH3_C2; k2 = rotl64<33>(k2); k2 *= MMH3_C1; h2 ^= k2;
    h2 = rotl64<31>(h2); h2 += h1; h2 = mul5_add(h2, 0x38495ab5);
}

SMG_HD uint64_t mmh3_finish(uint64_t h1, uint64_t h2, uint64_t len) {
    h1 ^= len; h2 ^= len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    return h1 + h2;
}

// The hash with the last multiply of both fmix64 left undone: h == fmix64_tail(a) + fmix64_tail(b).
struct Mmh3Open { uint64_t a, b; };
SMG_HD Mmh3Open mmh3_finish_open(uint64_t h1, uint64_t h2, uint64_t len) {
    h1 ^= len; h2 ^= len;
    h1 += h2; h2 += h1;
    return Mmh3Open{fmix64_head(h1), fmix64_head(h2)};
}
SMG_HD uint64_t mmh3_close(Mmh3Open o) { return fmix64_tail(o.a) + fmix64_tail(o.b); }
// top dword of mmh3_close(o), short of the carry out of the low dwords: the true value is this or this + 1
SMG_HD uint32_t mmh3_close_hi(Mmh3Open o) { return fmix64_tail_hi(o.a) + fmix64_tail_hi(o.b); }

// Key given as zero-padded little-endian dwords w[0 .. ceil(K/4)-1].
template <int K>
SMG_HD Mmh3Open mmh3_open_words(const uint32_t* w, uint64_t seed) {
    constexpr int NB = K / 16;     // full 16-byte blocks
    constexpr int T = K % 16;      // tail bytes
    constexpr int NW = (K + 3) / 4;
    uint64_t h1 = seed, h2 = seed;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        uint64_t k1 = (uint64_t)w[4 * b] | ((uint64_t)w[4 * b + 1] << 32);
        uint64_t k2 = (uint64_t)w[4 * b + 2] | ((uint64_t)w[4 * b + 3] << 32);
        mmh3_block(h1, h2, k1, k2);
    }
    constexpr int tb = 4 * NB;
    if (T > 8) {
        uint64_t k2 = (uint64_t)w[tb + 2];
        if (tb + 3 < NW) k2 |= (uint64_t)w[tb + 3] << 32;
        k2 *= MMH3_C2; k2 = rotl64<33>(k2); k2 *= MMH3_C1; h2 ^= k2;
    }
    if (T > 0) {
        uint64_t k1 = (uint64_t)w[tb];
        if (tb + 1 < NW) k1 |= (uint64_t)w[tb + 1] << 32;
        k1 *= MMH3_C1; k1 = rotl64<31>(k1); k1 *= MMH3_C2; h1 ^= k1;
    }
    return mmh3_finish_open(h1, h2, (uint64_t)K);
}
template <int K>
SMG_HD uint64_t mmh3_h1_words(const uint32_t* w, uint64_t seed) { return mmh3_close(mmh3_open_words<K>(w, seed)); }

// Any length, byte pointer (host `hash_murmur`, `add_word`; generic-k kernel).
SMG_HD uint64_t mmh3_h1_bytes(const uint8_t* data, uint64_t len, uint64_t seed) {
    uint64_t h1 = seed, h2 = seed;
    const uint64_t nblocks = len / 16;
    for (uint64_t i = 0; i < nblocks; ++i) {
        uint64_t k1 = 0, k2 = 0;
        for (int j = 7; j >= 0; --j) {
            k1 = (k1 << 8) | data[16 * i + j];
            k2 = (k2 << 8) | data[16 * i + 8 + j];
        }
        mmh3_block(h1, h2, k1, k2);
    }
    const uint8_t* tail = data + 16 * nblocks;
    const int t = (int)(len & 15);
    if (t > 8) {
        uint64_t k2 = 0;
        for (int j = t - 1; j >= 8; --j) k2 = (k2 << 8) | tail[j];
        k2 *= MMH3_C2; k2 = rotl64<33>(k2); k2 *= MMH3_C1; h2 ^= k2;
    }
    if (t > 0) {
        uint64_t k1 = 0;
        for (int j = (t > 8 ? 8 : t) - 1; j >= 0; --j) k1 = (k1 << 8) | tail[j];
        k1 *= MMH3_C1; k1 = rotl64<31>(k1); k1 *= MMH3_C2; h1 ^= k1;
    }
    return mmh3_finish(h1, h2, len);
}

}  // namespace smg
