// residue_core.hpp -- the per-lane work of the residue-window kernel (protein / dayhoff / hp sketches), host + device.
//
// Reference: src/core/src/signature.rs:307-393 -- every window of ksize / 3 residues is hashed with MurmurHash3 (no validity
// test in this mode); here the residues are one byte each in `aa`, the six translations of a DNA record separated by 0xFF
// bytes (protein.hip), and windows touching a separator do not exist.
//
// Round 4's kernel gave every window a lane that copied its k bytes into a scratch array and hashed them byte by byte:
// 18 G windows/s where the DNA kernel does 310 G k-mers/s.  Here a lane owns RW_P = 8 consecutive window starts -- one aligned
// 8-byte word of `aa` -- loads the 2 NB + 3 words its windows reach into, and window j is those words shifted down by j
// bytes (constant shifts: the loop over j is unrolled), hashed from registers with the block / tail structure of
// MurmurHash3_x64_128 (NB = k / 16 full blocks known at compile time, the tail's byte masks computed once per launch).
// Separators are rare (six per translated record): a lane whose words hold none -- one SWAR test per word -- skips the per-window
// check altogether.
#pragma once
#include <stdint.h>
#include <type_traits>
#include "murmur3.hpp"

namespace smg {

constexpr int RW_P = 8;                 // window starts per lane: the bytes of one aligned word
constexpr int RW_MAX_NB = 4;            // k <= 16 * 4 + 15 = 79 residues take this path

// 0x80 in every byte of w that equals 0xFF (exact per byte: no borrow crosses bytes)
SMG_HD uint64_t ff_bytes(uint64_t w) {
    const uint64_t v = ~w, m = 0x7f7f7f7f7f7f7f7full;
    return ~(((v & m) + m) | v | m);
}

// words a..b (b the higher addresses) as one 128-bit little-endian value shifted down by J bytes, low 64 bits
template <int J>
SMG_HD uint64_t funnel(uint64_t a, uint64_t b) {
    if constexpr (J == 0) return a;
    else return (a >> (8 * J)) | (b << (64 - 8 * J));
}

struct ResidueTail { uint32_t k, r; uint64_t mask1, mask2; };     // r = k & 15; byte masks of the tail's k1 / k2 words
SMG_HD ResidueTail residue_tail(uint32_t k) {
    ResidueTail t;
    t.k = k; t.r = k & 15u;
    const uint32_t r1 = t.r < 8u ? t.r : 8u, r2 = t.r > 8u ? t.r - 8u : 0u;
    t.mask1 = r1 == 8u ? ~0ull : ((1ull << (8u * r1)) - 1ull);
    t.mask2 = r2 == 0u ? 0ull : ((1ull << (8u * r2)) - 1ull);      // r2 <= 7
    return t;
}

// Window j of a lane: W[0 .. 2 NB + 2] are the lane's words; -> the hash, and whether a separator lies in the window (only
// evaluated when check_sep: some loaded word holds a separator)
template <int NB, int J>
SMG_HD uint64_t residue_window_hash(const uint64_t* W, const uint64_t* F, const ResidueTail& t, uint64_t seed, bool check_sep, bool* has_sep) {
    uint64_t h1 = seed, h2 = seed;
    uint64_t sep = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const uint64_t k1 = funnel<J>(W[2 * b], W[2 * b + 1]), k2 = funnel<J>(W[2 * b + 1], W[2 * b + 2]);
        mmh3_block(h1, h2, k1, k2);
        if (check_sep) sep |= funnel<J>(F[2 * b], F[2 * b + 1]) | funnel<J>(F[2 * b + 1], F[2 * b + 2]);
    }
    if (t.r > 8u) {
        uint64_t k2 = funnel<J>(W[2 * NB + 1], W[2 * NB + 2]) & t.mask2;
        k2 *= MMH3_C2; k2 = rotl64<33>(k2); k2 *= MMH3_C1; h2 ^= k2;
        if (check_sep) sep |= funnel<J>(F[2 * NB + 1], F[2 * NB + 2]) & t.mask2;
    }
    if (t.r > 0u) {
        uint64_t k1 = funnel<J>(W[2 * NB], W[2 * NB + 1]) & t.mask1;
        k1 *= MMH3_C1; k1 = rotl64<31>(k1); k1 *= MMH3_C2; h1 ^= k1;
        if (check_sep) sep |= funnel<J>(F[2 * NB], F[2 * NB + 1]) & t.mask1;
    }
    *has_sep = sep != 0;
    return mmh3_finish(h1, h2, (uint64_t)t.k);
}

// The RW_P windows starting at bytes 8 g .. 8 g + 7 of aa[0, n): emit(window start, hash) for every window that lies inside
// [0, n) and touches no separator.  aa64: `aa` as aligned 8-byte words; words past ceil(n / 8) are not read.
template <int NB, class Emit>
SMG_HD void residue_windows_lane(const uint64_t* aa64, uint64_t n, uint64_t g, const ResidueTail& t, uint64_t seed, Emit&& emit) {
    constexpr int NWORDS = 2 * NB + 3;
    const uint64_t n_words = (n + 7) / 8;
    uint64_t W[NWORDS], F[NWORDS];
    uint64_t any_sep = 0;
#pragma unroll
    for (int m = 0; m < NWORDS; ++m) {
        W[m] = g + (uint64_t)m < n_words ? aa64[g + (uint64_t)m] : 0ull;
        F[m] = ff_bytes(W[m]);
        any_sep |= F[m];
    }
    const bool check_sep = any_sep != 0;
    const uint64_t i0 = 8 * g;
    auto one = [&](auto jc) {
        constexpr int J = decltype(jc)::value;
        if (i0 + (uint64_t)J + (uint64_t)t.k > n) return;
        bool has_sep = false;
        const uint64_t h = residue_window_hash<NB, J>(W, F, t, seed, check_sep, &has_sep);
        if (!has_sep) emit(i0 + (uint64_t)J, h);
    };
    one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{}); one(std::integral_constant<int, 2>{});
    one(std::integral_constant<int, 3>{}); one(std::integral_constant<int, 4>{}); one(std::integral_constant<int, 5>{});
    one(std::integral_constant<int, 6>{}); one(std::integral_constant<int, 7>{});
}

}  // namespace smg
