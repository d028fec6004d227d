// Per-lane DNA k-mer -> canonical -> murmur -> keep logic of the sketch kernel.
//
// Replaces, for one run of P consecutive k-mer start positions:
//   src/core/src/signature.rs:189-233  SeqToHashes::new   (upper-casing, :214)
//   src/core/src/signature.rs:246-306  SeqToHashes::next  (revcomp :263, VALID scan :271-286,
//                                      canonical = min(kmer, krc) on ASCII :302-304, murmur)
//   src/core/src/encodings.rs:85-101,370-377  COMPLEMENT / VALID tables
//   src/core/src/signature.rs:38-58    add_sequence: skip hash 0, add the rest
//   src/core/src/sketch/minhash.rs:319  keep rule h <= max_hash
//
// Formulation (MI355X-first, nothing like the reference's byte loops):
// a lane owns P consecutive start positions and holds the P+K-1 ASCII bytes it
// needs as little-endian dwords in registers.  MurmurHash3 consumes the k-mer
// as little-endian 64-bit words, so
//   * the FORWARD k-mer's hash words are just byte-shifted views of the input
//     registers: one v_alignbyte_b32 per dword (none when the position is
//     dword aligned);
//   * the REVERSE-COMPLEMENT k-mer's words are byte-reversed views of the
//     complemented registers: one v_perm_b32 per dword does shift + reverse;
//   * complement and validity come from a 2-bit code ((c >> 1) & 3 maps
//     A,C,T,G -> 0,1,2,3 for both cases) fed to v_perm_b32 as a 4-entry LUT;
//   * canonical choice = big-endian compare of the first 8 bytes (2 bswaps a
//     side + one 64-bit compare); later bytes are looked at only if some lane
//     of the wave ties, which happens with probability 4^-8 per k-mer.
// No 2-bit packing, no per-base loops, no LDS traffic beyond the initial
// window read.  The murmur multiplies dominate (12 x 64-bit per k-mer).
//
// The same source compiles for the host (perm/alignbyte emulated) so the byte
// plumbing is checked against the oracle on CPU (tests/test_kmer_core_cpu.py)
// before it ever runs on a GPU.
#pragma once
#include "murmur3.hpp"
#include <utility>

namespace smg {

// ---- byte-permute primitives ------------------------------------------------
// perm_b32(hi, lo, sel): byte i of the result = byte sel.byte[i] of the 8-byte
// value {hi:lo} (0-3 -> lo, 4-7 -> hi); selector 0x0c yields 0x00.
SMG_HD uint32_t perm_b32(uint32_t hi, uint32_t lo, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    const uint64_t v = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t s = (sel >> (8 * i)) & 0xff;
        uint32_t b;
        if (s < 8) b = (uint32_t)(v >> (8 * s)) & 0xff;
        else if (s == 0x0c) b = 0;
        else b = 0xff;  // other special selectors are never used here
        r |= b << (8 * i);
    }
    return r;
#endif
}

// ({hi:lo} >> 8*n) truncated to 32 bits, n in 0..3
SMG_HD uint32_t alignbyte_b32(uint32_t hi, uint32_t lo, uint32_t n) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, n);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * n));
#endif
}

SMG_HD bool any_lane(bool p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(p) != 0ull;
#else
    return p;
#endif
}

// bitwise select: mask ? b : c  (v_bitop3_b32 is full rate on gfx950, v_cndmask_b32 is not)
SMG_HD uint32_t bitselect(uint32_t mask, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(mask, b, c, 0xCA);
#else
    return (mask & b) | (~mask & c);
#endif
}

// Make a value opaque to the optimiser at this point (keeps rare-path work inside its branch).
SMG_HD uint32_t opaque(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#endif
    return x;
}

SMG_HD uint32_t bswap32(uint32_t x) { return perm_b32(0u, x, 0x00010203u); }

// LUTs indexed by code = (ascii >> 1) & 3 :  A -> 0, C -> 1, T -> 2, G -> 3
constexpr uint32_t LUT_SELF = 'A' | ('C' << 8) | ('T' << 16) | ((uint32_t)'G' << 24);
constexpr uint32_t LUT_COMP = 'T' | ('G' << 8) | ('A' << 16) | ((uint32_t)'C' << 24);

template <int K, int P>
struct LaneGeom {
    static constexpr int NBYTES = P + K - 1;        // bytes a lane touches
    static constexpr int NW = (NBYTES + 3) / 4;     // dwords holding them
    static constexpr int NWK = (K + 3) / 4;         // dwords of one k-mer
    static constexpr int NCH = (K + 7) / 8;         // 8-byte chunks of one k-mer
    static constexpr uint32_t LAST_MASK = (K % 4) ? ((1u << (8 * (K % 4))) - 1u) : 0xffffffffu;
};

// selector for "4 bytes starting at byte a of {hi:lo}, reversed"
constexpr uint32_t rev_sel(int a) {
    return (uint32_t)(a + 3) | ((uint32_t)(a + 2) << 8) | ((uint32_t)(a + 1) << 16) | ((uint32_t)a << 24);
}
// selector for "nb (1..3) bytes starting at byte a, reversed, zero padded"
constexpr uint32_t rev_sel_partial(int a, int nb) {
    uint32_t s = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t b = (i < nb) ? (uint32_t)(a + nb - 1 - i) : 0x0cu;
        s |= b << (8 * i);
    }
    return s;
}

// One start position `O` (compile time) of the lane window.
template <int K, int P, int O>
struct PosOps {
    using G = LaneGeom<K, P>;

    // forward k-mer dword d: bytes O+4d .. O+4d+3 of the window
    template <int D>
    static SMG_HD uint32_t fwd(const uint32_t* U) {
        constexpr int q = (O >> 2) + D, a = O & 3;
        uint32_t v;
        if constexpr (a == 0) v = U[q];
        else if constexpr (q + 1 < G::NW) v = alignbyte_b32(U[q + 1], U[q], a);
        else v = U[q] >> (8 * a);
        if constexpr (D == G::NWK - 1) v &= G::LAST_MASK;
        return v;
    }

    // reverse-complement k-mer dword d: rc[j] = comp(win[O + K-1 - j]), j = 4d..4d+3
    template <int D>
    static SMG_HD uint32_t rev(const uint32_t* C) {
        constexpr int nb = (K - 4 * D) >= 4 ? 4 : (K - 4 * D);  // valid bytes in this dword
        constexpr int s = O + K - 4 * D - nb;                    // first window byte of the group
        constexpr int q = s >> 2, a = s & 3;
        constexpr uint32_t sel = (nb == 4) ? rev_sel(a) : rev_sel_partial(a, nb);
        if constexpr (q + 1 < G::NW) return perm_b32(C[q + 1], C[q], sel);
        else return perm_b32(0u, C[q], sel);
    }

    template <int CH>
    static SMG_HD uint64_t be_chunk(const uint32_t* W) {
        uint64_t v = (uint64_t)bswap32(W[2 * CH]) << 32;
        if constexpr (2 * CH + 1 < G::NWK) v |= bswap32(W[2 * CH + 1]);
        return v;
    }

    template <int CH>
    static SMG_HD uint64_t be_chunk_opaque(const uint32_t* W) {
        uint64_t v = (uint64_t)bswap32(opaque(W[2 * CH])) << 32;
        if constexpr (2 * CH + 1 < G::NWK) v |= bswap32(opaque(W[2 * CH + 1]));
        return v;
    }

    // Rare path (some lane's first 8 bytes tie): the operands are laundered through `opaque` so
    // the compiler cannot hoist these byte swaps out of the branch and run them for every k-mer.
    template <int CH>
    static SMG_HD void tie_break(const uint32_t* F, const uint32_t* R, bool& tie, bool& gt) {
        if constexpr (CH < G::NCH) {
            const uint64_t bf = be_chunk_opaque<CH>(F), br = be_chunk_opaque<CH>(R);
            gt = gt || (tie && bf > br);
            tie = tie && (bf == br);
            tie_break<CH + 1>(F, R, tie, gt);
        }
    }

    template <int... D>
    static SMG_HD void build(const uint32_t* U, const uint32_t* C, uint32_t* F, uint32_t* R,
                             std::integer_sequence<int, D...>) {
        ((F[D] = fwd<D>(U)), ...);
        ((R[D] = rev<D>(C)), ...);
    }

    // hash of the canonical k-mer at this position, last fmix64 multiplies left open (murmur3.hpp)
    static SMG_HD Mmh3Open hash_open(const uint32_t* U, const uint32_t* C, uint64_t seed) {
        uint32_t F[G::NWK], R[G::NWK];
        build(U, C, F, R, std::make_integer_sequence<int, G::NWK>{});
        const uint64_t bf = be_chunk<0>(F), br = be_chunk<0>(R);
        bool gt = bf > br;          // forward string > revcomp string -> take revcomp
        bool tie = bf == br;
        if (G::NCH > 1 && any_lane(tie)) tie_break<1>(F, R, tie, gt);
        uint32_t W[G::NWK];
        const uint32_t m = gt ? 0xffffffffu : 0u;
#pragma unroll
        for (int d = 0; d < G::NWK; ++d) W[d] = bitselect(m, R[d], F[d]);
        return mmh3_open_words<K>(W, seed);
    }
};

// Expand a nonzero-byte pattern of a dword into 4 bits (bit i = byte i != 0).
SMG_HD uint32_t nonzero_bytes4(uint32_t x) {
    uint32_t t = x | (x >> 4);
    t |= t >> 2;
    t |= t >> 1;
    t &= 0x01010101u;
    return (t * 0x01020408u) >> 24 & 0xfu;
}

// Process the P start positions of one lane.
//   raw[NW]  : the lane's window bytes as little-endian dwords (any case, any junk;
//              bytes past the end of the sequence must be non-ACGT, e.g. 0)
//   thr      : keep iff 1 <= h <= thr   (thr = max_hash, or 2^64-1 for num sketches)
//   emit(o,h): called for every kept k-mer (o = position within the lane's run)
//
// EARLY: test the top dword of the hash first and finish it only when some lane of the wave may keep its k-mer
// (1 wave-step in 16 at scaled = 1000); the result is the same either way.  Dense callers (every hash wanted) turn
// it off.
template <int K, int P, bool EARLY, class Emit, int... O>
SMG_HD void process_lane_impl(const uint32_t* raw, uint64_t seed, uint64_t thr, Emit&& emit,
                              std::integer_sequence<int, O...>) {
    using G = LaneGeom<K, P>;
    // top dword t of a kept hash satisfies t <= thr >> 32; the open form knows t or t - 1 (mod 2^32)
    const uint32_t thr_hi = (uint32_t)(thr >> 32);
    const uint32_t lim = thr_hi >= 0xfffffffeu ? 0xffffffffu : thr_hi + 1u;
    uint32_t U[G::NW], C[G::NW];
    uint32_t anybad = 0;
#pragma unroll
    for (int i = 0; i < G::NW; ++i) {
        const uint32_t u = raw[i] & 0xdfdfdfdfu;            // upper-case (signature.rs:214)
        const uint32_t code = (u >> 1) & 0x03030303u;
        U[i] = u;
        C[i] = perm_b32(0u, LUT_COMP, code);                 // encodings.rs:85-101
        uint32_t bad = perm_b32(0u, LUT_SELF, code) ^ u;     // != 0 where byte not in ACGT (encodings.rs:370-377)
        if (i == G::NW - 1 && (G::NBYTES % 4) != 0)          // ignore slack bytes past the lane's window
            bad &= (1u << (8 * (G::NBYTES % 4))) - 1u;
        anybad |= bad;
    }
    // bit b of (badlo, badhi, badtop) = window byte b is invalid (192 bits: k <= 128 at P = 16 ... 64).  Rare: built only if needed.
    uint64_t badlo = 0, badhi = 0, badtop = 0;
    if (anybad != 0) {
#pragma unroll
        for (int i = 0; i < G::NW; ++i) {
            const uint32_t u = U[i];
            const uint32_t code = (u >> 1) & 0x03030303u;
            uint32_t bad = perm_b32(0u, LUT_SELF, code) ^ u;
            if (i == G::NW - 1 && (G::NBYTES % 4) != 0) bad &= (1u << (8 * (G::NBYTES % 4))) - 1u;
            const uint64_t nib = nonzero_bytes4(bad);
            if (4 * i < 64) badlo |= nib << (4 * i);
            else if (4 * i < 128) badhi |= nib << (4 * i - 64);
            else badtop |= nib << (4 * i - 128);
        }
    }
    static_assert(G::NBYTES <= 192 && K <= 128 && P <= 64, "window too long for the 192-bit validity mask");
    (
        [&] {
            const Mmh3Open open = PosOps<K, P, O>::hash_open(U, C, seed);
            if constexpr (EARLY) {
                // s in {t, t - 1}: (s + 1) mod 2^32 <= thr_hi + 1 whenever t <= thr_hi
                if (!any_lane((uint32_t)(mmh3_close_hi(open) + 1u) <= lim)) return;
            }
            const uint64_t h = mmh3_close(open);
            bool ok = (h - 1) < thr;                          // h != 0 (signature.rs:50) and h <= thr (minhash.rs:319)
            if (anybad != 0) {
                // any invalid byte in [O, O+K) kills the k-mer (signature.rs:271-286, force=true)
                uint64_t lo, hi;                              // bits [O, O+128) of the mask (O < 64: P <= 64)
                if constexpr (O == 0) { lo = badlo; hi = badhi; }
                else { lo = (badlo >> O) | (badhi << (64 - O)); hi = (badhi >> O) | (badtop << (64 - O)); }
                const uint64_t mlo = K >= 64 ? ~0ull : ((1ull << K) - 1);
                const uint64_t mhi = K >= 128 ? ~0ull : K > 64 ? ((1ull << (K - 64)) - 1) : 0;
                if ((lo & mlo) | (hi & mhi)) ok = false;
            }
            if (ok) emit(O, h);
        }(),
        ...);
}

template <int K, int P, bool EARLY = true, class Emit>
SMG_HD void process_lane(const uint32_t* raw, uint64_t seed, uint64_t thr, Emit&& emit) {
    process_lane_impl<K, P, EARLY>(raw, seed, thr, static_cast<Emit&&>(emit), std::make_integer_sequence<int, P>{});
}

}  // namespace smg
