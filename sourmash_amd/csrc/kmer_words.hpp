// kmer_words.hpp -- one DNA k-mer of ANY length from a staged stretch of the sequence: canonical form -> MurmurHash3, 16 bytes at
// a time (round 5).  Host + device, like kmer_core.hpp: tests/native/kmer_words_emul.cpp runs it against the oracle on the CPU.
//
// Replaces, for k above the register-window kernel's 128 (kmer_core.hpp):
//   src/core/src/signature.rs:246-306  SeqToHashes::next, DNA branch: revcomp (:263), VALID scan (:271-286), canonical =
//                                      min(kmer, krc) as byte strings (:302-304), `_hash_murmur` (lib.rs:57-59)
// which treats every k alike.  Rounds 1-4 sent k = 129 .. 256 to a byte loop (two 256-byte arrays per lane in scratch memory:
// ~5 Gbase/s against 70 at k = 128) and refused k > 256.
//
// Formulation.  MurmurHash3 eats the key in 16-byte blocks, in order.  The stretch [base, base + L) of the sequence sits in LDS
// twice: upper-cased as it is, and as its REVERSE COMPLEMENT (byte j of the copy = complement of byte L - 1 - j).  The reverse
// complement of the k-mer at p is then a plain substring of the copy, at L - p - k: both strands are read the same way, ascending,
// 16 bytes at a time -- five aligned dword reads and four byte-aligns by the lane's own shift -- and a lane that hashes the
// reverse strand differs from its neighbour by an address.  (A first form read the reverse strand backwards from the one copy and
// complemented / reversed every block: 30 instructions of byte plumbing per block beside MurmurHash3's 45.)  The (k mod 16)-byte
// tail is one more 16-byte read, masked.  Which strand: big-endian compare of block 0 of both (both are loaded anyway; one of them
// is block 0 of the hash), later blocks only while some lane of the wave still ties (4^-16 per k-mer, or a palindrome).  Nothing is
// indexed by a run-time value except LDS addresses: no per-lane arrays, no scratch.
// Validity is a per-stretch prefix count of bytes outside ACGT (WwBad): a k-mer is good when its k bytes hold none.
#pragma once
#include "kmer_core.hpp"

namespace smg {

constexpr uint32_t WW_BACK = 32;     // bytes of slack behind either copy (a tail's 16-byte read starts inside the k-mer; five dwords are read)

struct WwBlock { uint32_t w[4]; };   // 16 key bytes as little-endian dwords

// where the two copies sit in the LDS array (byte offsets; L is a multiple of 16)
struct WwLayout {
    uint32_t L;          // staged bytes
    uint32_t rc_off;     // start of the reverse-complement copy: L + WW_BACK
    SMG_HD uint32_t fwd_at(uint32_t p) const { return p; }                          // k-mer p, forward strand
    SMG_HD uint32_t rev_at(uint32_t p, uint32_t k) const { return rc_off + L - p - k; }   // its reverse complement
    SMG_HD uint32_t dwords() const { return (rc_off + L + WW_BACK) / 4u; }
};
SMG_HD WwLayout ww_layout(uint32_t n_chunks) { return WwLayout{n_chunks * 16u, n_chunks * 16u + WW_BACK}; }

// the 16 bytes at byte index s of the LDS array
SMG_HD WwBlock ww_load16(const uint32_t* win, uint32_t s) {
    const uint32_t q = s >> 2, a = s & 3u;
    const uint32_t d0 = win[q], d1 = win[q + 1], d2 = win[q + 2], d3 = win[q + 3], d4 = win[q + 4];
    return WwBlock{{alignbyte_b32(d1, d0, a), alignbyte_b32(d2, d1, a), alignbyte_b32(d3, d2, a), alignbyte_b32(d4, d3, a)}};
}

// staging: the reverse complement of 4 upper-cased bases (bytes outside ACGT come out as some base: such k-mers are dropped)
SMG_HD uint32_t ww_revcomp4(uint32_t upper) {
    const uint32_t code = (upper >> 1) & 0x03030303u;
    return bswap32(perm_b32(0u, LUT_COMP, code));
}

// the first `t` (1 .. 15) key bytes of a block, the rest zero
SMG_HD WwBlock ww_first_bytes(WwBlock b, uint32_t t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t n = t > 4u * i ? t - 4u * i : 0u;     // bytes of dword i that belong
        b.w[i] = n >= 4u ? b.w[i] : n == 0u ? 0u : (b.w[i] & ((1u << (8u * n)) - 1u));
    }
    return b;
}

SMG_HD uint64_t ww_be_hi(const WwBlock& b) { return ((uint64_t)bswap32(b.w[0]) << 32) | bswap32(b.w[1]); }
SMG_HD uint64_t ww_be_lo(const WwBlock& b) { return ((uint64_t)bswap32(b.w[2]) << 32) | bswap32(b.w[3]); }

// Hash of the canonical k-mer that starts at stretch byte p (its k bytes are all in ACGT -- the caller checked); k >= 16.
SMG_HD uint64_t ww_hash(const uint32_t* win, WwLayout lay, uint32_t p, uint32_t k, uint64_t seed) {
    const uint32_t nb = k >> 4, t = k & 15u;                  // whole blocks, tail bytes
    const uint32_t f0 = lay.fwd_at(p), r0 = lay.rev_at(p, k);
    const WwBlock F = ww_load16(win, f0), R = ww_load16(win, r0);
    bool gt, tie;                                             // forward string > its reverse complement; equal so far
    {
        const uint64_t fh = ww_be_hi(F), rh = ww_be_hi(R), fl = ww_be_lo(F), rl = ww_be_lo(R);
        gt = fh > rh || (fh == rh && fl > rl);
        tie = fh == rh && fl == rl;
    }
    if (any_lane(tie)) {                                      // rare: look further, a block at a time
        for (uint32_t b = 1; b <= nb && any_lane(tie); ++b) {
            if (b == nb && t == 0u) break;
            WwBlock f = ww_load16(win, f0 + 16u * b), r = ww_load16(win, r0 + 16u * b);
            if (b == nb) { f = ww_first_bytes(f, t); r = ww_first_bytes(r, t); }
            const uint64_t fh = ww_be_hi(f), rh = ww_be_hi(r), fl = ww_be_lo(f), rl = ww_be_lo(r);
            gt = gt || (tie && (fh > rh || (fh == rh && fl > rl)));
            tie = tie && fh == rh && fl == rl;
        }
    }
    const bool rc = gt;                                       // hash the smaller string (signature.rs:302-304)
    const uint32_t s0 = rc ? r0 : f0;
    uint64_t h1 = seed, h2 = seed;
    {
        const uint32_t m = rc ? 0xffffffffu : 0u;
        mmh3_block(h1, h2, (uint64_t)bitselect(m, R.w[0], F.w[0]) | ((uint64_t)bitselect(m, R.w[1], F.w[1]) << 32),
                   (uint64_t)bitselect(m, R.w[2], F.w[2]) | ((uint64_t)bitselect(m, R.w[3], F.w[3]) << 32));
    }
    for (uint32_t b = 1; b < nb; ++b) {
        const WwBlock w = ww_load16(win, s0 + 16u * b);
        mmh3_block(h1, h2, (uint64_t)w.w[0] | ((uint64_t)w.w[1] << 32), (uint64_t)w.w[2] | ((uint64_t)w.w[3] << 32));
    }
    if (t) {                                                  // (wave-uniform: k is)
        const WwBlock w = ww_first_bytes(ww_load16(win, s0 + 16u * nb), t);
        if (t > 8u) {
            uint64_t k2 = (uint64_t)w.w[2] | ((uint64_t)w.w[3] << 32);
            k2 *= MMH3_C2; k2 = rotl64<33>(k2); k2 *= MMH3_C1; h2 ^= k2;
        }
        uint64_t k1 = (uint64_t)w.w[0] | ((uint64_t)w.w[1] << 32);
        k1 *= MMH3_C1; k1 = rotl64<31>(k1); k1 *= MMH3_C2; h1 ^= k1;
    }
    return mmh3_finish(h1, h2, (uint64_t)k);
}

// Bytes outside ACGT in a stretch: one bit per byte (bit j & 31 of bits[j >> 5]) and the count in front of every 32-byte word.
struct WwBad {
    const uint32_t* bits;
    const uint32_t* before;      // before[w] = bad bytes in words 0 .. w - 1
    SMG_HD uint32_t upto(uint32_t j) const {                  // bad bytes among stretch bytes 0 .. j - 1
        const uint32_t w = j >> 5, r = j & 31u;
        const uint32_t part = r ? bits[w] & ((1u << r) - 1u) : 0u;
#if defined(__HIP_DEVICE_COMPILE__)
        return before[w] + (uint32_t)__popc(part);
#else
        return before[w] + (uint32_t)__builtin_popcount(part);
#endif
    }
    SMG_HD bool clean(uint32_t p, uint32_t k) const { return upto(p + k) == upto(p); }
};

// upper-cased dword -> 4 bits, bit i set when byte i is not one of ACGT (encodings.rs:370-377; signature.rs:214 upper-cases)
SMG_HD uint32_t ww_bad4(uint32_t upper) {
    const uint32_t code = (upper >> 1) & 0x03030303u;
    return nonzero_bytes4(perm_b32(0u, LUT_SELF, code) ^ upper);
}

}  // namespace smg
