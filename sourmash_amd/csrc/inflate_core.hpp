// inflate_core.hpp -- DEFLATE (RFC 1951) decoding restated for a device where one wavefront decodes one run of blocks.
//
// Why: `sourmash sketch` reads genome.fna.gz (src/sourmash/command_sketch.py:697,746-768 through screed; the Rust bench inflates
// with niffler, src/core/benches/compute.rs:35-38).  Host inflate -- even on every core (pargz.hpp) -- fed the sketch kernel
// under 1 % of what it takes (2.1 Gbase/s against 313).  A deflate stream has no index, but every block of one
// can be decoded by itself if references into the 32 KB in front of it stay symbolic.  The scheme (gunzip.hip runs it):
//   scan      every bit position of the compressed bytes is tested for a dynamic-Huffman, non-final block header whose three
//             code-length sets are complete prefix codes (plausible_prefix: one 128-bit read per position; valid_dynamic_header:
//             the survivors, in full)                                                              -> candidate block starts
//   pass 1    one wavefront per candidate decodes from there WITHOUT output, through stored / fixed / final blocks, up to the
//             next dynamic non-final header                                         -> (end bit, bytes produced) per candidate
//   link      the chain of runs that begins at bit 0 and where each run ends on the next one's start (host: a few thousand
//             entries); candidates off the chain were false and are dropped; a gap in the chain refuses the member
//   pass 2    one wavefront per run of the chain decodes again, now writing 16-bit symbols at the run's final position:
//             a byte, or 0x8000 | m = "the byte at position (run start - 32768 + m)" for what is copied, directly or through
//             later copies, out of the window in front of the run
//   tails     the last 32 KB of every run resolved in stream order (each needs only the resolved tail in front of it)
//   resolve   every other symbol -> byte, all runs at once, each against its now final window
//   check     CRC-32 and length of the member against the gzip trailer
// Anything that does not add up (a gap in the chain, an invalid code, a checksum) refuses the member and the caller inflates
// it on the host (pargz.hpp / zlib): the result is the sequential result or an error, never something else.
//
// This header is the part shared by the kernels and the host: the bit reader, code tables, the block walk (decode_run) and
// the sink that turns symbols into wave-wide gathers and stores (WaveSink).  It compiles for the host as well, where a "lane"
// is a loop index: tests/native/inflate_emul.cpp runs the same code against zlib without a GPU.
#pragma once
#include <stdint.h>
#include <string.h>

#ifndef SMG_HD
#if defined(__HIPCC__)
#define SMG_HD __host__ __device__ __forceinline__
#else
#define SMG_HD inline
#endif
#endif

namespace smg {
namespace inf {

constexpr uint32_t WIN = 32768;
constexpr int LIT_ROOT = 10, DIST_ROOT = 8;            // direct-lookup bits of the two code tables; longer codes: canonical search
constexpr uint16_t MARK = 0x8000;                      // symbol = MARK | m: the byte at window offset m
constexpr uint32_t SYM_SLOW = 0x80, SYM_END = 0x40;    // flags beside a lane's symbol length (decode_run)
constexpr uint64_t MAX_RUN_BYTES = (1ull << 30) - 1;   // of one run (the sink keeps positions in 30 bits)
constexpr uint32_t MAX_BATCHES = 1u << 23;             // of one run: far beyond any block a compressor writes; bounds a false start

// status of a decoded run
enum : uint32_t { RUN_OK = 0, RUN_FINAL = 1, RUN_BAD_CODE = 2, RUN_BAD_BLOCK = 3, RUN_PAST_END = 4, RUN_BAD_DISTANCE = 5, RUN_TOO_LONG = 6 };

// A "lane" is a lane of the wavefront on the device and a loop index on the host (tests/native/inflate_emul.cpp): per-lane
// values are arrays of SMG_INF_LANES entries, of which a device lane owns the one.
#if defined(__HIP_DEVICE_COMPILE__)
#define SMG_INF_LANES 1
#define SMG_INF_EACH_LANE(lane, slot) const uint32_t lane = (uint32_t)(threadIdx.x & 63u); constexpr int slot = 0;
#else
#define SMG_INF_LANES 64
#define SMG_INF_EACH_LANE(lane, slot) for (uint32_t lane = 0, slot = 0; lane < 64u; ++lane, ++slot)
#endif
SMG_HD uint32_t lane_read(const uint32_t (&v)[SMG_INF_LANES], uint32_t lane) {      // lane: the same in every lane
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_readlane((int)v[0], (int)lane);
#else
    return v[lane];
#endif
}

// ---- bits: aligned 32-bit words, least significant bit first (RFC 1951 3.1.1) ----
// One lane by itself (the scan's full header test):
struct BitReader {
    const uint32_t* w;
    uint64_t pos, end;          // next unread bit, first bit behind the member's deflate data (both from the start of w)
    uint64_t buf;
    uint32_t cnt;
    uint64_t next;              // next word to load
    SMG_HD void init(const uint32_t* words, uint64_t bit, uint64_t end_bit) {
        w = words; pos = bit; end = end_bit;
        next = bit >> 5;
        const uint32_t s = (uint32_t)(bit & 31);
        buf = (uint64_t)(w[next++] >> s);
        cnt = 32 - s;
        refill();
    }
    SMG_HD void refill() {      // afterwards 33 .. 64 bits are in buf
        if (cnt <= 32) { buf |= (uint64_t)w[next++] << cnt; cnt += 32; }
    }
    SMG_HD uint32_t peek(uint32_t n) const { return (uint32_t)buf & ((1u << n) - 1u); }
    SMG_HD void drop(uint32_t n) { buf >>= n; cnt -= n; pos += n; }
    SMG_HD uint32_t take(uint32_t n) { const uint32_t v = peek(n); drop(n); return v; }
    SMG_HD void to_byte() { const uint32_t s = (uint32_t)((8 - (pos & 7)) & 7); drop(s); }
    SMG_HD bool past_end() const { return pos > end; }
};

// The whole wavefront on one stream (the block walk): the lanes hold the next 128 words of the stream between them (cur, nxt:
// one coalesced load per 64 words, issued 64 words before its first use), a word reaches the scalar side by a lane read, and
// the next 128 bits sit in (lo, hi), where every lane can look at "the bits from my lane number on" (lane_bits).
struct WaveBits {
    const uint32_t* w;
    uint64_t end;               // first bit behind the member's deflate data
    uint64_t base_word;         // the word lane 0 of cur holds
    uint32_t r0[SMG_INF_LANES], r1[SMG_INF_LANES];   // two chunks of 64 words taking turns: one is read, the other on its way
    uint32_t turn;              // 0: r0 is read (words base_word ..), r1 holds the 64 behind it; 1: the other way round
    uint64_t lo, hi;
    uint32_t cnt;               // valid bits in (lo, hi)
    uint32_t wi;                // next word to append, as a lane of the chunk being read
    SMG_HD void init(const uint32_t* words, uint64_t bit, uint64_t end_bit) {
        w = words; end = end_bit;
        base_word = bit >> 5;
        { SMG_INF_EACH_LANE(lane, slot) { r0[slot] = w[base_word + lane]; r1[slot] = w[base_word + 64u + lane]; } }
        lo = hi = 0; cnt = 0; wi = 0; turn = 0;
        refill();
        drop((uint32_t)(bit & 31u));
        refill();
    }
    SMG_HD void refill() {      // afterwards 97 .. 128 bits are valid
        while (cnt <= 96u) {
            if (wi == 64u) {    // the chunk just read is asked to fetch the 64 words behind the other one, and they trade places
                base_word += 64u;
                wi = 0;
                if (turn == 0u) { SMG_INF_EACH_LANE(lane, slot) { r0[slot] = w[base_word + 64u + lane]; } }
                else { SMG_INF_EACH_LANE(lane, slot) { r1[slot] = w[base_word + 64u + lane]; } }
                turn ^= 1u;
            }
            uint64_t x;
            if (turn == 0u) x = lane_read(r0, wi); else x = lane_read(r1, wi);
            ++wi;
            if (cnt < 64u) {
                lo |= x << cnt;
                if (cnt > 32u) hi |= x >> (64u - cnt);
            } else hi |= x << (cnt - 64u);
            cnt += 32u;
        }
    }
    SMG_HD uint64_t pos() const { return (base_word + wi) * 32u - cnt; }
    SMG_HD uint32_t peek(uint32_t n) const { return (uint32_t)lo & ((1u << n) - 1u); }
    SMG_HD void drop(uint32_t n) {                                    // n <= 127
        if (n >= 64u) { lo = hi >> (n - 64u); hi = 0; }
        else if (n) { lo = (lo >> n) | (hi << (64u - n)); hi >>= n; }
        cnt -= n;
    }
    SMG_HD uint32_t take(uint32_t n) { const uint32_t v = peek(n); drop(n); return v; }
    SMG_HD void to_byte() { drop((uint32_t)((8u - (pos() & 7u)) & 7u)); }
    SMG_HD bool past_end() const { return pos() > end; }
    // n <= 16 bits from bit `at` of the window (at + n <= cnt)
    SMG_HD uint32_t bits_at(uint32_t at, uint32_t n) const {
        const uint64_t v = at < 64u ? ((lo >> at) | (at ? hi << (64u - at) : 0ull)) : hi >> (at - 64u);
        return (uint32_t)v & ((1u << n) - 1u);
    }
    // the 32 bits that begin at this lane's number
    SMG_HD uint32_t lane_bits(uint32_t lane) const { return (uint32_t)((lo >> lane) | (lane ? hi << (64u - lane) : 0ull)); }
};

SMG_HD uint32_t ctz64(uint64_t m) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_ctzll(m);
#else
    uint32_t n = 0;
    while (!((m >> n) & 1u)) ++n;
    return n;
#endif
}

SMG_HD uint32_t bitrev15(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(v) >> 17;
#else
    uint32_t r = 0;
    for (int i = 0; i < 15; ++i) r |= ((v >> i) & 1u) << (14 - i);
    return r;
#endif
}

// ---- one prefix code: direct table for codes of at most `root` bits, canonical search beyond ----
struct CodeStore {             // per code length: how many symbols, the first canonical code, where its symbols begin in `sorted`
    uint16_t count[16], first[16], offset[16], fill[16];
};
// A table entry: bits 0-3 the code's length (0: no code this short), 4-5 the kind (0 literal / plain symbol, 1 length or
// distance, 2 end of block, 3 not a symbol of the format), 8-11 the number of extra bits, 16-31 the value (the literal, the
// symbol, or the base of the length / distance) -- everything the block walk needs in one lookup.
constexpr int CODE_PLAIN = 0, CODE_LITLEN = 1, CODE_DIST = 2;
SMG_HD uint32_t len_base(uint32_t s);
SMG_HD uint32_t len_extra(uint32_t s);
SMG_HD uint32_t dist_base(uint32_t s);
SMG_HD uint32_t dist_extra(uint32_t s);
SMG_HD uint32_t symbol_entry(int what, uint32_t s) {
    if (what == CODE_LITLEN) {
        if (s < 256u) return s << 16;
        if (s == 256u) return 2u << 4;
        if (s > 285u) return 3u << 4;
        return (len_base(s - 257u) << 16) | (len_extra(s - 257u) << 8) | (1u << 4);
    }
    if (what == CODE_DIST) {
        if (s > 29u) return 3u << 4;
        return (dist_base(s) << 16) | (dist_extra(s) << 8) | (1u << 4);
    }
    return s << 16;
}
struct Code {
    uint32_t* table;            // 1 << root entries
    uint16_t* sorted;           // symbols ordered by (length, symbol)
    CodeStore* st;              // (everything a decoder indexes at run time lives in the caller's scratch: LDS on the device)
};

// lens[0, n) -> code.  false: over-subscribed, or incomplete with more than one code / a code longer than one bit (the sets
// zlib's inflate_table refuses).
SMG_HD bool build_code(const uint8_t* lens, int n, int root, Code& c, int what) {
    CodeStore& t = *c.st;
    for (int i = 0; i < 16; ++i) t.count[i] = 0;
    for (int i = 0; i < n; ++i) t.count[lens[i]]++;
    t.count[0] = 0;
    int left = 1, maxlen = 0;
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - (int)t.count[l];
        if (left < 0) return false;
        if (t.count[l]) maxlen = l;
    }
    if (left > 0 && maxlen > 1) return false;
    uint32_t code = 0, off = 0;
    for (int l = 1; l <= 15; ++l) {
        code = (code + (l > 1 ? t.count[l - 1] : 0u)) << 1;
        t.first[l] = (uint16_t)code;
        t.offset[l] = (uint16_t)off;
        off += t.count[l];
        t.fill[l] = 0;
    }
    t.first[0] = t.offset[0] = t.fill[0] = 0;
    for (int i = 0; i < (1 << root); ++i) c.table[i] = 0;
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t rank = t.fill[l]++;
        c.sorted[t.offset[l] + rank] = (uint16_t)s;
        if (l <= root) {
            const uint32_t cw = t.first[l] + rank;                     // canonical code, first bit = most significant
            uint32_t r = 0;
            for (int b = 0; b < l; ++b) r |= ((cw >> b) & 1u) << (l - 1 - b);
            const uint32_t e = symbol_entry(what, (uint32_t)s) | (uint32_t)l;
            for (uint32_t i = r; i < (1u << root); i += 1u << l) c.table[i] = e;
        }
    }
    return true;
}

// The code whose first bits are v (15 of them), for codes longer than the table's root: -> its entry with the length in bits
// 0-3, or 0 when there is none.
SMG_HD uint32_t long_code(uint32_t v, const Code& c, int root, int what) {
    const uint32_t r = bitrev15(v & 0x7fffu);
    const CodeStore& t = *c.st;
    for (int l = root + 1; l <= 15; ++l) {
        const uint32_t d = (r >> (15 - l)) - t.first[l];
        if (d < t.count[l]) return symbol_entry(what, (uint32_t)c.sorted[t.offset[l] + d]) | (uint32_t)l;
    }
    return 0;
}

// length / distance symbol -> base value and extra bits (RFC 1951 3.2.5)
SMG_HD uint32_t len_base(uint32_t s) {    // s = symbol - 257, 0 .. 28
    return s < 8 ? 3 + s : s == 28 ? 258 : 3 + ((4 + (s & 3)) << ((s >> 2) - 1));
}
SMG_HD uint32_t len_extra(uint32_t s) { return s < 8 || s == 28 ? 0 : (s >> 2) - 1; }
SMG_HD uint32_t dist_base(uint32_t s) {   // 0 .. 29
    return s < 4 ? 1 + s : 1 + ((2 + (s & 1)) << ((s >> 1) - 1));
}
SMG_HD uint32_t dist_extra(uint32_t s) { return s < 4 ? 0 : (s >> 1) - 1; }

// scratch of one decoder: the tables of the two codes and the code lengths of a dynamic block (LDS on the device)
struct HeaderScratch {         // what reading a dynamic block's header needs (the scan's full test: one lane each)
    uint32_t cl_table[128];
    uint16_t cl_sorted[20];
    CodeStore cl_store;
    uint8_t lens[288 + 32 + 32];                                      // literal/length + distance code lengths; [320, 339): the code-length code's
};
struct Scratch : HeaderScratch {
    uint32_t lit_table[1 << LIT_ROOT], dist_table[1 << DIST_ROOT];
    uint16_t lit_sorted[288], dist_sorted[32];
    CodeStore lit_store, dist_store;
};

// Code lengths of a dynamic block (RFC 1951 3.2.7): the reader stands behind the 3 header bits.  strict: the conditions a
// SCANNED block start must meet on top of being decodable (end-of-block code present, the distance code complete, a single
// code, or absent).  -> false: not a dynamic block header.
template <class Bits>
SMG_HD bool read_dynamic_lengths(Bits& br, HeaderScratch& S, int& hlit, int& hdist, bool strict) {
    br.refill();
    hlit = (int)br.take(5) + 257;
    hdist = (int)br.take(5) + 1;
    const int hclen = (int)br.take(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    // entry i of the header is the length of symbol order[i] (RFC 1951 3.2.7): 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
    uint8_t* cl = S.lens + 320;
    for (int i = 0; i < 19; ++i) cl[i] = 0;
    for (int i = 0; i < hclen; ++i) {
        br.refill();
        // order[i] as arithmetic (no table in memory): i = 0, 1, 2 -> 16, 17, 18; 3 -> 0; then 8, 7, 9, 6, 10, 5, ... alternate around 8
        const int sym = i < 3 ? 16 + i : i == 3 ? 0 : (i & 1) ? 8 - ((i - 3) >> 1) : 8 + ((i - 4) >> 1);
        cl[sym] = (uint8_t)br.take(3);
    }
    Code cc;
    cc.table = S.cl_table;
    cc.sorted = S.cl_sorted;
    cc.st = &S.cl_store;
    if (!build_code(cl, 19, 7, cc, CODE_PLAIN)) return false;
    if (strict) {                                                     // the code-length code itself: complete
        int left = 1;
        for (int l = 1; l <= 7; ++l) left = (left << 1) - (int)S.cl_store.count[l];
        if (left != 0) return false;
    }
    int got = 0, prev = 0;
    const int total = hlit + hdist;
    while (got < total) {
        br.refill();
        if (br.past_end()) return false;
        const uint32_t e = cc.table[br.peek(7)];
        if (!(e & 15u)) return false;
        br.drop(e & 15u);
        const int sym = (int)(e >> 16);
        if (sym < 16) { S.lens[got++] = (uint8_t)sym; prev = sym; continue; }
        int rep, val = 0;
        if (sym == 16) { if (got == 0) return false; rep = 3 + (int)br.take(2); val = prev; }
        else if (sym == 17) rep = 3 + (int)br.take(3);
        else rep = 11 + (int)br.take(7);
        if (got + rep > total) return false;
        for (int i = 0; i < rep; ++i) S.lens[got++] = (uint8_t)val;
        prev = val;
    }
    if (S.lens[256] == 0) return false;                               // no end-of-block code (zlib: "invalid code -- missing end-of-block")
    return true;
}

// Is a dynamic-Huffman, non-final block header at `bit`?  The cheap part: header bits, and the Kraft sum of the code-length
// code on two 64-bit reads (7 of 8 positions fail the first test, almost all others the second).  lo / hi: bits [bit, bit+128).
SMG_HD bool plausible_prefix(uint64_t lo, uint64_t hi) {
    if ((lo & 7u) != 4u) return false;                                // BFINAL = 0, BTYPE = 10b
    const uint32_t hlit = (uint32_t)(lo >> 3) & 31u, hdist = (uint32_t)(lo >> 8) & 31u, hclen = ((uint32_t)(lo >> 13) & 15u) + 4u;
    if (hlit > 29u || hdist > 29u) return false;
    uint32_t kraft = 0;
    uint64_t v = (lo >> 17) | (hi << 47);                             // 3 bits per entry from bit 17 on: 57 bits at most
    for (uint32_t i = 0; i < hclen; ++i, v >>= 3) {
        const uint32_t l = (uint32_t)v & 7u;
        if (l) kraft += 128u >> l;
    }
    return kraft == 128u;
}

// The full test of a candidate (the scan's second kernel; one lane each): code lengths readable, the literal/length code
// complete with an end-of-block code, the distance code complete -- or one code, or none.  Written to live in registers: the
// code-length code's 19 lengths are a packed word, the two big codes are judged by their Kraft sums as their lengths go by,
// and the only table is the 128-entry decoder of the code-length code, reached through `tab` (Tab: get(i) / set(i, v) on 128
// bytes -- a plain array on the host, a lane's column of an LDS tile on the device).
struct PlainTab {
    uint8_t t[128];
    SMG_HD uint32_t get(uint32_t i) const { return t[i]; }
    SMG_HD void set(uint32_t i, uint32_t v) { t[i] = (uint8_t)v; }
};
template <class Tab>
SMG_HD bool valid_dynamic_header(const uint32_t* words, uint64_t bit, uint64_t end_bit, Tab& tab) {
    BitReader br;
    br.init(words, bit, end_bit);
    if (br.take(3) != 4u) return false;
    br.refill();
    const uint32_t hlit = br.take(5) + 257u, hdist = br.take(5) + 1u, hclen = br.take(4) + 4u;
    if (hlit > 286u || hdist > 30u) return false;
    uint64_t cl = 0;                                                  // length of code-length symbol s at bits [3 s, 3 s + 3)
    for (uint32_t i = 0; i < hclen; ++i) {
        br.refill();
        const uint32_t sym = i < 3u ? 16u + i : i == 3u ? 0u : (i & 1u) ? 8u - ((i - 3u) >> 1) : 8u + ((i - 4u) >> 1);
        cl |= (uint64_t)br.take(3) << (3u * sym);
    }
    uint64_t count = 0;                                               // symbols per length, 8 bits each
    uint32_t kraft = 0;
    for (uint32_t sym = 0; sym < 19u; ++sym) {
        const uint32_t l = (uint32_t)(cl >> (3u * sym)) & 7u;
        if (l) { count += 1ull << (8u * l); kraft += 128u >> l; }
    }
    if (kraft != 128u) return false;
    uint64_t next = 0;                                                // first code of every length, 8 bits each
    for (uint32_t l = 1, code = 0; l <= 7u; ++l) {
        code = (code + (l > 1u ? (uint32_t)(count >> (8u * (l - 1u))) & 0xffu : 0u)) << 1;
        next |= (uint64_t)(code & 0xffu) << (8u * l);
    }
    for (uint32_t sym = 0; sym < 19u; ++sym) {
        const uint32_t l = (uint32_t)(cl >> (3u * sym)) & 7u;
        if (!l) continue;
        const uint32_t code = (uint32_t)(next >> (8u * l)) & 0xffu;
        next += 1ull << (8u * l);
        uint32_t r = 0;
        for (uint32_t k = 0; k < l; ++k) r |= ((code >> k) & 1u) << (l - 1u - k);
        for (uint32_t i = r; i < 128u; i += 1u << l) tab.set(i, (sym << 3) | l);
    }
    const uint32_t total = hlit + hdist;
    uint32_t got = 0, prev = 0, k_lit = 0, k_dist = 0, used_dist = 0;
    bool eob = false;
    while (got < total) {
        br.refill();
        if (br.past_end()) return false;
        const uint32_t e = tab.get(br.peek(7));                      // (the code is complete: every index holds a symbol)
        br.drop(e & 7u);
        const uint32_t sym = e >> 3;
        uint32_t rep = 1, val = sym;
        if (sym == 16u) { if (got == 0u) return false; rep = 3u + br.take(2); val = prev; }
        else if (sym == 17u) { rep = 3u + br.take(3); val = 0; }
        else if (sym == 18u) { rep = 11u + br.take(7); val = 0; }
        if (got + rep > total) return false;
        if (val) {
            const uint32_t w = 32768u >> val;
            // the repeat may straddle the two codes
            const uint32_t in_lit = got >= hlit ? 0u : (hlit - got < rep ? hlit - got : rep);
            k_lit += w * in_lit;
            k_dist += w * (rep - in_lit);
            used_dist += rep - in_lit;
            if (got <= 256u && got + rep > 256u) eob = true;
        }
        got += rep;
        prev = val;
    }
    if (br.past_end() || !eob) return false;                          // (zlib: "invalid code -- missing end-of-block")
    if (k_lit != 32768u) return false;
    if (k_dist != 32768u && !(used_dist == 0u || (used_dist == 1u && k_dist == 16384u))) return false;
    return true;
}

// ---- pass 1's sink: the symbols of a run as 32-bit records, and the number of bytes they make ----
// A record: 0x80000000 | byte (a literal), or (distance - 1) << 9 | length (a match), or -- a stored block -- the pair
// 0x40000000 | length, byte offset of its bytes in the buffer.  A symbol takes at least one bit of the stream and a stored
// block at least 32, so the records of the run that begins at bit b fit into [b, end bit of the run) of a record buffer as long
// as the stream is in bits: no run needs to know how many symbols the runs in front of it hold.  (A FALSE block start inside
// a real run may write into that run's stretch; gunzip.hpp refuses the member if a candidate's first bit lies inside the
// records of a run of the chain.)
constexpr uint32_t REC_LITERAL = 0x80000000u, REC_STORED = 0x40000000u;

// set bits of m below bit `lane` (on the device: below the calling lane -- v_mbcnt, no lane-mask constant to keep or to spill)
SMG_HD uint32_t popc_below(uint64_t m, uint32_t lane) {
#if defined(__HIP_DEVICE_COMPILE__)
    (void)lane;
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
#else
    uint32_t n = 0;
    for (uint64_t b = m & ((1ull << lane) - 1ull); b; b &= b - 1) ++n;
    return n;
#endif
}

SMG_HD uint32_t popc64(uint64_t m) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popcll(m);
#else
    uint32_t n = 0;
    for (; m; m &= m - 1) ++n;
    return n;
#endif
}

struct RecordSink {
    uint32_t* rec = nullptr;    // the run's stretch of the record buffer (nullptr: count only)
    uint64_t cap = 0;           // records it may hold
    uint32_t n = 0;             // records written
    uint64_t nb = 0;            // bytes made by what the uniform path recorded
    uint32_t acc[SMG_INF_LANES];// bytes made by each lane's symbols of the batches: two vector instructions a batch, one sum a run
    bool bad = false;
    SMG_HD RecordSink() { SMG_INF_EACH_LANE(lane, slot) { (void)lane; acc[slot] = 0; } }
    SMG_HD void put(uint32_t r) {
        if (n >= cap) { bad = true; return; }
        if (rec) { SMG_INF_EACH_LANE(lane, slot) { (void)slot; if (lane == 0u) rec[n] = r; } }
        ++n;
    }
    SMG_HD void literal(uint32_t b) { put(REC_LITERAL | b); ++nb; }
    SMG_HD void match(uint32_t len, uint32_t dist) { put(((dist - 1u) << 9) | len); nb += len; }
    SMG_HD void stored(uint64_t byte_off, uint32_t len) { put(REC_STORED | len); put((uint32_t)byte_off); nb += len; }
    // the real symbols of a batch: bit l of mask <-> the symbol that begins at bit offset l; L: 0x80000000 | literal, or the
    // length of a match; D: its distance.  Lane l's record goes to slot n + (real symbols in front of l).
    SMG_HD void batch(uint64_t mask, const uint32_t (&L)[SMG_INF_LANES], const uint32_t (&D)[SMG_INF_LANES]) {
        const uint32_t k = popc64(mask);
        if ((uint64_t)n + k > cap) { bad = true; return; }
        SMG_INF_EACH_LANE(lane, slot) {
            if ((mask >> lane) & 1u) {
                const uint32_t x = L[slot];
                acc[slot] += x & 0x80000000u ? 1u : x;
                if (rec) rec[n + popc_below(mask, lane)] = x & 0x80000000u ? x : (((D[slot] - 1u) << 9) | x);
            }
        }
        n += k;
    }
    SMG_HD void finish() {}
    SMG_HD uint64_t total() const {
        uint64_t t = nb;
#if defined(__HIP_DEVICE_COMPILE__)
        uint32_t a = acc[0];                                          // (a run is cut off long before 2^32 bytes: decode_run's batch limit)
        for (int o = 32; o; o >>= 1) a += __shfl_xor(a, o);
        t += a;
#else
        for (int i = 0; i < SMG_INF_LANES; ++i) t += acc[i];
#endif
        return t;
    }
};

// ---- pass 2: records -> 16-bit symbols at their final place, 64 output positions at a time ----
// No code is decoded any more: 64 records are read side by side, a prefix sum of their lengths places them, and every output
// position finds its record by a binary search over those places (wave-private LDS).  A match whose source lies in the bytes
// of the same round -- distance < bytes in front of its end, and it is not the round's first record -- ends the round in
// front of it: everything it reads is then behind stores already issued (same wavefront, program order).  The gather of a
// tile is ISSUED and its store waits for the next tile (the trip to memory hides behind the next tile's search).
struct ExpandScratch { uint32_t pos[64], rec[64]; };

struct Expander {
    uint16_t* out;              // the run's first symbol
    uint64_t cap;               // symbols the run may write (pass 1's count)
    const uint8_t* bytes;       // the buffer of the files (stored blocks)
    bool no_window;             // the member's first run: nothing in front of it
    bool bad = false;
    uint32_t g0 = 0;            // symbols written (or under way)
    uint32_t pend_g0 = 0, pend_n = 0;
    uint16_t pend[SMG_INF_LANES];

    SMG_HD void commit() {
        if (!pend_n) return;
        { SMG_INF_EACH_LANE(lane, slot) { if (lane < pend_n) out[pend_g0 + lane] = pend[slot]; } }
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the stores above are in front of every later load of this wavefront
#endif
        pend_n = 0;
    }
    // a tile: lane j <-> output position g0 + j, src as WaveSink's: 0x40000000 | symbol, or a position relative to the run start
    SMG_HD void tile(const int32_t (&src)[SMG_INF_LANES], uint32_t n) {
        if ((uint64_t)g0 + n > cap) { bad = true; return; }
        commit();
        bool wrong = false;
        { SMG_INF_EACH_LANE(lane, slot) {
            if (lane < n) {
                const int32_t s = src[slot];
                uint16_t sym;
                if ((s & 0x40000000) && s >= 0) sym = (uint16_t)(s & 0xffff);
                else if (s >= 0) sym = out[s];
                else if (s < -(int32_t)WIN || no_window) { sym = 0; wrong = true; }
                else sym = (uint16_t)(MARK | (uint32_t)(s + (int32_t)WIN));
                pend[slot] = sym;
            }
        } }
#if defined(__HIP_DEVICE_COMPILE__)
        wrong = __builtin_amdgcn_ballot_w64(wrong) != 0ull;
#endif
        if (wrong) bad = true;
        pend_g0 = g0;
        pend_n = n;
        g0 += n;
    }

    SMG_HD void run(const uint32_t* rec, uint32_t n_rec, ExpandScratch& X) {
        uint32_t r0 = 0;
        while (r0 < n_rec && !bad) {
            // 64 records side by side
            uint32_t R[SMG_INF_LANES], LEN[SMG_INF_LANES], POS[SMG_INF_LANES];
            { SMG_INF_EACH_LANE(lane, slot) {
                const uint32_t i = r0 + lane;
                const uint32_t r = i < n_rec ? rec[i] : REC_STORED;     // (behind the end: stops the round like a stored block does)
                R[slot] = r;
                LEN[slot] = r & REC_LITERAL ? 1u : r & REC_STORED ? 0u : r & 0x1ffu;
            } }
            uint32_t total = 0;
#if defined(__HIP_DEVICE_COMPILE__)
            {
                uint32_t v = LEN[0];
                const uint32_t lane = threadIdx.x & 63u;
                for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(v, o); if (lane >= (uint32_t)o) v += u; }
                POS[0] = v - LEN[0];
                total = __shfl(v, 63);
            }
#else
            for (uint32_t l = 0; l < 64u; ++l) { POS[l] = total; total += LEN[l]; }
#endif
            // the round ends in front of the first record that cannot go with the ones before it
            uint64_t stop = 0;
            { SMG_INF_EACH_LANE(lane, slot) {
                const uint32_t r = R[slot];
                bool s = (r & (REC_LITERAL | REC_STORED)) == REC_STORED;
                if (!(r & (REC_LITERAL | REC_STORED))) s = POS[slot] >= 1u && (r >> 9) + 1u < POS[slot] + LEN[slot];
#if defined(__HIP_DEVICE_COMPILE__)
                (void)lane;
                stop = __builtin_amdgcn_ballot_w64(s);
#else
                if (s) stop |= 1ull << lane;
#endif
            } }
            const uint32_t h = stop ? ctz64(stop) : 64u;
            if (h == 0u) {                                            // a stored block (or nothing left): by itself
                const uint32_t r = lane_read(R, 0);
                if (r0 + 1u >= n_rec) { bad = true; break; }
                const uint32_t len = r & 0xffffu;
                const uint64_t off = lane_read(R, 1);
                for (uint32_t k = 0; k < len; k += 64u) {
                    int32_t src[SMG_INF_LANES];
                    const uint32_t m = len - k < 64u ? len - k : 64u;
                    { SMG_INF_EACH_LANE(lane, slot) { src[slot] = lane < m ? (int32_t)(0x40000000u | bytes[off + k + lane]) : 0; } }
                    tile(src, m);
                }
                r0 += 2u;
                continue;
            }
            const uint32_t T = h < 64u ? lane_read(POS, h) : total;
            { SMG_INF_EACH_LANE(lane, slot) { X.pos[lane] = lane < h ? POS[slot] : 0xffffffffu; X.rec[lane] = R[slot]; } }
            const uint32_t base = g0;
            for (uint32_t t0 = 0; t0 < T; t0 += 64u) {
                int32_t src[SMG_INF_LANES];
                const uint32_t m = T - t0 < 64u ? T - t0 : 64u;
                { SMG_INF_EACH_LANE(lane, slot) {
                    const uint32_t j = t0 + lane;
                    uint32_t lo = 0;                                  // the last record whose place is <= j
                    for (uint32_t step = 32u; step; step >>= 1) if (X.pos[lo + step] <= j) lo += step;
                    const uint32_t r = X.rec[lo];
                    int32_t sv;
                    if (r & REC_LITERAL) sv = (int32_t)(0x40000000u | (r & 0xffu));
                    else {
                        const uint32_t len = r & 0x1ffu, dist = (r >> 9) + 1u;
                        uint32_t k = j - X.pos[lo];
                        if (dist < len) k %= dist;
                        const int64_t p = (int64_t)base + X.pos[lo] - (int64_t)dist + k;
                        sv = (int32_t)(p < -(int64_t)WIN ? -(int32_t)WIN - 1 : p);
                    }
                    src[slot] = lane < m ? sv : 0;
                } }
                tile(src, m);
            }
            r0 += h;
        }
        commit();
    }
};

struct RunResult {
    uint64_t end_bit;           // where the run stopped: the next dynamic non-final header, or behind the final block
    uint64_t out_len;
    uint32_t n_records;
    uint32_t status;            // RUN_OK: stopped in front of a block header; RUN_FINAL: the stream's last block is done; else an error
};

// Decode from `bit` (a block header) through stored / fixed / final blocks until the next dynamic non-final block header
// (the next run's start) or the end of the final block.  max_out bounds a false candidate's run.
//
// The symbols of a block, a batch at a time: with >= 97 bits in the window every lane decodes the WHOLE symbol that would begin
// at the bit offset of its lane number -- literal / length entry, its extra bits, the distance entry behind them, its extra
// bits: two LDS lookups and ~40 vector instructions for 64 offsets at once -- and the uniform walk hops from real symbol to
// real symbol by lane reads (bits taken, bytes made, distance).  The chip runs one scalar instruction per CU and cycle for
// ALL of a CU's wavefronts, and every CU holds ~24 of these: what bounds the walk is its instruction count per symbol, not
// its latency.  A lane that meets a code longer than the table's root flags itself, and that symbol goes through the
// canonical search.
template <class Sink>
SMG_HD RunResult decode_run(const uint32_t* words, uint64_t bit, uint64_t end_bit, Scratch& S, Sink& sink, uint64_t max_out) {
    RunResult r;
    r.out_len = 0;
    r.status = RUN_OK;
    WaveBits br;
    br.init(words, bit, end_bit);
    Code lit, dist;
    lit.table = S.lit_table; lit.sorted = S.lit_sorted; lit.st = &S.lit_store;
    dist.table = S.dist_table; dist.sorted = S.dist_sorted; dist.st = &S.dist_store;
    uint32_t batches = 0;
    bool first = true;
    for (;;) {
        br.refill();
        if (br.pos() + 3 > br.end) { r.status = RUN_PAST_END; break; }
        const uint32_t hdr = br.peek(3);
        if (!first && hdr == 4u) break;                               // the next run begins here
        first = false;
        br.drop(3);
        const bool final_block = hdr & 1u;
        const uint32_t type = hdr >> 1;
        if (type == 0) {                                              // stored
            br.to_byte();
            br.refill();
            const uint32_t len = br.take(16);
            const uint32_t nlen = br.take(16);
            if ((len ^ nlen) != 0xffffu) { r.status = RUN_BAD_BLOCK; break; }
            const uint64_t at = br.pos();
            if (at + (uint64_t)len * 8 > br.end) { r.status = RUN_PAST_END; break; }
            sink.stored(at >> 3, len);
            br.init(words, at + (uint64_t)len * 8, end_bit);
        } else if (type == 3) {
            r.status = RUN_BAD_BLOCK;
            break;
        } else {
            bool ok;
            if (type == 1) {                                          // fixed codes (RFC 1951 3.2.6)
                for (int i = 0; i < 144; ++i) S.lens[i] = 8;
                for (int i = 144; i < 256; ++i) S.lens[i] = 9;
                for (int i = 256; i < 280; ++i) S.lens[i] = 7;
                for (int i = 280; i < 288; ++i) S.lens[i] = 8;
                for (int i = 0; i < 32; ++i) S.lens[288 + i] = 5;    // (codes 30 and 31 exist in the code and are refused as symbols)
                ok = build_code(S.lens, 288, LIT_ROOT, lit, CODE_LITLEN) && build_code(S.lens + 288, 32, DIST_ROOT, dist, CODE_DIST);
            } else {
                int hlit, hdist;
                ok = read_dynamic_lengths(br, S, hlit, hdist, false);
                ok = ok && build_code(S.lens, hlit, LIT_ROOT, lit, CODE_LITLEN) && build_code(S.lens + hlit, hdist, DIST_ROOT, dist, CODE_DIST);
            }
            if (!ok) { r.status = RUN_BAD_BLOCK; break; }
            bool end_of_block = false;
            while (!end_of_block) {
                br.refill();
                // every lane: the whole symbol that would begin at its bit offset -- bits it takes, bytes it makes, distance
                uint32_t A[SMG_INF_LANES], L[SMG_INF_LANES], D[SMG_INF_LANES];
                { SMG_INF_EACH_LANE(lane, slot) {
                    const uint64_t v64 = lane ? (br.lo >> lane) | (br.hi << (64u - lane)) : br.lo;
                    const uint32_t v = (uint32_t)v64;
                    const uint32_t e = lit.table[v & ((1u << LIT_ROOT) - 1u)];
                    const uint32_t cl = e & 15u, kind = (e >> 4) & 3u, eb = (e >> 8) & 15u;
                    const uint32_t o2 = cl + eb;                     // <= 20
                    const uint32_t len = (e >> 16) + ((v >> cl) & ((1u << eb) - 1u));
                    const uint32_t v2 = (uint32_t)(v64 >> o2);
                    const uint32_t d = dist.table[v2 & ((1u << DIST_ROOT) - 1u)];
                    const uint32_t dl = d & 15u, db = (d >> 8) & 15u;
                    uint32_t a;
                    if (cl == 0u || kind == 3u) a = SYM_SLOW;        // a long code, or no symbol: the uniform path looks (and reports)
                    else if (kind == 0u) a = cl;
                    else if (kind == 2u) a = cl | SYM_END;
                    else if (dl == 0u || ((d >> 4) & 3u) == 3u) a = SYM_SLOW;
                    else a = o2 + dl + db;                           // <= 20 + 28
                    A[slot] = a;
                    L[slot] = kind == 0u ? (0x80000000u | (e >> 16)) : len;
                    D[slot] = (d >> 16) + ((v2 >> dl) & ((1u << db) - 1u));
                } }
                // the uniform walk: from symbol to symbol by lane reads, as far as whole symbols lie inside the window; a lane
                // that flagged itself ends it
                const uint32_t o_max = br.cnt - 48u < 63u ? br.cnt - 48u : 63u;
                uint32_t o = 0, a = 0;
                uint64_t mask = 0;
                while (o <= o_max) {
                    a = lane_read(A, o);
                    if (a & (SYM_SLOW | SYM_END)) break;
                    mask |= 1ull << o;
                    o += a;
                }
                if (mask) sink.batch(mask, L, D);
                if (o <= o_max) {
                    if (a & SYM_END) { o += a & 0x3fu; end_of_block = true; }
                    else {                                            // this one symbol through the canonical search
                        uint32_t e = lit.table[br.bits_at(o, LIT_ROOT)];
                        if (!(e & 15u)) e = long_code(br.bits_at(o, 15), lit, LIT_ROOT, CODE_LITLEN);
                        const uint32_t kind = (e >> 4) & 3u;
                        if (!e || kind == 3u) r.status = RUN_BAD_CODE;
                        else if (kind == 0u) { sink.literal(e >> 16); o += e & 15u; }
                        else if (kind == 2u) { o += e & 15u; end_of_block = true; }
                        else {
                            const uint32_t eb = (e >> 8) & 15u;
                            const uint32_t o2 = o + (e & 15u) + eb;  // <= 63 + 20: bits_at reaches that far (cnt >= o_max + 48)
                            const uint32_t len = (e >> 16) + br.bits_at(o + (e & 15u), eb);
                            uint32_t d = dist.table[br.bits_at(o2, DIST_ROOT)];
                            if (!(d & 15u)) d = long_code(br.bits_at(o2, 15), dist, DIST_ROOT, CODE_DIST);
                            if (!d || ((d >> 4) & 3u) == 3u) r.status = RUN_BAD_CODE;
                            else {
                                const uint32_t db = (d >> 8) & 15u;
                                const uint32_t dd = (d >> 16) + br.bits_at(o2 + (d & 15u), db);
                                o = o2 + (d & 15u) + db;
                                sink.match(len, dd);
                            }
                        }
                    }
                }
                br.drop(o);
                if (r.status != RUN_OK) break;
                if (br.past_end()) { r.status = RUN_PAST_END; break; }
                if (++batches > MAX_BATCHES) { r.status = RUN_TOO_LONG; break; }
            }
            if (r.status != RUN_OK) break;
        }
        if (br.past_end()) { r.status = RUN_PAST_END; break; }
        if (final_block) { r.status = RUN_FINAL; break; }
    }
    sink.finish();
    if (sink.bad && (r.status == RUN_OK || r.status == RUN_FINAL)) r.status = RUN_TOO_LONG;
    r.end_bit = br.pos();
    r.out_len = sink.total();
    r.n_records = sink.n;
    if (r.out_len > max_out && (r.status == RUN_OK || r.status == RUN_FINAL)) r.status = RUN_TOO_LONG;
    return r;
}

// ---- CRC-32 (the gzip trailer's; reflected polynomial 0xEDB88320), host side of the check: pieces -> whole ----
// a(x) * b(x) mod p(x) on reflected 32-bit polynomials (bit 31 = x^0)
inline uint32_t crc_mul(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (uint32_t m = 1u << 31; m; m >>= 1) {
        if (a & m) p ^= b;
        b = b & 1u ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}
// x^(8 n) mod p(x)
inline uint32_t crc_xpow8(uint64_t n_bytes) {
    uint32_t r = 1u << 31, sq = 1u << 23;                            // x^0; x^8
    for (uint64_t n = n_bytes; n; n >>= 1) {
        if (n & 1u) r = crc_mul(r, sq);
        sq = crc_mul(sq, sq);
    }
    return r;
}
// crc of A ++ B from crc(A), crc(B) and x^(8 |B|)
inline uint32_t crc_join(uint32_t crc_a, uint32_t crc_b, uint32_t xpow_b) { return crc_mul(xpow_b, crc_a) ^ crc_b; }

}  // namespace inf
}  // namespace smg

// ---- host only: gzip framing, the chain of runs ----
#include <algorithm>
#include <string>
#include <vector>
namespace smg {
namespace inf {

// one gzip member inside a buffer (RFC 1952): where its deflate data begins, and what the trailer promises
struct Member {
    uint64_t deflate_byte = 0;      // first byte of the deflate stream (from the start of the buffer)
    uint64_t end_byte = 0;          // first byte behind the member's trailer = end of the file for a single member
    uint32_t want_crc = 0, want_isize = 0;
};

// header of a gzip file that is ONE member: false if it is not gzip / not deflate / too short
inline bool parse_single_member(const uint8_t* p, uint64_t size, Member& m) {
    if (size < 18 + 8 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8) return false;
    const uint8_t flg = p[3];
    if (flg & 0xe0) return false;                                     // reserved bits
    uint64_t o = 10;
    if (flg & 4) { if (o + 2 > size) return false; o += 2 + (uint64_t)(p[o] | (p[o + 1] << 8)); }
    if (flg & 8) { while (o < size && p[o]) ++o; ++o; }
    if (flg & 16) { while (o < size && p[o]) ++o; ++o; }
    if (flg & 2) o += 2;
    if (o + 8 >= size) return false;
    m.deflate_byte = o;
    m.end_byte = size;
    const uint8_t* t = p + size - 8;
    m.want_crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    m.want_isize = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
    return true;
}

// a candidate block start after pass 1
struct Cand {
    uint64_t bit = 0, end_bit = 0, out_len = 0;
    uint32_t status = 0, n_records = 0;
};

// The chain of runs from `first_bit` to the end of the final block.  cands: sorted by bit, holding first_bit itself.
// trailer_bit: where the member's 8-byte trailer begins (the final block must end in the byte in front of it).
// -> indices into cands, in stream order; empty + why on any gap.
inline std::vector<uint32_t> link_chain(const std::vector<Cand>& cands, uint64_t first_bit, uint64_t trailer_bit, std::string& why) {
    std::vector<uint32_t> chain;
    uint64_t at = first_bit;
    for (;;) {
        auto it = std::lower_bound(cands.begin(), cands.end(), at, [](const Cand& c, uint64_t b) { return c.bit < b; });
        if (it == cands.end() || it->bit != at) { why = "no block start was found at bit " + std::to_string(at) + " where the run in front ends"; return {}; }
        const Cand& c = *it;
        if (c.status != RUN_OK && c.status != RUN_FINAL) { why = "the run at bit " + std::to_string(at) + " does not decode (status " + std::to_string(c.status) + ")"; return {}; }
        if (c.end_bit <= at) { why = "a run of no bits"; return {}; }
        chain.push_back((uint32_t)(it - cands.begin()));
        if (c.status == RUN_FINAL) {
            if (((c.end_bit + 7) & ~7ull) != trailer_bit) { why = "the deflate stream ends at bit " + std::to_string(c.end_bit) + ", not in front of the trailer (more than one member?)"; return {}; }
            return chain;
        }
        at = c.end_bit;
    }
}

}  // namespace inf
}  // namespace smg
