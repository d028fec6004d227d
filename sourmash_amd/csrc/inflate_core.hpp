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

// status of a decoded run
enum : uint32_t { RUN_OK = 0, RUN_FINAL = 1, RUN_BAD_CODE = 2, RUN_BAD_BLOCK = 3, RUN_PAST_END = 4, RUN_BAD_DISTANCE = 5, RUN_TOO_LONG = 6 };

// ---- bits: aligned 32-bit words, least significant bit first (RFC 1951 3.1.1) ----
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
struct Code {
    uint16_t* table;            // 1 << root entries: symbol << 4 | length; 0 = no code this short
    uint16_t* sorted;           // symbols ordered by (length, symbol)
    uint16_t count[16], first[16], offset[16];
};

// lens[0, n) -> code.  false: over-subscribed, or incomplete with more than one code / a code longer than one bit (the sets
// zlib's inflate_table refuses).
SMG_HD bool build_code(const uint8_t* lens, int n, int root, Code& c) {
    for (int i = 0; i < 16; ++i) c.count[i] = 0;
    for (int i = 0; i < n; ++i) c.count[lens[i]]++;
    c.count[0] = 0;
    int left = 1, used = 0, maxlen = 0;
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - (int)c.count[l];
        if (left < 0) return false;
        used += c.count[l];
        if (c.count[l]) maxlen = l;
    }
    if (left > 0 && maxlen > 1) return false;
    uint32_t code = 0, off = 0;
    for (int l = 1; l <= 15; ++l) {
        code = (code + (l > 1 ? c.count[l - 1] : 0u)) << 1;
        c.first[l] = (uint16_t)code;
        c.offset[l] = (uint16_t)off;
        off += c.count[l];
    }
    c.first[0] = c.offset[0] = 0;
    for (int i = 0; i < (1 << root); ++i) c.table[i] = 0;
    uint16_t fill[16];
    for (int l = 0; l < 16; ++l) fill[l] = 0;
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t rank = fill[l]++;
        c.sorted[c.offset[l] + rank] = (uint16_t)s;
        if (l <= root) {
            const uint32_t cw = c.first[l] + rank;                     // canonical code, first bit = most significant
            uint32_t r = 0;
            for (int b = 0; b < l; ++b) r |= ((cw >> b) & 1u) << (l - 1 - b);
            const uint16_t e = (uint16_t)((s << 4) | l);
            for (uint32_t i = r; i < (1u << root); i += 1u << l) c.table[i] = e;
        }
    }
    return true;
}

// next symbol of the code (at least 15 bits in the reader); -1: no such code
SMG_HD int decode_sym(BitReader& br, const Code& c, int root) {
    const uint32_t v = (uint32_t)br.buf;
    const uint32_t e = c.table[v & ((1u << root) - 1u)];
    if (e & 15u) { br.drop(e & 15u); return (int)(e >> 4); }
    const uint32_t r = bitrev15(v & 0x7fffu);
    for (int l = root + 1; l <= 15; ++l) {
        const uint32_t d = (r >> (15 - l)) - c.first[l];
        if (d < c.count[l]) { br.drop((uint32_t)l); return (int)c.sorted[c.offset[l] + d]; }
    }
    return -1;
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
struct Scratch {
    uint16_t lit_table[1 << LIT_ROOT], dist_table[1 << DIST_ROOT];
    uint16_t lit_sorted[288], dist_sorted[32];
    uint16_t cl_table[128], cl_sorted[20];
    uint8_t lens[288 + 32 + 8];
};

// Code lengths of a dynamic block (RFC 1951 3.2.7): the reader stands behind the 3 header bits.  strict: the conditions a
// SCANNED block start must meet on top of being decodable (end-of-block code present, the distance code complete, a single
// code, or absent).  -> false: not a dynamic block header.
SMG_HD bool read_dynamic_lengths(BitReader& br, Scratch& S, int& hlit, int& hdist, bool strict) {
    br.refill();
    hlit = (int)br.take(5) + 257;
    hdist = (int)br.take(5) + 1;
    const int hclen = (int)br.take(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19];
    for (int i = 0; i < 19; ++i) cl[i] = 0;
    for (int i = 0; i < hclen; ++i) { br.refill(); cl[order[i]] = (uint8_t)br.take(3); }
    Code cc;
    cc.table = S.cl_table;
    cc.sorted = S.cl_sorted;
    if (!build_code(cl, 19, 7, cc)) return false;
    if (strict) {                                                     // the code-length code itself: complete
        int left = 1;
        for (int l = 1; l <= 7; ++l) left = (left << 1) - (int)cc.count[l];
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
        const int sym = (int)(e >> 4);
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
// complete, the distance code complete -- or one code, or none.  words: the whole buffer; end_bit: its last bit.
SMG_HD bool valid_dynamic_header(const uint32_t* words, uint64_t bit, uint64_t end_bit, Scratch& S) {
    BitReader br;
    br.init(words, bit, end_bit);
    if (br.take(3) != 4u) return false;
    int hlit, hdist;
    if (!read_dynamic_lengths(br, S, hlit, hdist, true)) return false;
    if (br.past_end()) return false;
    int left = 1;
    uint32_t cnt[16];
    for (int i = 0; i < 16; ++i) cnt[i] = 0;
    for (int i = 0; i < hlit; ++i) cnt[S.lens[i]]++;
    for (int l = 1; l <= 15; ++l) { left = (left << 1) - (int)cnt[l]; if (left < 0) return false; }
    if (left != 0) return false;
    for (int i = 0; i < 16; ++i) cnt[i] = 0;
    int used = 0;
    for (int i = 0; i < hdist; ++i) { cnt[S.lens[hlit + i]]++; used += S.lens[hlit + i] != 0; }
    left = 1;
    for (int l = 1; l <= 15; ++l) { left = (left << 1) - (int)cnt[l]; if (left < 0) return false; }
    if (left != 0 && !(used == 0 || (used == 1 && cnt[1] == 1))) return false;
    return true;
}

// ---- sinks: what a decoded run is turned into ----

// pass 1: only the number of bytes
struct CountSink {
    uint64_t n = 0;
    bool bad = false;
    SMG_HD void literal(uint32_t) { ++n; }
    SMG_HD void match(uint32_t len, uint32_t) { n += len; }
    SMG_HD void stored(const uint8_t*, uint32_t len) { n += len; }
    SMG_HD void finish() {}
};

// pass 2: 16-bit symbols at their final place.  The decoder is uniform across the wavefront; the sink gives every symbol's
// bytes to the next free lanes of a 64-wide group (lane i <-> output position g0 + i), and when the group is full the lanes
// gather their sources (the run's own earlier output, or a marker for what lies in front of the run) and store 64 symbols side
// by side.  A match whose source is in the unstored group flushes first; then every source of it lies behind stores already
// issued -- same wavefront, program order -- and an overlapping match (distance < length) reads its period.
#if defined(__HIP_DEVICE_COMPILE__)
#define SMG_INF_LANES 1
#define SMG_INF_EACH_LANE(lane, slot) const uint32_t lane = (uint32_t)(threadIdx.x & 63u); constexpr int slot = 0;
#else
#define SMG_INF_LANES 64
#define SMG_INF_EACH_LANE(lane, slot) for (uint32_t lane = 0, slot = 0; lane < 64u; ++lane, ++slot)
#endif

struct WaveSink {
    uint16_t* out;              // the run's first symbol
    uint64_t cap;               // symbols the run may write (pass 1's count)
    bool no_window;             // the member's first run: nothing in front of it
    uint32_t g0 = 0, o = 0;     // output position of lane 0 of the open group; lanes filled
    bool bad = false;
    int32_t src[SMG_INF_LANES]; // per lane: >= 0 literal symbol | 0x40000000; else source position relative to the run start, as (pos - 2^30) ... see below
    // encoding of src: bit 30 set -> literal in bits 0..15; otherwise a signed position (negative: in front of the run)

    SMG_HD void literal(uint32_t b) {
        { SMG_INF_EACH_LANE(lane, slot) { if (lane == o) src[slot] = (int32_t)(0x40000000u | b); } }
        if (++o == 64u) flush();
    }
    SMG_HD void match(uint32_t len, uint32_t dist) {
        uint32_t k0 = 0;
        int64_t mstart = (int64_t)g0 + o;
        while (k0 < len) {
            uint32_t n = len - k0 < 64u - o ? len - k0 : 64u - o;
            if (o > 0 && dist < o + n) {                              // a source in the unstored group: store it first
                flush();
                n = len - k0 < 64u ? len - k0 : 64u;
            }
            const int64_t base = mstart - (int64_t)dist;
            const bool periodic = dist < len;
            { SMG_INF_EACH_LANE(lane, slot) {
                const uint32_t i = lane - o;
                if (i < n) {
                    uint32_t idx = k0 + i;
                    if (periodic) idx %= dist;
                    const int64_t p = base + idx;
                    src[slot] = (int32_t)(p < -(int64_t)WIN ? -(int32_t)WIN - 1 : p);      // (beyond the window: flagged at the flush)
                }
            } }
            o += n;
            k0 += n;
            if (o == 64u) flush();
        }
    }
    SMG_HD void stored(const uint8_t* bytes, uint32_t len) {
        uint32_t k = 0;
        while (k < len) {
            const uint32_t n = len - k < 64u - o ? len - k : 64u - o;
            { SMG_INF_EACH_LANE(lane, slot) {
                const uint32_t i = lane - o;
                if (i < n) src[slot] = (int32_t)(0x40000000u | bytes[k + i]);
            } }
            o += n;
            k += n;
            if (o == 64u) flush();
        }
    }
    SMG_HD void flush() {
        if ((uint64_t)g0 + o > cap) { bad = true; o = 0; return; }
        bool wrong = false;
        { SMG_INF_EACH_LANE(lane, slot) {
            if (lane < o) {
                const int32_t s = src[slot];
                uint16_t sym;
                if (s & 0x40000000 && s >= 0) sym = (uint16_t)(s & 0xffff);
                else if (s >= 0) sym = out[s];
                else if (s < -(int32_t)WIN || no_window) { sym = 0; wrong = true; }
                else sym = (uint16_t)(MARK | (uint32_t)(s + (int32_t)WIN));
                out[g0 + lane] = sym;
            }
        } }
#if defined(__HIP_DEVICE_COMPILE__)
        wrong = __builtin_amdgcn_ballot_w64(wrong) != 0ull;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the stores above are in front of every later load of this wavefront
#endif
        if (wrong) bad = true;
        g0 += o;
        o = 0;
    }
    SMG_HD void finish() { if (o) flush(); }
};

struct RunResult {
    uint64_t end_bit;           // where the run stopped: the next dynamic non-final header, or behind the final block
    uint64_t out_len;
    uint32_t status;            // RUN_OK: stopped in front of a block header; RUN_FINAL: the stream's last block is done; else an error
};

// Decode from `bit` (a block header) through stored / fixed / final blocks until the next dynamic non-final block header
// (the next run's start) or the end of the final block.  max_out bounds a false candidate's run.
template <class Sink>
SMG_HD RunResult decode_run(const uint32_t* words, uint64_t bit, uint64_t end_bit, Scratch& S, Sink& sink, uint64_t max_out) {
    RunResult r;
    r.out_len = 0;
    r.status = RUN_OK;
    BitReader br;
    br.init(words, bit, end_bit);
    Code lit, dist;
    lit.table = S.lit_table; lit.sorted = S.lit_sorted;
    dist.table = S.dist_table; dist.sorted = S.dist_sorted;
    uint64_t produced = 0;
    bool first = true;
    for (;;) {
        br.refill();
        if (br.pos + 3 > br.end) { r.status = RUN_PAST_END; break; }
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
            br.refill();
            const uint32_t nlen = br.take(16);
            if ((len ^ nlen) != 0xffffu) { r.status = RUN_BAD_BLOCK; break; }
            if (br.pos + (uint64_t)len * 8 > br.end) { r.status = RUN_PAST_END; break; }
            sink.stored(reinterpret_cast<const uint8_t*>(words) + (br.pos >> 3), len);
            produced += len;
            br.init(words, br.pos + (uint64_t)len * 8, end_bit);
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
                for (int i = 0; i < 30; ++i) S.lens[288 + i] = 5;
                ok = build_code(S.lens, 288, LIT_ROOT, lit) && build_code(S.lens + 288, 30, DIST_ROOT, dist);
            } else {
                int hlit, hdist;
                ok = read_dynamic_lengths(br, S, hlit, hdist, false);
                ok = ok && build_code(S.lens, hlit, LIT_ROOT, lit) && build_code(S.lens + hlit, hdist, DIST_ROOT, dist);
            }
            if (!ok) { r.status = RUN_BAD_BLOCK; break; }
            for (;;) {
                br.refill();
                int sym = decode_sym(br, lit, LIT_ROOT);
                if (sym < 256) {
                    if (sym < 0) { r.status = RUN_BAD_CODE; break; }
                    sink.literal((uint32_t)sym);
                    ++produced;
                } else if (sym == 256) {
                    break;
                } else {
                    const uint32_t ls = (uint32_t)sym - 257u;
                    if (ls > 28u) { r.status = RUN_BAD_CODE; break; }
                    const uint32_t len = len_base(ls) + br.take(len_extra(ls));
                    br.refill();
                    const int ds = decode_sym(br, dist, DIST_ROOT);
                    if (ds < 0 || ds > 29) { r.status = RUN_BAD_CODE; break; }
                    const uint32_t d = dist_base((uint32_t)ds) + br.take(dist_extra((uint32_t)ds));
                    sink.match(len, d);
                    produced += len;
                }
                if (br.past_end()) { r.status = RUN_PAST_END; break; }
                if (produced > max_out) { r.status = RUN_TOO_LONG; break; }
            }
            if (r.status != RUN_OK) break;
        }
        if (br.past_end()) { r.status = RUN_PAST_END; break; }
        if (final_block) { r.status = RUN_FINAL; break; }
    }
    sink.finish();
    if (sink.bad && (r.status == RUN_OK || r.status == RUN_FINAL)) r.status = RUN_BAD_DISTANCE;
    r.end_bit = br.pos;
    r.out_len = produced;
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
