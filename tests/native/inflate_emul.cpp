// Host run of the device inflate scheme (sourmash_amd/csrc/inflate_core.hpp): scan -> pass 1 -> link -> pass 2 (the wave
// sink with its 64 lanes as a loop) -> tails -> resolve, on a gzip member held in memory.  Test infrastructure: the kernels
// of gunzip.hip run this same header on the device.
#include "../../sourmash_amd/csrc/inflate_core.hpp"
#include <stdio.h>
#include <memory>

using namespace smg::inf;

extern "C" {

// -> number of candidates (after the full test) written to out_bits (capacity cap); *n_prefix = survivors of the cheap test
uint64_t emul_scan(const uint8_t* data, uint64_t n_bytes, uint64_t from_bit, uint64_t to_bit, uint64_t* out_bits, uint64_t cap, uint64_t* n_prefix) {
    std::vector<uint32_t> words((n_bytes + 3) / 4 + 300, 0);
    memcpy(words.data(), data, n_bytes);
    PlainTab tab;
    uint64_t n = 0, np = 0;
    for (uint64_t b = from_bit; b < to_bit; ++b) {
        const uint64_t wi = b >> 6;
        const uint32_t sh = (uint32_t)(b & 63);
        uint64_t w0, w1, w2;
        memcpy(&w0, (const uint8_t*)words.data() + wi * 8, 8);
        memcpy(&w1, (const uint8_t*)words.data() + wi * 8 + 8, 8);
        memcpy(&w2, (const uint8_t*)words.data() + wi * 8 + 16, 8);
        const uint64_t lo = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
        const uint64_t hi = sh ? (w1 >> sh) | (w2 << (64 - sh)) : w1;
        if (!plausible_prefix(lo, hi)) continue;
        ++np;
        if (!valid_dynamic_header(words.data(), b, n_bytes * 8, tab)) continue;
        if (n < cap) out_bits[n] = b;
        ++n;
    }
    if (n_prefix) *n_prefix = np;
    return n;
}

// the whole scheme on one member; out: capacity cap.  -> bytes produced, or -(code) : 1 not a single gzip member, 2 chain broken,
// 3 pass 2 disagrees with pass 1, 4 a marker without a window, 5 output capacity, 6 trailer mismatch.  stats[0] candidates,
// [1] runs on the chain, [2] markers written by pass 2
int64_t emul_gunzip(const uint8_t* file, uint64_t size, uint8_t* out, uint64_t cap, uint64_t* stats, char* why_out, uint64_t why_cap) {
    Member m;
    if (!parse_single_member(file, size, m)) return -1;
    std::vector<uint32_t> words((size + 3) / 4 + 300, 0);
    memcpy(words.data(), file, size);
    const uint64_t first_bit = m.deflate_byte * 8, trailer_bit = (size - 8) * 8;
    std::vector<uint64_t> bits(1 << 20);
    uint64_t np = 0;
    const uint64_t nc = emul_scan(file, size, first_bit, trailer_bit, bits.data(), bits.size(), &np);
    if (nc > bits.size()) return -5;
    std::vector<Cand> cands;
    cands.reserve(nc + 1);
    { Cand c; c.bit = first_bit; cands.push_back(c); }
    for (uint64_t i = 0; i < nc; ++i)
        if (bits[i] != first_bit) { Cand c; c.bit = bits[i]; cands.push_back(c); }
    std::unique_ptr<Scratch> S(new Scratch());
    std::vector<uint32_t> rec(size * 8 + 64);
    for (size_t i = 0; i < cands.size(); ++i) {                       // pass 1: records at [bit, next candidate's bit)
        Cand& c = cands[i];
        RecordSink sink;
        sink.rec = rec.data() + c.bit;
        sink.cap = (i + 1 < cands.size() ? cands[i + 1].bit : trailer_bit) - c.bit;
        const RunResult r = decode_run(words.data(), c.bit, trailer_bit, *S, sink, MAX_RUN_BYTES);
        c.end_bit = r.end_bit; c.out_len = r.out_len; c.status = r.status; c.n_records = r.n_records;
    }
    std::string why;
    const std::vector<uint32_t> chain = link_chain(cands, first_bit, trailer_bit, why);
    if (chain.empty()) { if (why_out && why_cap) snprintf(why_out, why_cap, "%s", why.c_str()); return -2; }
    uint64_t total = 0;
    std::vector<uint64_t> start(chain.size());
    for (size_t i = 0; i < chain.size(); ++i) { start[i] = total; total += cands[chain[i]].out_len; }
    if (stats) { stats[0] = cands.size(); stats[1] = chain.size(); stats[2] = 0; }
    if (total > cap) return -5;
    std::vector<uint16_t> sym(total + 64);
    std::unique_ptr<ExpandScratch> X(new ExpandScratch());
    for (size_t i = 0; i < chain.size(); ++i) {                       // pass 2: records -> symbols
        const Cand& c = cands[chain[i]];
        Expander ex;
        ex.out = sym.data() + start[i];
        ex.cap = c.out_len;
        ex.bytes = (const uint8_t*)words.data();
        ex.no_window = i == 0;
        ex.run(rec.data() + c.bit, c.n_records, *X);
        if (ex.bad || ex.g0 != c.out_len) return -3;
    }
    // tails in stream order, then everything else (here: one loop does both, in order)
    uint64_t markers = 0;
    for (size_t i = 0; i < chain.size(); ++i) {
        const uint64_t s0 = start[i], n = cands[chain[i]].out_len;
        for (uint64_t k = 0; k < n; ++k) {
            const uint16_t v = sym[s0 + k];
            if (v & MARK) {
                ++markers;
                const uint64_t mpos = s0 + (uint64_t)(v & 0x7fff);
                if (mpos < WIN) return -4;
                out[s0 + k] = out[mpos - WIN];
            } else out[s0 + k] = (uint8_t)v;
        }
    }
    if (stats) stats[2] = markers;
    if ((uint32_t)(total & 0xffffffffu) != m.want_isize) return -6;
    return (int64_t)total;
}

uint32_t emul_crc_join(uint32_t a, uint32_t b, uint64_t len_b) { return crc_join(a, b, crc_xpow8(len_b)); }

}
