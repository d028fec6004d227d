// Host emulation of the sketch kernel's lane logic (test-only artefact).
// Compiles sourmash_amd/csrc/kmer_core.hpp for the CPU (v_perm / v_alignbyte
// emulated) and walks a buffer exactly as the HIP kernel does: lanes of P
// start positions, window bytes past the end read as 0.  tests/
// test_kmer_core_cpu.py compares the result with the oracle.
#include <cstring>
#include <vector>
#include <algorithm>
#include <utility>
#include "../../sourmash_amd/csrc/kmer_core.hpp"

template <int K, int P>
static uint64_t run(const uint8_t* seq, uint64_t len, uint64_t seed, uint64_t thr, uint64_t* out, uint64_t cap) {
    using G = smg::LaneGeom<K, P>;
    uint64_t n = 0;
    for (uint64_t start = 0; start < len; start += P) {
        uint32_t raw[G::NW];
        uint8_t bytes[G::NW * 4];
        for (int b = 0; b < G::NW * 4; ++b) bytes[b] = (start + b < len) ? seq[start + b] : 0;
        // slack bytes beyond NBYTES may hold real data on the GPU; mimic with junk
        for (int b = G::NBYTES; b < G::NW * 4; ++b) bytes[b] = (start + b < len) ? seq[start + b] : (uint8_t)'N';
        std::memcpy(raw, bytes, sizeof(raw));
        smg::process_lane<K, P>(raw, seed, thr, [&](int, uint64_t h) { if (n < cap) out[n] = h; ++n; });
    }
    return n;
}

// every ksize the GPU dispatch instantiates (sketch.hip / sketch_long.hip: K = 1 .. 128 at P = 16) plus a few other lane widths
typedef uint64_t (*run_fn)(const uint8_t*, uint64_t, uint64_t, uint64_t, uint64_t*, uint64_t);
template <int... KS>
static run_fn pick16(uint32_t k, std::integer_sequence<int, KS...>) {
    static const run_fn table[] = {&run<KS + 1, 16>...};
    return k >= 1 && k <= sizeof...(KS) ? table[k - 1] : nullptr;
}

extern "C" uint64_t emul_sketch(const uint8_t* seq, uint64_t len, uint32_t k, uint32_t p, uint64_t seed,
                                uint64_t thr, uint64_t* out, uint64_t cap) {
    if (p == 16) {
        const run_fn f = pick16(k, std::make_integer_sequence<int, 128>());
        return f ? f(seq, len, seed, thr, out, cap) : ~0ull;
    }
#define CASE(KK, PP) if (k == KK && p == PP) return run<KK, PP>(seq, len, seed, thr, out, cap);
    CASE(31, 8) CASE(17, 8) CASE(15, 4)
#undef CASE
    return ~0ull;
}

// murmur3.hpp's split fmix64: returns the number of (a, b) pairs out of n for which the open form disagrees with
// the closed one (full value, or top dword not in {t, t - 1}).
extern "C" uint64_t emul_open_form_violations(uint64_t n, uint64_t seed) {
    uint64_t bad = 0, x = seed;
    auto next = [&] { x += 0x9e3779b97f4a7c15ULL; uint64_t z = x; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
                      z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); };
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t h1 = next(), h2 = next();
        if (i % 7 == 0) h2 = (uint64_t)0 - h1;                 // sums that wrap
        if (smg::fmix64(h1) != smg::fmix64_tail(smg::fmix64_head(h1))) ++bad;
        smg::Mmh3Open o{smg::fmix64_head(h1), smg::fmix64_head(h2)};
        const uint64_t h = smg::fmix64(h1) + smg::fmix64(h2);
        if (smg::mmh3_close(o) != h) ++bad;
        const uint32_t t = (uint32_t)(h >> 32), s = smg::mmh3_close_hi(o);
        if (s != t && (uint32_t)(s + 1u) != t) ++bad;
    }
    return bad;
}
