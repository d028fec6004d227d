// Host emulation of the sketch kernel's lane logic (test-only artefact).
// Compiles sourmash_amd/csrc/kmer_core.hpp for the CPU (v_perm / v_alignbyte
// emulated) and walks a buffer exactly as the HIP kernel does: lanes of P
// start positions, window bytes past the end read as 0.  tests/
// test_kmer_core_cpu.py compares the result with the oracle.
#include <cstring>
#include <vector>
#include <algorithm>
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

extern "C" uint64_t emul_sketch(const uint8_t* seq, uint64_t len, uint32_t k, uint32_t p, uint64_t seed,
                                uint64_t thr, uint64_t* out, uint64_t cap) {
#define CASE(KK, PP) if (k == KK && p == PP) return run<KK, PP>(seq, len, seed, thr, out, cap);
    CASE(31, 16) CASE(31, 8) CASE(21, 16) CASE(51, 16) CASE(4, 16) CASE(3, 16) CASE(5, 16) CASE(10, 16)
    CASE(16, 16) CASE(32, 16) CASE(17, 8) CASE(15, 4) CASE(33, 16) CASE(63, 16) CASE(1, 16) CASE(8, 16) CASE(9,16)
#undef CASE
    return ~0ull;
}
