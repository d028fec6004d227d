// Host emulation of the long-k sketch kernel (test-only artefact): sourmash_amd/csrc/kmer_words.hpp compiled for the CPU and driven
// stretch by stretch exactly as sketch_words.hip's kernel does -- the same staging (16-byte chunks, upper-casing, bad-byte bits,
// counts in front of every 32-byte word, the reverse-complement copy, junk in the slack), the same per-position calls.  tests/test_kmer_words_cpu.py
// compares the result with the oracle.
#include <cstring>
#include <vector>
#include "../../sourmash_amd/csrc/kmer_words.hpp"

// out[i] = hash of the k-mer starting at i, 0 where it holds a byte outside ACGT; `tile` start positions per stretch, `skip` bytes of
// alignment prefix in front of the buffer (the launcher's pointer rounding).  Returns the number of good k-mers.
extern "C" uint64_t emul_words_dense(const uint8_t* seq_in, uint64_t len_in, uint32_t k, uint32_t tile, uint32_t skip, uint64_t seed,
                                     uint64_t* out, uint64_t cap) {
    if (k < 16 || len_in < k) return 0;
    std::vector<uint8_t> seq(skip + len_in, 0x41);          // the prefix holds VALID bases: blanking it is the stager's job
    std::memcpy(seq.data() + skip, seq_in, len_in);
    const uint64_t len = skip + len_in;
    const uint32_t bytes = tile + k - 1;
    const uint32_t n_chunks = (bytes + 15) / 16, n_words = (n_chunks + 1) / 2 + 1;
    const smg::WwLayout lay = smg::ww_layout(n_chunks);
    const uint32_t win_dwords = lay.dwords();
    std::vector<uint32_t> win(win_dwords), bits(n_words), before(n_words);
    uint64_t good = 0;
    for (uint64_t base = 0; base < len; base += tile) {
        std::fill(win.begin(), win.end(), 0xa5a5a5a5u);     // junk everywhere the stager must write
        std::fill(bits.begin(), bits.end(), 0xffffffffu);
        for (uint32_t c = 0; c < n_chunks; ++c) {
            const uint64_t off = base + (uint64_t)c * 16;
            uint32_t w[4] = {0, 0, 0, 0};
            for (uint64_t b = off; b < len && b < off + 16; ++b) w[(b - off) >> 2] |= (uint32_t)seq[b] << (8 * ((b - off) & 3));
            if (off == 0 && skip)
                for (uint32_t b = 0; b < skip; ++b) w[b >> 2] &= ~(0xffu << (8 * (b & 3)));
            uint32_t nib = 0;
            for (int i = 0; i < 4; ++i) {
                w[i] &= 0xdfdfdfdfu;
                nib |= smg::ww_bad4(w[i]) << (4 * i);
                win[c * 4 + i] = w[i];
                win[lay.rc_off / 4 + (n_chunks - 1 - c) * 4 + (3 - i)] = smg::ww_revcomp4(w[i]);
            }
            reinterpret_cast<uint16_t*>(bits.data())[c] = (uint16_t)nib;
        }
        for (uint32_t c = n_chunks; c < n_words * 2; ++c) reinterpret_cast<uint16_t*>(bits.data())[c] = 0;
        uint32_t run = 0;
        for (uint32_t w = 0; w < n_words; ++w) { before[w] = run; run += (uint32_t)__builtin_popcount(bits[w]); }
        const smg::WwBad bad{bits.data(), before.data()};
        for (uint32_t p = 0; p < tile; ++p) {
            const bool ok = base + p + k <= len && (run == 0 || bad.clean(p, k));
            if (!ok) continue;
            const uint64_t h = smg::ww_hash(win.data(), lay, p, k, seed);
            const uint64_t pos = base + p - skip;
            ++good;
            if (pos < cap) out[pos] = h;
        }
    }
    return good;
}
