// Host emulation of the residue-window kernel's lane logic (test-only artefact): sourmash_amd/csrc/residue_core.hpp compiled for
// the CPU and walked lane by lane exactly as protein.hip's kernel does, next to the naive definition (every window's k bytes
// hashed byte by byte, windows touching a 0xFF separator skipped).  tests/test_residue_core_cpu.py compares both with each other
// and with the oracle.
#include <cstring>
#include <vector>
#include "../../sourmash_amd/csrc/residue_core.hpp"

template <int NB>
static uint64_t run(const uint8_t* aa, uint64_t n, uint32_t k, uint64_t seed, uint64_t* starts, uint64_t* hashes, uint64_t cap) {
    const smg::ResidueTail t = smg::residue_tail(k);
    // the kernel reads aligned words: a copy padded to whole words, junk in the padding (it must not matter)
    std::vector<uint64_t> words((n + 7) / 8 + 1, 0x5a5a5a5a5a5a5a5aull);
    std::memcpy(words.data(), aa, n);
    uint64_t cnt = 0;
    const uint64_t lanes = (n + 7) / 8;
    for (uint64_t g = 0; g < lanes; ++g)
        smg::residue_windows_lane<NB>(words.data(), n, g, t, seed, [&](uint64_t start, uint64_t h) {
            if (cnt < cap) { starts[cnt] = start; hashes[cnt] = h; }
            ++cnt;
        });
    return cnt;
}

extern "C" uint64_t emul_residue_windows(const uint8_t* aa, uint64_t n, uint32_t k, uint64_t seed, uint64_t* starts, uint64_t* hashes, uint64_t cap) {
    if (k == 0 || n < k) return 0;
    switch (k / 16) {
    case 0: return run<0>(aa, n, k, seed, starts, hashes, cap);
    case 1: return run<1>(aa, n, k, seed, starts, hashes, cap);
    case 2: return run<2>(aa, n, k, seed, starts, hashes, cap);
    case 3: return run<3>(aa, n, k, seed, starts, hashes, cap);
    case 4: return run<4>(aa, n, k, seed, starts, hashes, cap);
    default: return ~0ull;
    }
}

extern "C" uint64_t naive_residue_windows(const uint8_t* aa, uint64_t n, uint32_t k, uint64_t seed, uint64_t* starts, uint64_t* hashes, uint64_t cap) {
    uint64_t cnt = 0;
    if (k == 0 || n < k) return 0;
    for (uint64_t i = 0; i + k <= n; ++i) {
        bool ok = true;
        for (uint32_t j = 0; j < k; ++j) ok = ok && aa[i + j] != 0xff;
        if (!ok) continue;
        if (cnt < cap) { starts[cnt] = i; hashes[cnt] = smg::mmh3_h1_bytes(aa + i, k, seed); }
        ++cnt;
    }
    return cnt;
}

// The translate kernel's tables (protein.hip: byte -> nucleotide code, three codes -> residue) against the scalar functions they
// are filled from, over every triple of bytes and the three alphabets, both strands: -> number of disagreements.
#include "../../sourmash_amd/csrc/residues.hpp"
extern "C" uint64_t check_translate_tables() {
    uint64_t bad = 0;
    for (uint32_t hf = 2; hf <= 4; ++hf) {
        uint8_t code_f[256], code_r[256], codon[216];
        for (int t = 0; t < 256; ++t) {
            const uint8_t c = smg::ascii_upper((uint8_t)t);
            code_f[t] = (uint8_t)smg::nt_code(c);
            code_r[t] = (uint8_t)smg::nt_code(smg::dna_complement_or_nul(c));
        }
        const char* letters = "ACGTN?";
        for (int t = 0; t < 216; ++t)
            codon[t] = smg::residue_encode(smg::translate_codon((uint8_t)letters[t / 36], (uint8_t)letters[(t / 6) % 6], (uint8_t)letters[t % 6]), hf);
        for (int a = 0; a < 256; ++a)
            for (int b = 0; b < 256; ++b)
                for (int c = 0; c < 256; ++c) {
                    const uint8_t ua = smg::ascii_upper((uint8_t)a), ub = smg::ascii_upper((uint8_t)b), uc = smg::ascii_upper((uint8_t)c);
                    const uint8_t fwd = smg::residue_encode(smg::translate_codon(ua, ub, uc), hf);
                    bad += fwd != codon[code_f[a] * 36 + code_f[b] * 6 + code_f[c]];
                    const uint8_t rev = smg::residue_encode(smg::translate_codon(smg::dna_complement_or_nul(ua), smg::dna_complement_or_nul(ub),
                                                                                 smg::dna_complement_or_nul(uc)), hf);
                    bad += rev != codon[code_r[a] * 36 + code_r[b] * 6 + code_r[c]];
                }
    }
    return bad;
}

// The translate kernel's lane (translate_core.hpp: one aligned output word per lane) against the per-byte definition.
#include "../../sourmash_amd/csrc/translate_core.hpp"
extern "C" uint64_t emul_translate(const uint8_t* seq, uint64_t len, uint32_t hf, uint8_t* out_fast, uint8_t* out_naive) {
    uint8_t code_f[256], code_r[256], codon[216];
    for (int t = 0; t < 256; ++t) {
        const uint8_t c = smg::ascii_upper((uint8_t)t);
        code_f[t] = (uint8_t)smg::nt_code(c);
        code_r[t] = (uint8_t)smg::nt_code(smg::dna_complement_or_nul(c));
    }
    const char* letters = "ACGTN?";
    for (int t = 0; t < 216; ++t)
        codon[t] = smg::residue_encode(smg::translate_codon((uint8_t)letters[t / 36], (uint8_t)letters[(t / 6) % 6], (uint8_t)letters[t % 6]), hf);
    const smg::TranslateTables T{code_f, code_r, codon};
    const smg::TranslateLayout L = smg::translate_layout(len);
    // the kernel reads aligned words: a padded copy, junk in the padding
    std::vector<uint32_t> words((len + 3) / 4 + 1, 0x51515151u);
    std::memcpy(words.data(), seq, len);
    const uint8_t* s8 = reinterpret_cast<const uint8_t*>(words.data());
    const uint64_t total = L.start[6];
    for (uint64_t G = 0; G < (total + 3) / 4; ++G) {
        const uint32_t w = smg::translate_word(s8, words.data(), L, T, G);
        for (int j = 0; j < 4; ++j)
            if (4 * G + (uint64_t)j < total) out_fast[4 * G + (uint64_t)j] = (uint8_t)(w >> (8 * j));
    }
    for (uint64_t o = 0; o < total; ++o) out_naive[o] = smg::translate_one(s8, L, T, o);
    return total;
}
