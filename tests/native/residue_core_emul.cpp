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
