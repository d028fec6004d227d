// residues.hpp -- codon table and reduced amino-acid alphabets, shared by host and device.
// Reference: src/core/src/encodings.rs:103-347 (CODONTABLE with its third-position-N entries, DAYHOFFTABLE,
// HPTABLE; anything not in a table becomes 'X'), :85-93 (complement: bytes other than ACGTN become NUL).
#pragma once
#include <stdint.h>
#include "murmur3.hpp"   // SMG_HD

namespace smg {

SMG_HD uint8_t ascii_upper(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }

SMG_HD int nt_code(uint8_t c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : c == 'N' ? 4 : 5; }

// upper-case bases in, residue out
SMG_HD uint8_t translate_codon(uint8_t a, uint8_t b, uint8_t c) {
    // standard code, index = first * 16 + second * 4 + third with A C G T = 0 1 2 3
    const char* table = "KNKNTTTTRSRSIIMI" "QHQHPPPPRRRRLLLL" "EDEDAAAAGGGGVVVV" "*Y*YSSSS*CWCLFLF";
    const int x = nt_code(a), y = nt_code(b), z = nt_code(c);
    if (x > 3 || y > 3 || z > 4) return 'X';
    const char* row = table + x * 16 + y * 4;
    if (z == 4)   // ..N: the table only lists the four-fold degenerate families
        return (row[0] == row[1] && row[1] == row[2] && row[2] == row[3]) ? (uint8_t)row[0] : (uint8_t)'X';
    return (uint8_t)row[z];
}

SMG_HD uint8_t aa_to_dayhoff(uint8_t aa) {
    switch (aa) {
    case 'C': return 'a';
    case 'A': case 'G': case 'P': case 'S': case 'T': return 'b';
    case 'D': case 'E': case 'N': case 'Q': return 'c';
    case 'H': case 'K': case 'R': return 'd';
    case 'I': case 'L': case 'M': case 'V': return 'e';
    case 'F': case 'W': case 'Y': return 'f';
    case '*': return '*';
    default: return 'X';
    }
}

SMG_HD uint8_t aa_to_hp(uint8_t aa) {
    switch (aa) {
    case 'A': case 'F': case 'G': case 'I': case 'L': case 'M': case 'P': case 'V': case 'W': case 'Y': return 'h';
    case 'N': case 'C': case 'S': case 'T': case 'D': case 'E': case 'R': case 'H': case 'K': case 'Q': return 'p';
    case '*': return '*';
    default: return 'X';
    }
}

// hash_function: 2 protein, 3 dayhoff, 4 hp
SMG_HD uint8_t residue_encode(uint8_t aa, uint32_t hf) { return hf == 3 ? aa_to_dayhoff(aa) : hf == 4 ? aa_to_hp(aa) : aa; }

SMG_HD uint8_t dna_complement_or_nul(uint8_t c) {
    return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c == 'N' ? 'N' : 0;
}

}  // namespace smg
