// translate_core.hpp -- the per-lane work of the six-frame translation kernel (protein.hip), host + device.
//
// Reference: src/core/src/signature.rs:307-393 (frames 0..2, forward strand then reverse complement per frame, to_aa drops a trailing
// partial codon) and src/core/src/encodings.rs:103-347 (codon table, alphabets).  Output layout (protein.hip): segment
// s = 2 * frame + strand holds (len - frame) / 3 residues followed by one 0xFF separator, segments back to back.
//
// A lane owns one aligned 4-byte WORD of the output: four consecutive residues of one segment.  Their 12 bases are contiguous in the
// input (descending on the reverse strand), at a byte offset from a 4-byte boundary that is the same for the whole segment -- so the
// lane loads the 16 aligned bytes around them, shifts them into place with a segment-uniform byte shift, and reads the bases at
// compile-time positions.  Words that straddle two segments, hold a separator, or whose 16 bytes would reach outside the input take
// the per-residue path (a handful per record).
#pragma once
#include <stdint.h>
#include "residues.hpp"

namespace smg {

struct TranslateTables {          // filled from the scalar functions of residues.hpp (they are the definition)
    const uint8_t* code_f;        // [256] byte -> nucleotide code 0..5 (A C G T N other)
    const uint8_t* code_r;        // [256] byte -> code of its complement
    const uint8_t* codon;         // [216] three codes -> residue of the sketch's alphabet
};

struct TranslateLayout {
    uint64_t len;
    uint64_t start[7];            // start[s]: first output byte of segment s; start[6] = total
};
SMG_HD TranslateLayout translate_layout(uint64_t len) {
    TranslateLayout L;
    L.len = len;
    L.start[0] = 0;
    for (int s = 0; s < 6; ++s) L.start[s + 1] = L.start[s] + (len - (uint64_t)(s >> 1)) / 3 + 1;
    return L;
}

// one output byte by the definition (the separator included)
SMG_HD uint8_t translate_one(const uint8_t* seq, const TranslateLayout& L, const TranslateTables& T, uint64_t o) {
    int s = 0;
    while (o >= L.start[s + 1]) ++s;
    const uint64_t i = o - L.start[s];
    if (i == L.start[s + 1] - L.start[s] - 1) return 0xff;
    const uint64_t p = (uint64_t)(s >> 1) + 3 * i;
    uint32_t x, y, z;
    if (s & 1) { x = T.code_r[seq[L.len - 1 - p]]; y = T.code_r[seq[L.len - 2 - p]]; z = T.code_r[seq[L.len - 3 - p]]; }
    else { x = T.code_f[seq[p]]; y = T.code_f[seq[p + 1]]; z = T.code_f[seq[p + 2]]; }
    return T.codon[x * 36 + y * 6 + z];
}

SMG_HD uint32_t byte_of(uint32_t w, int j) { return (w >> (8 * j)) & 0xffu; }

// output word G (bytes 4G .. 4G + 3).  seq32: the input as aligned 4-byte words (the caller guarantees 4-byte alignment of seq; only
// whole words inside [0, len) are read).  -> the four output bytes, little-endian
SMG_HD uint32_t translate_word(const uint8_t* seq, const uint32_t* seq32, const TranslateLayout& L, const TranslateTables& T, uint64_t G) {
    const uint64_t o0 = 4 * G;
    int s = 0;
    while (o0 >= L.start[s + 1]) ++s;
    const uint64_t seg_res = L.start[s + 1] - L.start[s] - 1;         // residues of the segment (its last byte is the separator)
    const uint64_t i0 = o0 - L.start[s];
    const uint64_t n_words = L.len / 4;                                // whole words inside the input: nothing past its end is read
    bool fast = i0 + 4 <= seg_res;                                    // four residues of this segment, no separator among them
    uint64_t first = 0;                                               // lowest input byte of the 12 bases
    if (fast) {
        const uint64_t p0 = (uint64_t)(s >> 1) + 3 * i0;
        first = (s & 1) ? L.len - 12 - p0 : p0;                       // reverse strand: bases len-1-p0 down to len-12-p0
        fast = (first >> 2) + 4 <= n_words;                           // the 16 aligned bytes lie inside the readable words
    }
    if (!fast) {
        uint32_t out = 0;
        for (int j = 0; j < 4; ++j)
            if (o0 + (uint64_t)j < L.start[6]) out |= (uint32_t)translate_one(seq, L, T, o0 + (uint64_t)j) << (8 * j);
        return out;
    }
    const uint64_t A = first >> 2;
    const uint32_t r = (uint32_t)(first & 3);                         // the same for every word of a segment
    const uint32_t w0 = seq32[A], w1 = seq32[A + 1], w2 = seq32[A + 2], w3 = seq32[A + 3];
    // D: the 12 bases as three words, ascending input order
    uint32_t d0, d1, d2;
    if (r == 0) { d0 = w0; d1 = w1; d2 = w2; }
    else {
        const uint32_t sh = 8 * r, ih = 32 - sh;
        d0 = (w0 >> sh) | (w1 << ih); d1 = (w1 >> sh) | (w2 << ih); d2 = (w2 >> sh) | (w3 << ih);
    }
    const uint32_t d[3] = {d0, d1, d2};
    uint32_t out = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t x, y, z;
        if (s & 1) {                                                  // residue q of the word: bases 11-3q, 10-3q, 9-3q, complemented
            const int a = 11 - 3 * q;
            x = T.code_r[byte_of(d[a >> 2], a & 3)];
            y = T.code_r[byte_of(d[(a - 1) >> 2], (a - 1) & 3)];
            z = T.code_r[byte_of(d[(a - 2) >> 2], (a - 2) & 3)];
        } else {
            const int a = 3 * q;
            x = T.code_f[byte_of(d[a >> 2], a & 3)];
            y = T.code_f[byte_of(d[(a + 1) >> 2], (a + 1) & 3)];
            z = T.code_f[byte_of(d[(a + 2) >> 2], (a + 2) & 3)];
        }
        out |= (uint32_t)T.codon[x * 36 + y * 6 + z] << (8 * q);
    }
    return out;
}

}  // namespace smg
