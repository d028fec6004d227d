/*
 * oracle.c -- CPU restatement of the sourmash FracMinHash hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under sourmash_amd/ may include, link,
 * import or call this file.  Allowed users: tests/, __graft_entry__.smoke()
 * and bench.py's `cpu_baseline` leg (as the checker / the timed CPU side,
 * never as the product).
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this file against the
 * reference's own known-answer tests and golden fixtures (SURVEY.md section
 * 8c): hash_murmur KATs, the E. coli k=21/31/51 scaled=1000 golden sketch
 * (md5 0a8632c67e6d88f737ddb510bef90337 for k=31), the 7x7 compare matrix,
 * the 12-round golden gather order, md5sum KATs.
 *
 * Every function cites the reference file:line (under /root/reference) whose
 * behaviour it restates.  The algorithm of the un-vendored dependency
 * `murmurhash3 = 0.0.5` (src/core/Cargo.toml:45, Cargo.lock:987-990) is the
 * public-domain MurmurHash3_x64_128 of Austin Appleby with a 64-bit seed used
 * for both h1 and h2; `md5 = 0.7.0` (Cargo.lock:931-934) is RFC 1321.
 *
 * Deliberately written the slow, obvious way the reference does it (upper-case
 * copy, full reverse complement, byte-wise min of forward/revcomp k-mer,
 * two-pointer merges) so that it is an independent statement of the algorithm
 * rather than a copy of the GPU formulation.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* MurmurHash3_x64_128, low 64 bits.                                         */
/* reference: src/core/src/lib.rs:57-59 (_hash_murmur = murmurhash3_x64_128  */
/* (kmer, seed).0); C export src/core/src/ffi/mod.rs:22-31.                  */
/* ------------------------------------------------------------------------ */
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

static inline uint64_t fmix64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

static inline uint64_t load_le64(const uint8_t *p) {
    uint64_t v = 0;
    for (int i = 7; i >= 0; --i) v = (v << 8) | p[i];
    return v;
}

ORC_API uint64_t orc_hash_murmur(const uint8_t *data, uint64_t len, uint64_t seed) {
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    uint64_t h1 = seed, h2 = seed;
    const uint64_t nblocks = len / 16;
    for (uint64_t i = 0; i < nblocks; ++i) {
        uint64_t k1 = load_le64(data + 16 * i);
        uint64_t k2 = load_le64(data + 16 * i + 8);
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    const uint8_t *tail = data + nblocks * 16;
    uint64_t k1 = 0, k2 = 0;
    switch (len & 15) {
    case 15: k2 ^= (uint64_t)tail[14] << 48; /* fallthrough */
    case 14: k2 ^= (uint64_t)tail[13] << 40; /* fallthrough */
    case 13: k2 ^= (uint64_t)tail[12] << 32; /* fallthrough */
    case 12: k2 ^= (uint64_t)tail[11] << 24; /* fallthrough */
    case 11: k2 ^= (uint64_t)tail[10] << 16; /* fallthrough */
    case 10: k2 ^= (uint64_t)tail[9] << 8;   /* fallthrough */
    case 9:  k2 ^= (uint64_t)tail[8];
             k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; /* fallthrough */
    case 8:  k1 ^= (uint64_t)tail[7] << 56; /* fallthrough */
    case 7:  k1 ^= (uint64_t)tail[6] << 48; /* fallthrough */
    case 6:  k1 ^= (uint64_t)tail[5] << 40; /* fallthrough */
    case 5:  k1 ^= (uint64_t)tail[4] << 32; /* fallthrough */
    case 4:  k1 ^= (uint64_t)tail[3] << 24; /* fallthrough */
    case 3:  k1 ^= (uint64_t)tail[2] << 16; /* fallthrough */
    case 2:  k1 ^= (uint64_t)tail[1] << 8;  /* fallthrough */
    case 1:  k1 ^= (uint64_t)tail[0];
             k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    }
    h1 ^= len; h2 ^= len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2;
    return h1;
}

/* ------------------------------------------------------------------------ */
/* max_hash <-> scaled.  reference: src/core/src/sketch/minhash.rs:21-34     */
/* (`u64::MAX as f64` rounds to 2^64; the quotient is truncated).            */
/* ------------------------------------------------------------------------ */
ORC_API uint64_t orc_max_hash_for_scaled(uint64_t scaled) {
    if (scaled == 0) return 0;
    if (scaled == 1) return UINT64_MAX;
    return (uint64_t)(18446744073709551616.0 / (double)scaled);
}

ORC_API uint64_t orc_scaled_for_max_hash(uint64_t max_hash) {
    if (max_hash == 0) return 0;
    double q = 18446744073709551616.0 / (double)max_hash;
    /* Rust `as u64` saturates */
    if (q >= 18446744073709551616.0) return UINT64_MAX;
    return (uint64_t)q;
}

/* ------------------------------------------------------------------------ */
/* DNA tables.  reference: src/core/src/encodings.rs:85-101 (COMPLEMENT,     */
/* revcomp), :370-377 (VALID = A,C,G,T only, tested after upper-casing).     */
/* ------------------------------------------------------------------------ */
static inline uint8_t dna_complement(uint8_t c) {
    switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    case 'N': return 'N';
    default:  return 0;
    }
}
static inline int dna_valid(uint8_t c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }
static inline uint8_t ascii_upper(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }

/* ------------------------------------------------------------------------ */
/* seq -> hashes, DNA branch.                                                */
/* reference: src/core/src/signature.rs:189-233 (SeqToHashes::new: upper-    */
/* case copy, max_index) and :246-306 (next(): full revcomp once, VALID scan,*/
/* force -> Ok(0) for a bad k-mer, else Err(InvalidDNA{kmer}); canonical =   */
/* lexicographic min of forward and revcomp ASCII k-mer; murmur with seed).  */
/*                                                                           */
/* out must hold len-k+1 entries.  Writes one entry per k-mer (0 for a bad   */
/* k-mer under force).  Returns the number of k-mers written; on a bad       */
/* k-mer with force==0 returns -1 - (index of the offending k-mer), having   */
/* written the hashes of all earlier k-mers (streaming semantics,            */
/* signature.rs:48-54).                                                      */
/* ------------------------------------------------------------------------ */
ORC_API int64_t orc_seq_to_hashes_dna(const uint8_t *seq, uint64_t len, uint32_t k,
                                      uint64_t seed, int force, uint64_t *out) {
    if (len < k || k == 0) return 0;               /* signature.rs:206-210,257-261 */
    uint8_t *up = (uint8_t *)malloc(len), *rc = (uint8_t *)malloc(len);
    for (uint64_t i = 0; i < len; ++i) up[i] = ascii_upper(seq[i]);          /* :214 */
    for (uint64_t i = 0; i < len; ++i) rc[i] = dna_complement(up[len - 1 - i]); /* :263 */
    const uint64_t n_kmers = len - k + 1;
    int64_t ret = (int64_t)n_kmers;
    for (uint64_t i = 0; i < n_kmers; ++i) {
        int bad = 0;
        for (uint32_t j = 0; j < k; ++j)
            if (!dna_valid(up[i + j])) { bad = 1; break; }                   /* :271-286 */
        if (bad) {
            if (!force) { ret = -1 - (int64_t)i; break; }
            out[i] = 0;
            continue;
        }
        const uint8_t *fwd = up + i;
        const uint8_t *rev = rc + (len - k - i);                              /* :300-301 */
        const uint8_t *canon = memcmp(fwd, rev, k) <= 0 ? fwd : rev;          /* :302-304 */
        out[i] = orc_hash_murmur(canon, k, seed);
    }
    free(up); free(rc);
    return ret;
}

/* ------------------------------------------------------------------------ */
/* seq -> hashes, protein / dayhoff / hp branches.                           */
/* reference: src/core/src/encodings.rs:103-368 (codon table incl. the       */
/* third-position-N entries, dayhoff and hp alphabets, unknown -> 'X',       */
/* to_aa drops a trailing partial codon), src/core/src/signature.rs:307-393  */
/* (translate: frames 0..2, forward then reverse complement per frame, every */
/* window hashed -- no validity test in this mode; protein input: windows of */
/* the upper-cased residues, mapped first for dayhoff / hp).                 */
/* hash_function: 2 protein, 3 dayhoff, 4 hp (include/sourmash.h:11-17).     */
/* ------------------------------------------------------------------------ */
static int nt_code(uint8_t c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : c == 'N' ? 4 : 5; }

ORC_API uint8_t orc_translate_codon(uint8_t a, uint8_t b, uint8_t c) {
    /* rows TCAG as in the standard table, written out by first/second base */
    static const char *TABLE[4][4] = {
        /* A */ {"KNKN", "TTTT", "RSRS", "IIMI"},   /* AA* AC* AG* AT*  (third base A C G T) */
        /* C */ {"QHQH", "PPPP", "RRRR", "LLLL"},
        /* G */ {"EDED", "AAAA", "GGGG", "VVVV"},
        /* T */ {"*Y*Y", "SSSS", "*CWC", "LFLF"},
    };
    const int x = nt_code(a), y = nt_code(b), z = nt_code(c);
    if (x > 3 || y > 3 || z > 4) return 'X';
    const char *row = TABLE[x][y];
    if (z == 4)                                   /* ..N: only the four-fold degenerate families are in the table */
        return (row[0] == row[1] && row[1] == row[2] && row[2] == row[3]) ? (uint8_t)row[0] : 'X';
    return (uint8_t)row[z];
}

ORC_API uint8_t orc_aa_to_dayhoff(uint8_t aa) {
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

ORC_API uint8_t orc_aa_to_hp(uint8_t aa) {
    switch (aa) {
    case 'A': case 'F': case 'G': case 'I': case 'L': case 'M': case 'P': case 'V': case 'W': case 'Y': return 'h';
    case 'N': case 'C': case 'S': case 'T': case 'D': case 'E': case 'R': case 'H': case 'K': case 'Q': return 'p';
    case '*': return '*';
    default: return 'X';
    }
}

static uint8_t aa_encode(uint8_t aa, uint32_t hf) { return hf == 3 ? orc_aa_to_dayhoff(aa) : hf == 4 ? orc_aa_to_hp(aa) : aa; }

/* Hashes of every residue k-mer, in the reference's order.  ksize is the STORED ksize (3 x residues).  out must
 * hold the return value of a call with out == NULL.  is_protein: seq holds residues; else DNA to translate. */
ORC_API uint64_t orc_seq_to_hashes_protein(const uint8_t *seq, uint64_t len, uint32_t ksize, uint32_t hf,
                                           uint64_t seed, int is_protein, uint64_t *out) {
    const uint64_t k = ksize / 3;                                  /* signature.rs:199-203 */
    uint64_t n = 0;
    if (k == 0 || len < k) return 0;
    if (is_protein) {
        uint8_t *aa = (uint8_t *)malloc(len);
        for (uint64_t i = 0; i < len; ++i) aa[i] = aa_encode(ascii_upper(seq[i]), hf);
        for (uint64_t i = 0; i + k <= len; ++i, ++n)
            if (out) out[n] = orc_hash_murmur(aa + i, k, seed);
        free(aa);
        return n;
    }
    if (len < 3 * k) return 0;                                     /* signature.rs:259-261 */
    uint8_t *up = (uint8_t *)malloc(len), *rc = (uint8_t *)malloc(len), *aa = (uint8_t *)malloc(len / 3 + 1);
    for (uint64_t i = 0; i < len; ++i) up[i] = ascii_upper(seq[i]);
    for (uint64_t i = 0; i < len; ++i) {                           /* encodings.rs:85-101: unknown bases complement to NUL */
        const uint8_t c = up[len - 1 - i];
        rc[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c == 'N' ? 'N' : 0;
    }
    for (int frame = 0; frame < 3; ++frame) {
        for (int strand = 0; strand < 2; ++strand) {
            const uint8_t *s = strand ? rc : up;
            uint64_t na = 0;
            for (uint64_t i = (uint64_t)frame; i + 3 <= len; i += 3)
                aa[na++] = aa_encode(orc_translate_codon(s[i], s[i + 1], s[i + 2]), hf);
            for (uint64_t i = 0; i + k <= na; ++i, ++n)
                if (out) out[n] = orc_hash_murmur(aa + i, k, seed);
        }
    }
    free(up); free(rc); free(aa);
    return n;
}

/* ------------------------------------------------------------------------ */
/* Sketch container (Vec-backed KmerMinHash semantics).                      */
/* reference: src/core/src/sketch/minhash.rs:36-913.                         */
/* ------------------------------------------------------------------------ */
typedef struct {
    uint32_t num;
    uint32_t ksize;
    uint32_t hash_function; /* 1 = DNA (include/sourmash.h:11-17) */
    uint64_t seed;
    uint64_t max_hash;
    int track_abundance;
    uint64_t *mins;
    uint64_t *abunds;
    uint64_t n, cap;
} orc_mh;

ORC_API orc_mh *orc_mh_new(uint64_t scaled, uint32_t ksize, uint32_t hash_function,
                           uint64_t seed, int track_abundance, uint32_t num) {
    /* minhash.rs:186-221 */
    orc_mh *m = (orc_mh *)calloc(1, sizeof(orc_mh));
    m->num = num; m->ksize = ksize; m->hash_function = hash_function; m->seed = seed;
    m->max_hash = orc_max_hash_for_scaled(scaled);
    m->track_abundance = track_abundance;
    m->cap = 1024;
    m->mins = (uint64_t *)malloc(m->cap * 8);
    m->abunds = (uint64_t *)malloc(m->cap * 8);
    return m;
}

ORC_API void orc_mh_free(orc_mh *m) {
    if (!m) return;
    free(m->mins); free(m->abunds); free(m);
}

ORC_API orc_mh *orc_mh_clone(const orc_mh *s) {
    orc_mh *m = (orc_mh *)malloc(sizeof(orc_mh));
    *m = *s;
    m->mins = (uint64_t *)malloc(m->cap * 8);
    m->abunds = (uint64_t *)malloc(m->cap * 8);
    memcpy(m->mins, s->mins, s->n * 8);
    memcpy(m->abunds, s->abunds, s->n * 8);
    return m;
}

ORC_API uint64_t orc_mh_size(const orc_mh *m) { return m->n; }
ORC_API const uint64_t *orc_mh_mins(const orc_mh *m) { return m->mins; }
ORC_API const uint64_t *orc_mh_abunds(const orc_mh *m) { return m->track_abundance ? m->abunds : NULL; }
ORC_API uint64_t orc_mh_max_hash(const orc_mh *m) { return m->max_hash; }
ORC_API void orc_mh_clear(orc_mh *m) { m->n = 0; }

static void mh_reserve(orc_mh *m, uint64_t want) {
    if (want <= m->cap) return;
    while (m->cap < want) m->cap *= 2;
    m->mins = (uint64_t *)realloc(m->mins, m->cap * 8);
    m->abunds = (uint64_t *)realloc(m->abunds, m->cap * 8);
}

/* lower bound: first index with mins[idx] >= h */
static uint64_t mh_lower_bound(const uint64_t *a, uint64_t n, uint64_t h) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (a[mid] < h) lo = mid + 1; else hi = mid;
    }
    return lo;
}

ORC_API void orc_mh_remove_hash(orc_mh *m, uint64_t h) {
    /* minhash.rs:406-416 */
    uint64_t pos = mh_lower_bound(m->mins, m->n, h);
    if (pos < m->n && m->mins[pos] == h) {
        memmove(m->mins + pos, m->mins + pos + 1, (m->n - pos - 1) * 8);
        memmove(m->abunds + pos, m->abunds + pos + 1, (m->n - pos - 1) * 8);
        m->n--;
    }
}

ORC_API void orc_mh_add_hash_with_abundance(orc_mh *m, uint64_t h, uint64_t abundance) {
    /* minhash.rs:313-383 */
    uint64_t current_max = m->n ? m->mins[m->n - 1] : UINT64_MAX;
    if (h > m->max_hash && m->max_hash != 0) return;          /* :319  keep h <= max_hash */
    if (m->num == 0 && m->max_hash == 0) return;              /* :324 */
    if (abundance == 0) { orc_mh_remove_hash(m, h); return; } /* :329-332 */
    if (m->n == 0) {                                          /* :337-344 */
        mh_reserve(m, 1);
        m->mins[0] = h; m->abunds[0] = abundance; m->n = 1;
        return;
    }
    if (h <= m->max_hash || h <= current_max || m->n < m->num) {  /* :346 */
        uint64_t pos = mh_lower_bound(m->mins, m->n, h);
        if (pos == m->n) {                                    /* :354-361 */
            mh_reserve(m, m->n + 1);
            m->mins[m->n] = h; m->abunds[m->n] = abundance; m->n++;
        } else if (m->mins[pos] != h) {                       /* :362-377 */
            mh_reserve(m, m->n + 1);
            memmove(m->mins + pos + 1, m->mins + pos, (m->n - pos) * 8);
            memmove(m->abunds + pos + 1, m->abunds + pos, (m->n - pos) * 8);
            m->mins[pos] = h; m->abunds[pos] = abundance; m->n++;
            if (m->num != 0 && m->n > m->num) m->n--;         /* pop the largest */
        } else if (m->track_abundance) {                      /* :378-381 */
            m->abunds[pos] += abundance;
        }
    }
}

ORC_API void orc_mh_add_hash(orc_mh *m, uint64_t h) { orc_mh_add_hash_with_abundance(m, h, 1); }

ORC_API void orc_mh_add_many(orc_mh *m, const uint64_t *hs, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) orc_mh_add_hash(m, hs[i]);  /* minhash.rs:525-530 */
}

ORC_API void orc_mh_remove_many(orc_mh *m, const uint64_t *hs, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) orc_mh_remove_hash(m, hs[i]); /* minhash.rs:425-430 */
}

/* add_sequence: SigsTrait default, src/core/src/signature.rs:38-58: walk the
 * k-mer hashes, skip hash == 0 (:50), add the rest; on error the earlier
 * hashes stay added.  Returns 0, or -1 - kmer_index on InvalidDNA. */
static void mh_add_protein_hashes(orc_mh *m, const uint8_t *seq, uint64_t len, int is_protein) {
    const uint64_t n = orc_seq_to_hashes_protein(seq, len, m->ksize, m->hash_function, m->seed, is_protein, NULL);
    if (!n) return;
    uint64_t *hs = (uint64_t *)malloc(n * 8);
    orc_seq_to_hashes_protein(seq, len, m->ksize, m->hash_function, m->seed, is_protein, hs);
    for (uint64_t i = 0; i < n; ++i)
        if (hs[i] != 0) orc_mh_add_hash(m, hs[i]);
    free(hs);
}

/* signature.rs:60-80 */
ORC_API void orc_mh_add_protein(orc_mh *m, const uint8_t *seq, uint64_t len) { mh_add_protein_hashes(m, seq, len, 1); }

ORC_API int64_t orc_mh_add_sequence(orc_mh *m, const uint8_t *seq, uint64_t len, int force) {
    if (m->hash_function != 1) { mh_add_protein_hashes(m, seq, len, 0); return 0; }   /* translate, no validity test */
    if (len < m->ksize) return 0;
    uint64_t nk = len - m->ksize + 1;
    uint64_t *hs = (uint64_t *)malloc(nk * 8);
    int64_t r = orc_seq_to_hashes_dna(seq, len, m->ksize, m->seed, force, hs);
    uint64_t upto = r >= 0 ? (uint64_t)r : (uint64_t)(-1 - r);
    for (uint64_t i = 0; i < upto; ++i)
        if (hs[i] != 0) orc_mh_add_hash(m, hs[i]);
    free(hs);
    return r >= 0 ? 0 : r;
}

/* check_compatible, minhash.rs:886-912.  0 = ok; else the SourmashErrorCode
 * (include/sourmash.h:25-28): order ksize, hash_function, max_hash, seed. */
ORC_API uint32_t orc_mh_check_compatible(const orc_mh *a, const orc_mh *b) {
    if (a->ksize != b->ksize) return 101;
    if (a->hash_function != b->hash_function) return 102;
    if (a->max_hash != b->max_hash) return 103;
    if (a->seed != b->seed) return 104;
    return 0;
}

/* merge, minhash.rs:432-516: sorted union, abundances summed on equal keys,
 * truncated to num if num != 0. */
ORC_API uint32_t orc_mh_merge(orc_mh *a, const orc_mh *b) {
    uint32_t e = orc_mh_check_compatible(a, b);
    if (e) return e;
    uint64_t cap = a->n + b->n + 1;
    uint64_t *mm = (uint64_t *)malloc(cap * 8), *ma = (uint64_t *)malloc(cap * 8);
    uint64_t i = 0, j = 0, n = 0;
    while (i < a->n && j < b->n) {
        if (a->mins[i] < b->mins[j]) { mm[n] = a->mins[i]; ma[n++] = a->abunds[i++]; }
        else if (b->mins[j] < a->mins[i]) { mm[n] = b->mins[j]; ma[n++] = b->abunds[j++]; }
        else { mm[n] = a->mins[i]; ma[n++] = a->abunds[i++] + b->abunds[j++]; }
    }
    while (i < a->n) { mm[n] = a->mins[i]; ma[n++] = a->abunds[i++]; }
    while (j < b->n) { mm[n] = b->mins[j]; ma[n++] = b->abunds[j++]; }
    if (a->num != 0 && n > a->num) n = a->num;
    free(a->mins); free(a->abunds);
    a->mins = mm; a->abunds = ma; a->n = n; a->cap = cap;
    /* merged abundances exist only if both track (minhash.rs:437-442) */
    if (!(a->track_abundance && b->track_abundance)) a->track_abundance = 0;
    return 0;
}

/* the two-pointer walk, minhash.rs:915-953 (Intersection) / :1765-1807
 * (intersection_size): returns common, writes union size. */
ORC_API uint64_t orc_intersection_size(const uint64_t *a, uint64_t na, const uint64_t *b,
                                       uint64_t nb, uint64_t *union_size) {
    /* the same walk without data-dependent branches (a[i] < b[j]: ++i; b[j] < a[i]: ++j; equal: both and ++common) --
     * the full-size parity runs of tests/test_gpu_full_configs.py take 10^12 of these steps */
    uint64_t i = 0, j = 0, common = 0, uni = 0;
    while (i < na && j < nb) {
        const uint64_t x = a[i], y = b[j];
        common += (x == y);
        i += (x <= y);
        j += (y <= x);
        ++uni;
    }
    uni += (na - i) + (nb - j);
    if (union_size) *union_size = uni;
    return common;
}

/* Four of the walks above at once, one list `a` against four lists b[0..3]: the same steps in the same order for
 * every pair, interleaved only so that an out-of-order core overlaps the four load->compare->advance chains (the
 * full-size parity runs take 10^12 steps).  common[t] = |a ∩ b[t]|. */
static void intersection_size_x4(const uint64_t *a, uint64_t na, const uint64_t *const b[4], const uint64_t nb[4],
                                 uint64_t common[4]) {
    uint64_t i0 = 0, i1 = 0, i2 = 0, i3 = 0, j0 = 0, j1 = 0, j2 = 0, j3 = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    const uint64_t *b0 = b[0], *b1 = b[1], *b2 = b[2], *b3 = b[3];
    while (i0 < na && j0 < nb[0] && i1 < na && j1 < nb[1] && i2 < na && j2 < nb[2] && i3 < na && j3 < nb[3]) {
        const uint64_t x0 = a[i0], y0 = b0[j0], x1 = a[i1], y1 = b1[j1], x2 = a[i2], y2 = b2[j2], x3 = a[i3], y3 = b3[j3];
        c0 += (x0 == y0); i0 += (x0 <= y0); j0 += (y0 <= x0);
        c1 += (x1 == y1); i1 += (x1 <= y1); j1 += (y1 <= x1);
        c2 += (x2 == y2); i2 += (x2 <= y2); j2 += (y2 <= x2);
        c3 += (x3 == y3); i3 += (x3 <= y3); j3 += (y3 <= x3);
    }
    common[0] = c0 + orc_intersection_size(a + i0, na - i0, b0 + j0, nb[0] - j0, NULL);
    common[1] = c1 + orc_intersection_size(a + i1, na - i1, b1 + j1, nb[1] - j1, NULL);
    common[2] = c2 + orc_intersection_size(a + i2, na - i2, b2 + j2, nb[2] - j2, NULL);
    common[3] = c3 + orc_intersection_size(a + i3, na - i3, b3 + j3, nb[3] - j3, NULL);
}

/* intersection list, minhash.rs:1721-1763 */
ORC_API uint64_t orc_intersection(const uint64_t *a, uint64_t na, const uint64_t *b,
                                  uint64_t nb, uint64_t *out) {
    uint64_t i = 0, j = 0, n = 0;
    while (i < na && j < nb) {
        if (a[i] < b[j]) ++i;
        else if (b[j] < a[i]) ++j;
        else { out[n++] = a[i]; ++i; ++j; }
    }
    return n;
}

/* downsample_scaled, minhash.rs:777-798: new sketch at the coarser scaled,
 * re-adding every hash (for a sorted set: the prefix <= new max_hash).
 * Returns NULL with *err = 109 (CannotUpsampleScaled) on upsample. */
ORC_API orc_mh *orc_mh_downsample_scaled(const orc_mh *m, uint64_t scaled, uint32_t *err) {
    *err = 0;
    uint64_t cur = orc_scaled_for_max_hash(m->max_hash);
    if (cur == scaled || cur == 0) return orc_mh_clone(m);
    if (cur > scaled) { *err = 109; return NULL; }
    orc_mh *n = orc_mh_new(scaled, m->ksize, m->hash_function, m->seed, m->track_abundance, m->num);
    for (uint64_t i = 0; i < m->n; ++i)
        orc_mh_add_hash_with_abundance(n, m->mins[i], m->track_abundance ? m->abunds[i] : 1);
    return n;
}

/* count_common, minhash.rs:539-558: optional downsample of the finer sketch
 * to the coarser scaled; check_compatible; count of the merge walk. */
ORC_API uint64_t orc_mh_count_common(const orc_mh *a, const orc_mh *b, int downsample, uint32_t *err) {
    *err = 0;
    uint64_t sa = orc_scaled_for_max_hash(a->max_hash), sb = orc_scaled_for_max_hash(b->max_hash);
    if (downsample && sa != sb) {
        const orc_mh *first = sa > sb ? a : b, *second = sa > sb ? b : a;
        orc_mh *d = orc_mh_downsample_scaled(second, orc_scaled_for_max_hash(first->max_hash), err);
        if (!d) return 0;
        uint64_t r = orc_mh_count_common(first, d, 0, err);
        orc_mh_free(d);
        return r;
    }
    *err = orc_mh_check_compatible(a, b);
    if (*err) return 0;
    return orc_intersection_size(a->mins, a->n, b->mins, b->n, NULL);
}

/* intersection_size incl. the num (bottom-k) rule, minhash.rs:593-621:
 * for num sketches intersect (A∩B) with the merged-and-truncated union. */
ORC_API uint64_t orc_mh_intersection_size(const orc_mh *a, const orc_mh *b, uint64_t *union_size, uint32_t *err) {
    *err = orc_mh_check_compatible(a, b);
    if (*err) { *union_size = 0; return 0; }
    if (a->num != 0) {
        orc_mh *c = orc_mh_new(orc_scaled_for_max_hash(a->max_hash), a->ksize, a->hash_function,
                               a->seed, a->track_abundance, a->num);
        orc_mh_merge(c, a);
        orc_mh_merge(c, b);
        uint64_t cap = (a->n < b->n ? a->n : b->n) + 1;
        uint64_t *i1 = (uint64_t *)malloc(cap * 8);
        uint64_t n1 = orc_intersection(a->mins, a->n, b->mins, b->n, i1);
        uint64_t common = orc_intersection_size(i1, n1, c->mins, c->n, NULL);
        *union_size = c->n;
        free(i1); orc_mh_free(c);
        return common;
    }
    return orc_intersection_size(a->mins, a->n, b->mins, b->n, union_size);
}

/* jaccard, minhash.rs:624-631: common / max(1, union) as f64 */
ORC_API double orc_mh_jaccard(const orc_mh *a, const orc_mh *b, uint32_t *err) {
    uint64_t uni = 0;
    uint64_t common = orc_mh_intersection_size(a, b, &uni, err);
    if (*err) return 0.0;
    return (double)common / (double)(uni > 1 ? uni : 1);
}

/* angular_similarity, minhash.rs:635-680 */
ORC_API double orc_mh_angular_similarity(const orc_mh *a, const orc_mh *b, uint32_t *err) {
    *err = orc_mh_check_compatible(a, b);
    if (*err) return 0.0;
    if (!a->track_abundance || !b->track_abundance) { *err = 108; return 0.0; }
    uint64_t a_sq = 0, b_sq = 0, prod = 0;
    for (uint64_t i = 0; i < a->n; ++i) a_sq += a->abunds[i] * a->abunds[i];
    for (uint64_t i = 0; i < b->n; ++i) b_sq += b->abunds[i] * b->abunds[i];
    uint64_t i = 0, j = 0;
    while (i < a->n && j < b->n) {
        if (a->mins[i] < b->mins[j]) ++i;
        else if (b->mins[j] < a->mins[i]) ++j;
        else { prod += a->abunds[i] * b->abunds[j]; ++i; ++j; }
    }
    double na = sqrt((double)a_sq), nb = sqrt((double)b_sq);
    if (na == 0.0 || nb == 0.0) return 0.0;
    double p = (double)prod / (na * nb);
    if (p > 1.0) p = 1.0;
    return 1.0 - 2.0 * acos(p) / 3.14159265358979323846264338327950288;
}

/* similarity, minhash.rs:682-702 */
ORC_API double orc_mh_similarity(const orc_mh *a, const orc_mh *b, int ignore_abundance,
                                 int downsample, uint32_t *err) {
    *err = 0;
    uint64_t sa = orc_scaled_for_max_hash(a->max_hash), sb = orc_scaled_for_max_hash(b->max_hash);
    if (downsample && sa != sb) {
        const orc_mh *first = sa > sb ? a : b, *second = sa > sb ? b : a;
        orc_mh *d = orc_mh_downsample_scaled(second, orc_scaled_for_max_hash(first->max_hash), err);
        if (!d) return 0.0;
        double r = orc_mh_similarity(first, d, ignore_abundance, 0, err);
        orc_mh_free(d);
        return r;
    }
    if (ignore_abundance || !a->track_abundance || !b->track_abundance)
        return orc_mh_jaccard(a, b, err);
    return orc_mh_angular_similarity(a, b, err);
}

/* ------------------------------------------------------------------------ */
/* md5 (RFC 1321) + md5sum of a sketch.                                      */
/* reference: src/core/src/sketch/minhash.rs:290-307: md5 over the decimal   */
/* ASCII of ksize followed by the decimal ASCII of every hash, no separators.*/
/* ------------------------------------------------------------------------ */
typedef struct { uint32_t s[4]; uint64_t nbytes; uint8_t buf[64]; uint32_t fill; } md5_ctx;

static const uint32_t MD5_K[64] = {
    0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501,
    0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
    0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
    0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
    0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
    0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
    0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1,
    0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
static const uint8_t MD5_R[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22,
                                  5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20,
                                  4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                                  6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};

static void md5_block(md5_ctx *c, const uint8_t *p) {
    uint32_t w[16];
    for (int i = 0; i < 16; ++i)
        w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) |
               ((uint32_t)p[4 * i + 3] << 24);
    uint32_t a = c->s[0], b = c->s[1], cc = c->s[2], d = c->s[3];
    for (int i = 0; i < 64; ++i) {
        uint32_t f; int g;
        if (i < 16) { f = (b & cc) | (~b & d); g = i; }
        else if (i < 32) { f = (d & b) | (~d & cc); g = (5 * i + 1) & 15; }
        else if (i < 48) { f = b ^ cc ^ d; g = (3 * i + 5) & 15; }
        else { f = cc ^ (b | ~d); g = (7 * i) & 15; }
        uint32_t t = a + f + MD5_K[i] + w[g];
        a = d; d = cc; cc = b;
        b = b + ((t << MD5_R[i]) | (t >> (32 - MD5_R[i])));
    }
    c->s[0] += a; c->s[1] += b; c->s[2] += cc; c->s[3] += d;
}
static void md5_init(md5_ctx *c) {
    c->s[0] = 0x67452301; c->s[1] = 0xefcdab89; c->s[2] = 0x98badcfe; c->s[3] = 0x10325476;
    c->nbytes = 0; c->fill = 0;
}
static void md5_update(md5_ctx *c, const uint8_t *p, uint64_t n) {
    c->nbytes += n;
    while (n) {
        uint32_t take = 64 - c->fill; if (take > n) take = (uint32_t)n;
        memcpy(c->buf + c->fill, p, take);
        c->fill += take; p += take; n -= take;
        if (c->fill == 64) { md5_block(c, c->buf); c->fill = 0; }
    }
}
static void md5_final(md5_ctx *c, uint8_t out[16]) {
    uint64_t bits = c->nbytes * 8;
    uint8_t pad = 0x80;
    md5_update(c, &pad, 1);
    uint8_t z = 0;
    while (c->fill != 56) md5_update(c, &z, 1);
    uint8_t lenb[8];
    for (int i = 0; i < 8; ++i) lenb[i] = (uint8_t)(bits >> (8 * i));
    md5_update(c, lenb, 8);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(c->s[i] >> (8 * j));
}

ORC_API void orc_md5_hex(const uint8_t *data, uint64_t n, char out[33]) {
    md5_ctx c; uint8_t d[16];
    md5_init(&c); md5_update(&c, data, n); md5_final(&c, d);
    for (int i = 0; i < 16; ++i) sprintf(out + 2 * i, "%02x", d[i]);
    out[32] = 0;
}

ORC_API void orc_md5sum_hashes(uint32_t ksize, const uint64_t *mins, uint64_t n, char out[33]) {
    md5_ctx c; uint8_t d[16]; char buf[32];
    md5_init(&c);
    int l = sprintf(buf, "%u", ksize);
    md5_update(&c, (const uint8_t *)buf, (uint64_t)l);
    for (uint64_t i = 0; i < n; ++i) {
        l = sprintf(buf, "%llu", (unsigned long long)mins[i]);
        md5_update(&c, (const uint8_t *)buf, (uint64_t)l);
    }
    md5_final(&c, d);
    for (int i = 0; i < 16; ++i) sprintf(out + 2 * i, "%02x", d[i]);
    out[32] = 0;
}

ORC_API void orc_mh_md5sum(const orc_mh *m, char out[33]) { orc_md5sum_hashes(m->ksize, m->mins, m->n, out); }

/* ------------------------------------------------------------------------ */
/* Bulk DNA sketching for the CPU baseline and for large parity checks:      */
/* every byte outside ACGTacgt separates records (force=True semantics:      */
/* every k-mer covering it is dropped, signature.rs:271-286), the kept set   */
/* is { h : h != 0 && h <= max_hash }, returned sorted and unique            */
/* (add_hash into a set, minhash.rs:313-383 with num == 0).                  */
/* Same per-k-mer work as orc_seq_to_hashes_dna, run over `nthreads` OpenMP  */
/* slices with a k-1 halo.  Returns the number of unique hashes written to   */
/* *out (malloc'ed; free with orc_free).                                     */
/* ------------------------------------------------------------------------ */
static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

typedef struct { uint64_t *v; uint64_t n, cap; } u64vec;
static void u64vec_push(u64vec *v, uint64_t x) {
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->v = (uint64_t *)realloc(v->v, v->cap * 8); }
    v->v[v->n++] = x;
}

static void sketch_slice(const uint8_t *seq, uint64_t len, uint64_t kmer_lo, uint64_t kmer_hi,
                         uint32_t k, uint64_t seed, uint64_t max_hash, u64vec *out) {
    /* k-mers with start index in [kmer_lo, kmer_hi) */
    uint8_t *fwd = (uint8_t *)malloc((size_t)k * 2), *rev = fwd + k;       /* any k: the reference has no limit */
    (void)len;
    for (uint64_t i = kmer_lo; i < kmer_hi; ++i) {
        int bad = 0;
        for (uint32_t j = 0; j < k; ++j) {
            uint8_t c = ascii_upper(seq[i + j]);
            if (!dna_valid(c)) { bad = 1; break; }
            fwd[j] = c;
            rev[k - 1 - j] = dna_complement(c);
        }
        if (bad) continue;
        const uint8_t *canon = memcmp(fwd, rev, k) <= 0 ? fwd : rev;
        uint64_t h = orc_hash_murmur(canon, k, seed);
        if (h != 0 && (max_hash == 0 || h <= max_hash)) u64vec_push(out, h);
    }
    free(fwd);
}

ORC_API uint64_t orc_sketch_dna_bulk(const uint8_t *seq, uint64_t len, uint32_t k, uint64_t seed,
                                     uint64_t max_hash, int nthreads, uint64_t **out) {
    *out = NULL;
    if (len < k || k == 0) return 0;
    uint64_t nk = len - k + 1;
    if (nthreads < 1) nthreads = 1;
    u64vec *parts = (u64vec *)calloc((size_t)nthreads, sizeof(u64vec));
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
#endif
    for (int t = 0; t < nthreads; ++t) {
        uint64_t lo = nk * (uint64_t)t / (uint64_t)nthreads, hi = nk * (uint64_t)(t + 1) / (uint64_t)nthreads;
        sketch_slice(seq, len, lo, hi, k, seed, max_hash, &parts[t]);
    }
    uint64_t total = 0;
    for (int t = 0; t < nthreads; ++t) total += parts[t].n;
    uint64_t *all = (uint64_t *)malloc((total ? total : 1) * 8);
    uint64_t n = 0;
    for (int t = 0; t < nthreads; ++t) { memcpy(all + n, parts[t].v, parts[t].n * 8); n += parts[t].n; free(parts[t].v); }
    free(parts);
    qsort(all, n, 8, cmp_u64);
    uint64_t u = 0;
    for (uint64_t i = 0; i < n; ++i)
        if (i == 0 || all[i] != all[i - 1]) all[u++] = all[i];
    *out = all;
    return u;
}

ORC_API void orc_free(void *p) { free(p); }

/* ------------------------------------------------------------------------ */
/* Synthetic inputs shared by tests/bench.            */
/* splitmix64 is the public-domain generator of Steele/Lea/Flood (Vigna's    */
/* reference constants).                                                     */
/* ------------------------------------------------------------------------ */
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
ORC_API uint64_t orc_splitmix64(uint64_t x) { return splitmix64(x); }

/* C2 generator: base i (global index) = "ACGT"[(splitmix64(seed ^ (i/32)) >> (2*(i%32))) & 3];
 * every position p with (p + 1) % (record_len + 1) == 0 holds the record
 * separator '\n' instead (so records are record_len bases long). record_len==0
 * means no separators. */
ORC_API void orc_synth_dna(uint8_t *out, uint64_t start, uint64_t n, uint64_t seed, uint64_t record_len) {
    static const char ACGT[4] = {'A', 'C', 'G', 'T'};
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t p = start + i;
        if (record_len && (p + 1) % (record_len + 1) == 0) { out[i] = '\n'; continue; }
        uint64_t w = splitmix64(seed ^ (p >> 5));
        out[i] = (uint8_t)ACGT[(w >> (2 * (p & 31))) & 3];
    }
}

/* ------------------------------------------------------------------------ */
/* All-pairs compare on a CSR of sorted sketches.                            */
/* reference: src/sourmash/compare.py:14-64 (compare_serial: ones on the     */
/* diagonal, similarity(i,j) for every i<j, symmetric fill) with             */
/* similarity = jaccard = common / max(1, union) (minhash.rs:624-631).       */
/* Writes u32 common[n][n] (diagonal = row size) and f64 jaccard[n][n].      */
/* ------------------------------------------------------------------------ */
ORC_API void orc_compare_all_pairs(const uint64_t *hashes, const uint64_t *offsets, uint64_t n,
                                   uint32_t *common, double *jaccard, int nthreads) {
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
#endif
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t ni = offsets[i + 1] - offsets[i];
        if (common) common[i * n + i] = (uint32_t)ni;
        if (jaccard) jaccard[i * n + i] = 1.0;               /* compare.py:33 np.ones */
        uint64_t j = i + 1;
        for (; j + 4 <= n; j += 4) {                          /* four columns at a time (see intersection_size_x4) */
            const uint64_t *b[4]; uint64_t nb[4], c4[4];
            for (int t = 0; t < 4; ++t) { b[t] = hashes + offsets[j + t]; nb[t] = offsets[j + t + 1] - offsets[j + t]; }
            intersection_size_x4(hashes + offsets[i], ni, b, nb, c4);
            for (int t = 0; t < 4; ++t) {
                /* the walk's union count is every step + both tails = ni + nj - common (sorted unique lists) */
                uint64_t c = c4[t], uni = ni + nb[t] - c;
                if (common) { common[i * n + j + t] = (uint32_t)c; common[(j + t) * n + i] = (uint32_t)c; }
                if (jaccard) {
                    double s = (double)c / (double)(uni > 1 ? uni : 1);
                    jaccard[i * n + j + t] = s; jaccard[(j + t) * n + i] = s;
                }
            }
        }
        for (; j < n; ++j) {
            uint64_t nj = offsets[j + 1] - offsets[j], uni = 0;
            uint64_t c = orc_intersection_size(hashes + offsets[i], ni, hashes + offsets[j], nj, &uni);
            if (common) { common[i * n + j] = (uint32_t)c; common[j * n + i] = (uint32_t)c; }
            if (jaccard) {
                double s = (double)c / (double)(uni > 1 ? uni : 1);
                jaccard[i * n + j] = s; jaccard[j * n + i] = s;
            }
        }
    }
}

/* ------------------------------------------------------------------------ */
/* Greedy gather (min-set-cover) on a CSR database.                          */
/* reference: src/sourmash/index/__init__.py:735-909 (CounterGather: add ->  */
/* overlap = |Q ∩ D|, :783-789; peek -> most_common()[0], ties to the        */
/* first-inserted = lowest index, :856-857; stop when best < threshold,      */
/* :860-861; intersect_mh = cur_query ∩ match, :875-876; consume -> every    */
/* counter -= |intersect ∩ D_d|, :897-909) and src/sourmash/search.py:       */
/* 877-949 (GatherDatabases.__next__: query <- query minus the whole match   */
/* sketch, :915-919); threshold src/sourmash/search.py:15-37.                */
/*                                                                           */
/* Inputs: sorted query, CSR db (all at the query's scaled), threshold_bp.    */
/* Outputs: per round the dataset index and |intersect|.  Returns rounds.    */
/* ------------------------------------------------------------------------ */
ORC_API uint64_t orc_gather_mt(const uint64_t *query, uint64_t nq, const uint64_t *hashes,
                               const uint64_t *offsets, uint64_t ndb, uint64_t threshold_bp,
                               uint64_t scaled, uint64_t *out_idx, uint64_t *out_isect,
                               uint64_t max_rounds, int nthreads) {
    /* nthreads > 1: the two loops over the datasets (prefetch, consume) and the arg-max run on OpenMP threads.  Every
     * dataset's update is independent and the arg-max keeps the reference's rule (largest counter, first-inserted =
     * lowest index on ties), so the result does not depend on the thread count. */
    if (nthreads < 1) nthreads = 1;
    uint64_t *q = (uint64_t *)malloc((nq ? nq : 1) * 8);
    memcpy(q, query, nq * 8);
    uint64_t *counter = (uint64_t *)malloc((ndb ? ndb : 1) * 8);
    uint64_t *isect = (uint64_t *)malloc((nq ? nq : 1) * 8);
    /* prefetch: overlap of every dataset with the original query; datasets with
     * zero overlap never enter the counter (index/__init__.py:783-789). */
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
    for (uint64_t g = 0; g < (ndb + 3) / 4; ++g) {
        const uint64_t d0 = g * 4;
        if (d0 + 4 <= ndb) {
            const uint64_t *b[4]; uint64_t nb[4];
            for (int t = 0; t < 4; ++t) { b[t] = hashes + offsets[d0 + t]; nb[t] = offsets[d0 + t + 1] - offsets[d0 + t]; }
            intersection_size_x4(q, nq, b, nb, counter + d0);
        } else {
            for (uint64_t d = d0; d < ndb; ++d)
                counter[d] = orc_intersection_size(q, nq, hashes + offsets[d], offsets[d + 1] - offsets[d], NULL);
        }
    }
    uint64_t rounds = 0;
    while (rounds < max_rounds) {
        if (nq == 0) break;                                   /* search.py:879-880; index/__init__.py:838-839 */
        /* calc_threshold_from_bp, search.py:15-37 (float arithmetic as in Python) */
        double n_threshold_hashes = 0.0;
        if (threshold_bp) {
            n_threshold_hashes = (double)threshold_bp / (double)scaled;
            if (n_threshold_hashes / (double)nq > 1.0) break; /* unattainable -> [] */
        }
        uint64_t best = 0, best_d = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
        {
            uint64_t my_best = 0, my_d = 0;
#ifdef _OPENMP
#pragma omp for schedule(static) nowait
#endif
            for (uint64_t d = 0; d < ndb; ++d)
                if (counter[d] > my_best) { my_best = counter[d]; my_d = d; }   /* strict > : first-inserted wins ties */
#ifdef _OPENMP
#pragma omp critical
#endif
            if (my_best > best || (my_best == best && my_best != 0 && my_d < best_d)) { best = my_best; best_d = my_d; }
        }
        if (best == 0) break;                                           /* empty counter */
        if ((double)best < n_threshold_hashes) break;                   /* index/__init__.py:860-861 */
        const uint64_t *m = hashes + offsets[best_d];
        uint64_t nm = offsets[best_d + 1] - offsets[best_d];
        uint64_t ni = orc_intersection(q, nq, m, nm, isect);            /* :875-876 */
        out_idx[rounds] = best_d; out_isect[rounds] = ni;
        ++rounds;
        /* consume (:897-909): every live counter -= |intersect ∩ D_d|; zero -> deleted */
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
        for (uint64_t g = 0; g < (ndb + 3) / 4; ++g) {
            const uint64_t d0 = g * 4;
            if (d0 + 4 <= ndb && counter[d0] && counter[d0 + 1] && counter[d0 + 2] && counter[d0 + 3]) {
                const uint64_t *b[4]; uint64_t nb[4], c4[4];
                for (int t = 0; t < 4; ++t) { b[t] = hashes + offsets[d0 + t]; nb[t] = offsets[d0 + t + 1] - offsets[d0 + t]; }
                intersection_size_x4(isect, ni, b, nb, c4);
                for (int t = 0; t < 4; ++t) counter[d0 + t] = c4[t] >= counter[d0 + t] ? 0 : counter[d0 + t] - c4[t];
                continue;
            }
            for (uint64_t d = d0; d < d0 + 4 && d < ndb; ++d) {
                if (counter[d] == 0) continue;
                uint64_t c = orc_intersection_size(isect, ni, hashes + offsets[d], offsets[d + 1] - offsets[d], NULL);
                counter[d] = c >= counter[d] ? 0 : counter[d] - c;
            }
        }
        /* query <- query minus the whole match sketch (search.py:915-919) */
        uint64_t i = 0, j = 0, w = 0;
        while (i < nq) {
            while (j < nm && m[j] < q[i]) ++j;
            if (j < nm && m[j] == q[i]) { ++i; continue; }
            q[w++] = q[i++];
        }
        nq = w;
    }
    free(q); free(counter); free(isect);
    return rounds;
}

ORC_API uint64_t orc_gather(const uint64_t *query, uint64_t nq, const uint64_t *hashes,
                            const uint64_t *offsets, uint64_t ndb, uint64_t threshold_bp,
                            uint64_t scaled, uint64_t *out_idx, uint64_t *out_isect,
                            uint64_t max_rounds) {
    return orc_gather_mt(query, nq, hashes, offsets, ndb, threshold_bp, scaled, out_idx, out_isect, max_rounds, 1);
}

/* ------------------------------------------------------------------------ */
/* Host float layer of the containment family (test infrastructure like the  */
/* rest of this file): scalar restatements, one pair at a time, using this   */
/* libm's pow() -- the function CPython's float `**` ends in.                */
/* ------------------------------------------------------------------------ */

/* src/sourmash/minhash.py:819-841 MinHash.contained_by: count_common / (denom * bias_factor) with
 * bias_factor = 1 - (1 - 1/scaled)^(denom * scaled), clamped to [0, 1]; 0.0 for an empty self. */
ORC_API double orc_contained_by(uint64_t common, uint64_t denom, uint64_t scaled) {
    if (denom == 0) return 0.0;
    double total_denom = (double)(denom * scaled);
    double bias_factor = 1.0 - pow(1.0 - 1.0 / (double)scaled, total_denom);
    double containment = (double)common / ((double)denom * bias_factor);
    if (containment >= 1.0) return 1.0;
    if (containment <= 0.0) return 0.0;
    return containment;
}

/* src/sourmash/minhash.py:881-905 MinHash.max_containment: the same with denom = min(|self|, |other|) */
ORC_API double orc_max_containment(uint64_t common, uint64_t n_self, uint64_t n_other, uint64_t scaled) {
    return orc_contained_by(common, n_self < n_other ? n_self : n_other, scaled);
}

/* src/sourmash/minhash.py:946-959 MinHash.avg_containment: mean of the two directed containments */
ORC_API double orc_avg_containment(uint64_t common, uint64_t n_self, uint64_t n_other, uint64_t scaled) {
    double c1 = orc_contained_by(common, n_self, scaled), c2 = orc_contained_by(common, n_other, scaled);
    return (c1 + c2) / 2;
}

/* src/sourmash/distance_utils.py:276-283 containment_to_distance, point estimate only:
 * 1.0 for containment 0, 0.0 for containment 1, else 1 - containment^(1/ksize).
 * ANI = 1 - distance (distance_utils.py ANIResult.ani). */
ORC_API double orc_containment_to_distance_point(double containment, uint32_t ksize) {
    if (containment == 0.0) return 1.0;
    if (containment == 1.0) return 0.0;
    return 1.0 - pow(containment, 1.0 / (double)ksize);
}

/* ------------------------------------------------------------------------ */
/* compare_serial over sketch OBJECTS (bottom-k sketches, abundance-tracking */
/* sketches, mixed scaled values): src/sourmash/compare.py:14-64 -- ones on  */
/* the diagonal, out[i][j] = out[j][i] = siglist[i].similarity(siglist[j],   */
/* ignore_abundance, downsample) for every i < j -- with similarity =         */
/* orc_mh_similarity above (minhash.rs:682-702: downsample the finer sketch,  */
/* then jaccard incl. the num rule of :593-621, or angular :635-680).         */
/* *err = code of the first failing pair in the loop's order (0: none).       */
/* ------------------------------------------------------------------------ */
ORC_API void orc_similarity_matrix(const orc_mh *const *mhs, uint64_t n, int ignore_abundance, int downsample,
                                   double *out, uint32_t *err, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    *err = 0;
    uint32_t *row_err = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
#endif
    for (uint64_t i = 0; i < n; ++i) {
        out[i * n + i] = 1.0;
        for (uint64_t j = i + 1; j < n; ++j) {
            uint32_t e = 0;
            double s = orc_mh_similarity(mhs[i], mhs[j], ignore_abundance, downsample, &e);
            if (e && !row_err[i]) row_err[i] = e;
            out[i * n + j] = s; out[j * n + i] = s;
        }
    }
    for (uint64_t i = 0; i < n && !*err; ++i) *err = row_err[i];
    free(row_err);
}

/* ------------------------------------------------------------------------ */
/* Jaccard -> distance: src/sourmash/distance_utils.py:349-407                */
/* (jaccard_to_distance) with r1_to_q :128-131, var_n_mutated :134-152,       */
/* exp_n_mutated :155-157, in the reference's operation order (Python floats; */
/* `**` on floats is libm pow).  Returns the point estimate, *err_lower_bound */
/* = the approximation error the reference compares with 1e-4                 */
/* (jaccardANIResult: ani is None when it is exceeded).  *bad = 1 when the    */
/* reference would raise "varN <0.0".                                         */
/* ------------------------------------------------------------------------ */
ORC_API double orc_jaccard_to_distance(double jaccard, uint32_t ksize, uint64_t n_unique_kmers, double *err_lower_bound, int *bad) {
    *bad = 0;
    if (jaccard == 0.0) { *err_lower_bound = 0.0; return 1.0; }
    if (jaccard == 1.0) { *err_lower_bound = 0.0; return 0.0; }
    const double k = (double)ksize, L = (double)n_unique_kmers;
    const double r1 = 1.0 - pow(2.0 * jaccard / (1.0 + jaccard), 1.0 / k);
    const double q = 1.0 - pow(1.0 - r1, k);                               /* r1_to_q */
    const double exp_n_mut = L * q;                                        /* exp_n_mutated */
    double varN = 0.0;
    if (r1 != 0.0) {                                                       /* var_n_mutated */
        varN = L * (1.0 - q) * (q * (2.0 * k + (2.0 / r1) - 1.0) - 2.0 * k)
             + k * (k - 1.0) * pow(1.0 - q, 2.0)
             + (2.0 * (1.0 - q) / pow(r1, 2.0)) * ((1.0 + (k - 1.0) * (1.0 - q)) * r1 - q);
        if (varN < 0.0) { *bad = 1; varN = 0.0; }
    }
    *err_lower_bound = 1.0 * L * varN / pow(L + exp_n_mut, 3.0);
    return r1;
}
