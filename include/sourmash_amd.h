/*
 * sourmash_amd.h -- C-ABI of libsourmash_amd.so, the MI355X-native FracMinHash engine.
 *
 * PART 1 is a drop-in for the hot-path subset of the reference's cffi boundary
 * (reference header: include/sourmash.h, generated from the Rust shims under src/core/src/ffi/).
 * Every declaration has the reference's name, argument order, ownership rule
 * and error convention; the comment on each group cites the reference lines it
 * replaces.  A maintainer switches the path over by loading this library where
 * `sourmash._lowlevel.lib` is loaded today.
 *
 * PART 2 are additive batch entry points (prefix smgpu_) that collapse the
 * reference's per-record / per-pair / per-round Python loops into single calls
 * on device-resident data.  Plain pointers and sizes only -- no torch types.
 *
 * Error convention (src/core/src/ffi/utils.rs:17-19,58-83,195-207;
 * src/sourmash/utils.py:65-78): a fallible function stores its error in
 * thread-local state and returns an all-zero value; callers bracket calls with
 * sourmash_err_clear() / sourmash_err_get_last_code().  There is NO CPU
 * fallback: k-mer hashing and sketch intersection run on the GPU or fail with
 * SOURMASH_ERROR_CODE_INTERNAL.
 */
#ifndef SOURMASH_AMD_H_INCLUDED
#define SOURMASH_AMD_H_INCLUDED

#include <stdarg.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ======================= PART 1: reference-compatible ABI ======================= */

/* include/sourmash.h:11-17 (src/core/src/ffi/mod.rs:33-68) */
/* (anonymous enum + integer typedef: identical ABI to the reference's `enum X {..}; typedef uint32_t X;`,
 * and also valid C++) */
enum {
  HASH_FUNCTIONS_MURMUR64_DNA = 1,
  HASH_FUNCTIONS_MURMUR64_PROTEIN = 2,
  HASH_FUNCTIONS_MURMUR64_DAYHOFF = 3,
  HASH_FUNCTIONS_MURMUR64_HP = 4,
};
typedef uint32_t HashFunctions;

/* include/sourmash.h:19-53 (src/core/src/errors.rs:101-143) -- values are ABI */
enum {
  SOURMASH_ERROR_CODE_NO_ERROR = 0,
  SOURMASH_ERROR_CODE_PANIC = 1,
  SOURMASH_ERROR_CODE_INTERNAL = 2,
  SOURMASH_ERROR_CODE_MSG = 3,
  SOURMASH_ERROR_CODE_UNKNOWN = 4,
  SOURMASH_ERROR_CODE_MISMATCH_K_SIZES = 101,
  SOURMASH_ERROR_CODE_MISMATCH_DNA_PROT = 102,
  SOURMASH_ERROR_CODE_MISMATCH_SCALED = 103,
  SOURMASH_ERROR_CODE_MISMATCH_SEED = 104,
  SOURMASH_ERROR_CODE_MISMATCH_SIGNATURE_TYPE = 105,
  SOURMASH_ERROR_CODE_NON_EMPTY_MIN_HASH = 106,
  SOURMASH_ERROR_CODE_MISMATCH_NUM = 107,
  SOURMASH_ERROR_CODE_NEEDS_ABUNDANCE_TRACKING = 108,
  SOURMASH_ERROR_CODE_CANNOT_UPSAMPLE_SCALED = 109,
  SOURMASH_ERROR_CODE_NO_MIN_HASH_FOUND = 110,
  SOURMASH_ERROR_CODE_EMPTY_SIGNATURE = 111,
  SOURMASH_ERROR_CODE_MULTIPLE_SKETCHES_FOUND = 112,
  SOURMASH_ERROR_CODE_INVALID_DNA = 1101,
  SOURMASH_ERROR_CODE_INVALID_PROT = 1102,
  SOURMASH_ERROR_CODE_INVALID_CODON_LENGTH = 1103,
  SOURMASH_ERROR_CODE_INVALID_HASH_FUNCTION = 1104,
  SOURMASH_ERROR_CODE_READ_DATA = 1201,
  SOURMASH_ERROR_CODE_STORAGE = 1202,
  SOURMASH_ERROR_CODE_HLL_PRECISION_BOUNDS = 1301,
  SOURMASH_ERROR_CODE_ANI_ESTIMATION_ERROR = 1401,
  SOURMASH_ERROR_CODE_IO = 100001,
  SOURMASH_ERROR_CODE_UTF8_ERROR = 100002,
  SOURMASH_ERROR_CODE_PARSE_INT = 100003,
  SOURMASH_ERROR_CODE_SERDE_ERROR = 100004,
  SOURMASH_ERROR_CODE_NIFFLER_ERROR = 100005,
  SOURMASH_ERROR_CODE_CSV_ERROR = 100006,
  SOURMASH_ERROR_CODE_ROCKS_DB_ERROR = 100007,
};
typedef uint32_t SourmashErrorCode;

/* opaque handles, include/sourmash.h:55-69: created by *_new / *_from_params,
 * owned by the caller, released by *_free (NULL tolerated, ffi/utils.rs:50-55) */
typedef struct SourmashComputeParameters SourmashComputeParameters;
typedef struct SourmashKmerMinHash SourmashKmerMinHash;
typedef struct SourmashSignature SourmashSignature;

/* include/sourmash.h:74-87 (ffi/utils.rs:209-316) */
typedef struct {
  char *data;
  uintptr_t len;
  bool owned;
} SourmashStr;

/* ---- library / errors: include/sourmash.h:416-467 (ffi/utils.rs:85-193) ---- */
void sourmash_init(void);
void sourmash_err_clear(void);
SourmashErrorCode sourmash_err_get_last_code(void);
SourmashStr sourmash_err_get_last_message(void);
SourmashStr sourmash_err_get_backtrace(void);
void sourmash_str_free(SourmashStr *s);
SourmashStr sourmash_str_from_cstr(const char *s);

/* ---- include/sourmash.h:133 (ffi/mod.rs:22-31; lib.rs:57-59): MurmurHash3_x64_128 h1 ---- */
uint64_t hash_murmur(const char *kmer, uint64_t seed);
/* include/sourmash.h:416-418,467 -- residue helpers of the protein / dayhoff / hp sketches */
char sourmash_aa_to_dayhoff(char aa);
char sourmash_aa_to_hp(char aa);
char sourmash_translate_codon(const char *codon);

/* ---- compute parameters: include/sourmash.h:89-131 (ffi/cmd/compute.rs:1-170) ---- */
SourmashComputeParameters *computeparams_new(void);
void computeparams_free(SourmashComputeParameters *ptr);
bool computeparams_dayhoff(const SourmashComputeParameters *ptr);
bool computeparams_dna(const SourmashComputeParameters *ptr);
bool computeparams_hp(const SourmashComputeParameters *ptr);
bool computeparams_protein(const SourmashComputeParameters *ptr);
bool computeparams_track_abundance(const SourmashComputeParameters *ptr);
const uint32_t *computeparams_ksizes(const SourmashComputeParameters *ptr, uintptr_t *size);
void computeparams_ksizes_free(uint32_t *ptr, uintptr_t insize);
uint32_t computeparams_num_hashes(const SourmashComputeParameters *ptr);
uint64_t computeparams_scaled(const SourmashComputeParameters *ptr);
uint64_t computeparams_seed(const SourmashComputeParameters *ptr);
void computeparams_set_dayhoff(SourmashComputeParameters *ptr, bool v);
void computeparams_set_dna(SourmashComputeParameters *ptr, bool v);
void computeparams_set_hp(SourmashComputeParameters *ptr, bool v);
void computeparams_set_protein(SourmashComputeParameters *ptr, bool v);
void computeparams_set_track_abundance(SourmashComputeParameters *ptr, bool v);
void computeparams_set_ksizes(SourmashComputeParameters *ptr, const uint32_t *ksizes_ptr, uintptr_t insize);
void computeparams_set_num_hashes(SourmashComputeParameters *ptr, uint32_t num);
void computeparams_set_scaled(SourmashComputeParameters *ptr, uint64_t scaled);
void computeparams_set_seed(SourmashComputeParameters *ptr, uint64_t new_seed);

/* ---- sketch object: include/sourmash.h:169-273 (ffi/minhash.rs:1-483) ---- */
SourmashKmerMinHash *kmerminhash_new(uint64_t scaled, uint32_t k, HashFunctions hash_function, uint64_t seed,
                                     bool track_abundance, uint32_t n);
void kmerminhash_free(SourmashKmerMinHash *ptr);
void kmerminhash_slice_free(uint64_t *ptr, uintptr_t insize);
/* HOT: sequence -> k-mers -> canonical -> murmur -> keep (signature.rs:38-58,246-306). NUL-terminated. */
void kmerminhash_add_sequence(SourmashKmerMinHash *ptr, const char *sequence, bool force);
const uint64_t *kmerminhash_seq_to_hashes(SourmashKmerMinHash *ptr, const char *sequence, uintptr_t insize,
                                          bool force, bool bad_kmers_as_zeroes, bool is_protein, uintptr_t *size);
void kmerminhash_add_hash(SourmashKmerMinHash *ptr, uint64_t h);
void kmerminhash_add_hash_with_abundance(SourmashKmerMinHash *ptr, uint64_t h, uint64_t abundance);
void kmerminhash_add_many(SourmashKmerMinHash *ptr, const uint64_t *hashes_ptr, uintptr_t insize);
void kmerminhash_add_from(SourmashKmerMinHash *ptr, const SourmashKmerMinHash *other);
void kmerminhash_add_word(SourmashKmerMinHash *ptr, const char *word);
void kmerminhash_add_protein(SourmashKmerMinHash *ptr, const char *sequence);   /* out of scope: raises */
void kmerminhash_remove_hash(SourmashKmerMinHash *ptr, uint64_t h);
void kmerminhash_remove_many(SourmashKmerMinHash *ptr, const uint64_t *hashes_ptr, uintptr_t insize);
void kmerminhash_remove_from(SourmashKmerMinHash *ptr, const SourmashKmerMinHash *other);
void kmerminhash_clear(SourmashKmerMinHash *ptr);
const uint64_t *kmerminhash_get_mins(const SourmashKmerMinHash *ptr, uintptr_t *size);
uintptr_t kmerminhash_get_mins_size(const SourmashKmerMinHash *ptr);
const uint64_t *kmerminhash_get_abunds(SourmashKmerMinHash *ptr, uintptr_t *size);
void kmerminhash_set_abundances(SourmashKmerMinHash *ptr, const uint64_t *hashes_ptr, const uint64_t *abunds_ptr,
                                uintptr_t insize, bool clear);
SourmashStr kmerminhash_md5sum(const SourmashKmerMinHash *ptr);
void kmerminhash_merge(SourmashKmerMinHash *ptr, const SourmashKmerMinHash *other);
bool kmerminhash_is_compatible(const SourmashKmerMinHash *ptr, const SourmashKmerMinHash *other);
/* HOT: sorted-u64 merge intersections (minhash.rs:539-631,635-702,915-953,1721-1807) */
uint64_t kmerminhash_count_common(const SourmashKmerMinHash *ptr, const SourmashKmerMinHash *other, bool downsample);
SourmashKmerMinHash *kmerminhash_intersection(const SourmashKmerMinHash *ptr, const SourmashKmerMinHash *other);
uint64_t kmerminhash_intersection_union_size(const SourmashKmerMinHash *ptr, const SourmashKmerMinHash *other,
                                             uint64_t *union_size);
double kmerminhash_jaccard(const SourmashKmerMinHash *ptr, const SourmashKmerMinHash *other);
double kmerminhash_similarity(const SourmashKmerMinHash *ptr, const SourmashKmerMinHash *other,
                              bool ignore_abundance, bool downsample);
double kmerminhash_angular_similarity(const SourmashKmerMinHash *ptr, const SourmashKmerMinHash *other);
uint32_t kmerminhash_num(const SourmashKmerMinHash *ptr);
uint32_t kmerminhash_ksize(const SourmashKmerMinHash *ptr);
uint64_t kmerminhash_seed(const SourmashKmerMinHash *ptr);
uint64_t kmerminhash_max_hash(const SourmashKmerMinHash *ptr);
HashFunctions kmerminhash_hash_function(const SourmashKmerMinHash *ptr);
void kmerminhash_hash_function_set(SourmashKmerMinHash *ptr, HashFunctions hash_function);
bool kmerminhash_is_protein(const SourmashKmerMinHash *ptr);
bool kmerminhash_dayhoff(const SourmashKmerMinHash *ptr);
bool kmerminhash_hp(const SourmashKmerMinHash *ptr);
bool kmerminhash_track_abundance(const SourmashKmerMinHash *ptr);
void kmerminhash_enable_abundance(SourmashKmerMinHash *ptr);
void kmerminhash_disable_abundance(SourmashKmerMinHash *ptr);

/* ---- signature container: include/sourmash.h:364-414 (ffi/signature.rs:1-344) ---- */
SourmashSignature *signature_new(void);
void signature_free(SourmashSignature *ptr);
SourmashSignature *signature_from_params(const SourmashComputeParameters *ptr);
uintptr_t signature_len(const SourmashSignature *ptr);
/* HOT: fans the record out over every sketch of the signature (signature.rs:661-677) */
void signature_add_sequence(SourmashSignature *ptr, const char *sequence, bool force);
void signature_add_protein(SourmashSignature *ptr, const char *sequence);       /* out of scope: raises */
void signature_set_name(SourmashSignature *ptr, const char *name);
void signature_set_filename(SourmashSignature *ptr, const char *name);
SourmashStr signature_get_name(const SourmashSignature *ptr);
SourmashStr signature_get_filename(const SourmashSignature *ptr);
SourmashStr signature_get_license(const SourmashSignature *ptr);
SourmashKmerMinHash *signature_first_mh(const SourmashSignature *ptr);
SourmashKmerMinHash **signature_get_mhs(const SourmashSignature *ptr, uintptr_t *size);
void signature_set_mh(SourmashSignature *ptr, const SourmashKmerMinHash *other);
void signature_push_mh(SourmashSignature *ptr, const SourmashKmerMinHash *other);
bool signature_eq(const SourmashSignature *ptr, const SourmashSignature *other);
SourmashStr signature_save_json(const SourmashSignature *ptr);
SourmashSignature **signatures_load_buffer(const char *ptr, uintptr_t insize, bool _ignore_md5sum, uintptr_t ksize,
                                           const char *select_moltype, uintptr_t *size);
SourmashSignature **signatures_load_path(const char *ptr, bool _ignore_md5sum, uintptr_t ksize,
                                         const char *select_moltype, uintptr_t *size);
const uint8_t *signatures_save_buffer(const SourmashSignature *const *ptr, uintptr_t size, uint8_t compression,
                                      uintptr_t *osize);
/* generic byte-buffer free (src/sourmash/signature.py:514 frees signatures_save_buffer output with it) */
void nodegraph_buffer_free(uint8_t *ptr, uintptr_t insize);

/* ============================ PART 2: batch extensions ============================ */
/* Same conventions (TLS error, zero on failure).  "d_" pointers are device
 * pointers valid on the current HIP device; `stream` is a hipStream_t passed as
 * void* (NULL = the null stream).  None of the *_raw functions allocates or
 * synchronises unless stated. */

/* number of visible HIP devices (0 when none / no driver); never fails */
int32_t smgpu_device_count(void);
/* 1 if the k-mer / intersection kernels can run in this process */
bool smgpu_available(void);

/* Collapses the per-record loop of src/sourmash/command_sketch.py:746-768 +
 * src/core/src/signature.rs:38-58: sketch a whole buffer (records separated by any
 * byte outside ACGTacgt, e.g. '\n'; no NUL needed) into `ptr`. */
void smgpu_minhash_add_buffer(SourmashKmerMinHash *ptr, const char *buf, uintptr_t len, bool force);

/* The whole `sourmash sketch dna <file>` inner loop (command_sketch.py:697,746-768) in one call: a native
 * FASTA / FASTQ (plain or gzip) reader strips headers and line breaks, streams 64 MiB chunks through
 * pinned staging buffers to the GPU, and every sketch of the signature (all ksizes) is filled in the same
 * pass.  force=True semantics (bytes outside ACGTacgt drop the k-mers covering them).  Returns the
 * number of sequence bytes read; *n_records = number of records. */
/* Per-record sketching without a launch per record.  kmerminhash_add_sequence / signature_add_sequence (the reference's
 * loop: src/sourmash/command_sketch.py:746-768, src/core/src/signature.rs:38-58) validate the record, queue it in the
 * sketch and return; the queue goes through the sketch kernel in one launch when it reaches 32 MiB or when any accessor
 * looks at the sketch.  force = false still raises InvalidDNA from the offending call, after the k-mers in front of the
 * first bad one were queued (signature.rs:48-54).  smgpu_minhash_add_sequence_rc is add_sequence with the error code as
 * the return value (len bytes, a NUL ends the record early like the C string of the reference's entry point);
 * smgpu_minhash_flush settles the queue now; smgpu_minhash_pending_bytes tells how much is queued. */
uint32_t smgpu_minhash_add_sequence_rc(SourmashKmerMinHash *ptr, const char *sequence, uintptr_t len, bool force);
void smgpu_minhash_flush(SourmashKmerMinHash *ptr);
uint64_t smgpu_minhash_pending_bytes(const SourmashKmerMinHash *ptr);
uint64_t smgpu_signature_add_file(SourmashSignature *ptr, const char *path, uint64_t *n_records);
/* `sourmash sketch dna` over many files at once: n_threads workers (0 = min(16, host cores)), each with its own
 * stream, pinned ring and device buffers, one signature per file in input order (every ksize of `params` in it).
 * Returns an array of n signature handles (free each with signature_free and the array with nodegraph_buffer_free,
 * like signatures_load_buffer's). */
SourmashSignature **smgpu_sketch_files(const char *const *paths, uintptr_t n, const SourmashComputeParameters *params,
                                       uint32_t n_threads, uint64_t *total_bases);
uint64_t smgpu_minhash_add_file(SourmashKmerMinHash *ptr, const char *path, uint64_t *n_records);
/* The inflater behind the .gz ingest by itself (csrc/gunzip.hpp: the members' compressed bytes go to HBM, every deflate block of
 * a member is decoded at once, CRC-32 and length are checked against the trailer; what niffler / screed do on one host thread,
 * src/core/benches/compute.rs:35-38, src/sourmash/command_sketch.py:697).  n single-member gzip files, inflated in ONE batch;
 * their bytes come back to out[0, capacity) one behind the other, lens[i] = bytes of file i or UINT64_MAX when the device refused
 * it (several members, damaged, not gzip): such a file is the host inflater's.  stats (NULL or 16 doubles): [0] survivors of the
 * cheap header test, [1] candidates decoded, [2] runs on the chains, [3..8] milliseconds: scan, pass 1, link, pass 2, tails +
 * resolve + CRC, everything on the device; [9] H2D copy + file reads.  Returns the bytes written. */
uint64_t smgpu_gunzip_files(const char *const *paths, uintptr_t n, uint8_t *out, uint64_t capacity, uint64_t *lens, double *stats);
/* out[0]: signature documents smgpu_sketchset_load has parsed with the device doing inflate and number parsing (csrc/sigload.hpp)
 * since the library was loaded, out[1]: documents its host parser took (SMG_SIGLOAD_DEVICE=0 sends every document there). */
void smgpu_sigload_counters(uint64_t *out);
/* out[0]: gzip files the ingest inflated on the device since the library was loaded, out[1]: files it handed to the host inflater
 * after the device refused them. */
void smgpu_gunzip_counters(uint64_t *out);

/* Scratch size needed by smgpu_sketch_dna_raw for an output capacity. */
uint64_t smgpu_sketch_workspace_bytes(uint64_t out_capacity);
/* Device-resident sketching: d_seq[0,len) ASCII (any alignment) -> sorted unique
 * kept hashes (1 <= h <= max_hash; max_hash 0 = keep all) in d_out[0, n).
 * d_result (device, 2 x u64): [0] kept k-mer occurrences, [1] unique hashes n.
 * Synchronises the stream once (the sort needs the kept count).  Returns n, or
 * UINT64_MAX with an error set; if kept > out_capacity the error says so and the
 * caller retries with a larger buffer. */
uint64_t smgpu_sketch_dna_raw(const uint8_t *d_seq, uint64_t len, uint32_t ksize, uint64_t seed, uint64_t max_hash,
                              uint64_t *d_out, uint64_t out_capacity, uint64_t *d_result, void *d_workspace,
                              uint64_t workspace_bytes, void *stream);
/* Only the k-mer kernel (no sort): appends kept hashes unordered to d_out, adds the
 * count to *d_count (device u64, caller zeroes).  Fully asynchronous. */
void smgpu_sketch_dna_kernel_raw(const uint8_t *d_seq, uint64_t len, uint32_t ksize, uint64_t seed, uint64_t max_hash,
                                 uint64_t *d_out, uint64_t out_capacity, uint64_t *d_count, void *stream);
/* Union of hash vectors on the device (the `merge` of flat scaled sketches, src/core/src/sketch/minhash.rs:432-516, for
 * vectors already in HBM -- e.g. the all-gathered per-rank sketches): d_keys[0, n) in any order, duplicates allowed ->
 * the sorted distinct values in d_out[0, m) (capacity n); *d_n_out (device u64) = m.  d_keys is used as scratch.  The
 * workspace is smgpu_sketch_workspace_bytes(n) bytes.  Radix sort + run-length encode (csrc/device_sort.hip).
 * Synchronises the stream once.  Returns m, or UINT64_MAX with an error set. */
uint64_t smgpu_sort_unique_raw(uint64_t *d_keys, uint64_t n, uint64_t *d_out, uint64_t *d_n_out, void *d_workspace,
                               uint64_t workspace_bytes, void *stream);
/* The kernels of protein / dayhoff / hp sketches on device-resident input (src/core/src/signature.rs:307-393,
 * src/core/src/encodings.rs:103-368): the residues of a protein sequence -- or, translate = true, the six-frame translation of
 * DNA -- go to d_aa (capacity aa_capacity bytes: the residue count rounded up to 8 -- the window kernel reads whole aligned
 * 8-byte words; translated DNA needs 2 * len + 6 residues), every window of k_aa residues is hashed and
 * the hashes 1 <= h <= max_hash are appended unordered to d_out, their count added to *d_count (device u64, caller zeroes).
 * -> residues written to d_aa.  Fully asynchronous.  hash_function: 2 protein, 3 dayhoff, 4 hp. */
uint64_t smgpu_sketch_residues_kernels_raw(const uint8_t *d_seq, uint64_t len, uint32_t k_aa, uint32_t hash_function, uint64_t seed,
                                           uint64_t max_hash, bool translate, uint8_t *d_aa, uint64_t aa_capacity, uint64_t *d_out,
                                           uint64_t out_capacity, uint64_t *d_count, void *stream);
/* Synthetic random DNA written straight into HBM (BASELINE config C2 generator). */
void smgpu_synth_dna_raw(uint8_t *d_out, uint64_t start, uint64_t n, uint64_t seed, uint64_t record_len, void *stream);

/* Collapses compare_all_pairs (src/sourmash/compare.py:14-64,328-358): CSR of n
 * sorted sketches on device -> u32 common[(row_hi-row_lo)][n] and/or f64
 * jaccard[(row_hi-row_lo)][n] for rows [row_lo,row_hi).  Either output may be NULL
 * (d_common is required as scratch if d_jaccard is given).  Asynchronous. */
void smgpu_compare_raw(const uint64_t *d_hashes, const uint64_t *d_offsets, uint32_t n, uint32_t row_lo,
                       uint32_t row_hi, uint32_t *d_common, double *d_jaccard, void *stream);
/* Multi-GPU form of the same kernel (BASELINE config C4: row-block shard per GPU): this launch owns
 * the 16-row tiles rb_first, rb_first + rb_stride, ... (rb_count tiles) of the n x n problem and
 * computes their tiles on/above the diagonal into d_common[rb_count*16][n] (zero-initialised by the
 * caller).  After one all-gather of the shards, smgpu_symmetrize_raw mirrors the upper triangle of the
 * full matrix and smgpu_jaccard_raw converts counts (rows [row_lo,row_hi) of a full matrix slice that
 * starts at row_lo) to f64 Jaccard. */
void smgpu_compare_blocks_raw(const uint64_t *d_hashes, const uint64_t *d_offsets, uint32_t n, uint32_t rb_first,
                              uint32_t rb_stride, uint32_t rb_count, uint32_t *d_common, void *stream);
void smgpu_symmetrize_raw(uint32_t *d_common, uint32_t n, void *stream);
void smgpu_jaccard_raw(const uint32_t *d_common, const uint64_t *d_offsets, uint32_t n, uint32_t row_lo,
                       uint32_t row_hi, double *d_jaccard, void *stream);
/* Indexed paths of the same comparison.  smgpu_bitindex_new groups every (hash, row) of the collection by hash (hash-space
 * buckets with LDS tables, csrc/dictindex.hip; a radix sort when a bucket holds more than 1,024 distinct hashes;
 * synchronises the stream once) and splits the hashes by how many sketches hold them: frequent ones become bit columns
 * (|A ∩ B| = popcount(A & B) over bit rows), rare ones inverted lists whose pairs are incremented directly.  A
 * collection drawn from one pool ends up all bit rows, a collection of unrelated genomes all inverted lists.
 * Returns NULL -- with no error set -- when the cost model prefers the merge kernel (smgpu_compare_*_raw).
 * smgpu_bitindex_compare_raw fills d_common[rb_count*16][n] for the owned 16-row tiles, ALL columns. */
typedef struct SmgpuBitIndex SmgpuBitIndex;
SmgpuBitIndex *smgpu_bitindex_new(const uint64_t *d_hashes, const uint64_t *d_offsets, uint32_t n, void *stream);
/* same with the knobs exposed: total_hashes = d_offsets[n] if the caller knows it (0: read back, one more
 * synchronisation); threshold > 0 forces the frequent/rare split instead of the cost model's (and then never returns NULL
 * unless the bit rows would exceed the memory cap); one_shot: the index will serve a single compare, so its own build
 * time counts against it. */
SmgpuBitIndex *smgpu_bitindex_new_ex(const uint64_t *d_hashes, const uint64_t *d_offsets, uint32_t n, uint64_t total_hashes,
                                     uint32_t threshold, bool one_shot, void *stream);
void smgpu_bitindex_free(SmgpuBitIndex *ptr);
uint64_t smgpu_bitindex_universe(const SmgpuBitIndex *ptr);
uint32_t smgpu_bitindex_builder(const SmgpuBitIndex *ptr);   /* which builder made it: 1 sort-free dictionary passes, 2 radix sort (diagnostic) */
/* how the index splits the collection: hashes held by more than `threshold` sketches are bit columns, the others
 * are inverted lists costing `rare_pairs` matrix increments per compare */
void smgpu_bitindex_stats(const SmgpuBitIndex *ptr, uint64_t *frequent_hashes, uint64_t *rare_pairs, uint32_t *threshold);
void smgpu_bitindex_compare_raw(const SmgpuBitIndex *ptr, uint32_t rb_first, uint32_t rb_stride, uint32_t rb_count,
                                uint32_t *d_common, void *stream);
/* The same for callers that mirror the triangle afterwards (smgpu_symmetrize_raw, as the all-pairs drivers do): only the
 * entries on or above the diagonal of the owned rows are computed (the counterpart of smgpu_compare_blocks_raw); what lies
 * below is left as it was.  Half the tiles of a world-size-1 launch. */
void smgpu_bitindex_compare_upper_raw(const SmgpuBitIndex *ptr, uint32_t rb_first, uint32_t rb_stride, uint32_t rb_count,
                                      uint32_t *d_common, void *stream);
/* Host float helper for the containment / ANI matrices of compare (src/sourmash/compare.py:67-187): out[i] =
 * pow(x[i], y[ny == 1 ? 0 : i]) with the host libm, i.e. the bits of the reference's per-pair Python `**`
 * (src/sourmash/minhash.py:832-834, src/sourmash/distance_utils.py:283).  n_threads = 0: every host core. */
void smgpu_host_pow_f64(const double *x, const double *y, uintptr_t ny, double *out, uintptr_t n, uint32_t n_threads);
/* Host helper of add_sequence(force = false): index of the first byte of seq[0, len) outside ACGTacgt (what makes a k-mer
 * invalid: src/core/src/encodings.rs:370-377 after the upper-casing of src/core/src/signature.rs:214), or UINTPTR_MAX
 * when there is none.  No device involved; exported so that the CPU test suite can pin the vectorised scan. */
uintptr_t smgpu_first_invalid_dna_byte(const char *seq, uintptr_t len);
/* n sketch handles -> n x n matrices on the host (either may be NULL): what src/sourmash/compare.py:326-358 walks pair by pair
 * through kmerminhash_similarity / kmerminhash_count_common (src/core/src/ffi/minhash.rs:409-457).  The sketches are packed into
 * pinned chunks by worker threads and travel as one H2D copy per chunk; the matrices come back through the same ring. */
void smgpu_compare_all_pairs(const SourmashKmerMinHash *const *mhs, uintptr_t n, uint32_t *common_out,
                             double *jaccard_out);
/* The same for a list with SEVERAL scaled values and downsample = true (compare.py:14-64; src/core/src/sketch/minhash.rs:682-702):
 * common_out[i][j] = |A ∩ B| with both sketches downsampled to the pair's coarser scaled -- a prefix of the finer row
 * (minhash.rs:777-798) -- and the row's own size on the diagonal.  class_of[i] = index of sketch i's scaled value in the ascending
 * list of the list's distinct values, class_max_hash[c] = max_hash of value c; sizes_out [n_classes][n] receives the size of
 * sketch i downsampled to value c (0 where c is finer than the sketch).  One upload, no downsampled host objects.  Errors: the
 * ksize / molecule / seed mismatch of the first sketch that does not match sketch 0. */
void smgpu_compare_all_pairs_mixed(const SourmashKmerMinHash *const *mhs, uintptr_t n, const uint32_t *class_of,
                                   const uint64_t *class_max_hash, uintptr_t n_classes, uint32_t *common_out, uint64_t *sizes_out);
/* Views for batched callers (src/sourmash/compare.py:326-358 reads `sig.minhash` per pair, and signature_first_mh
 * (src/core/src/ffi/signature.rs:167-182) CLONES the sketch every time): out_mhs[i] = the first sketch of sigs[i] as a BORROWED
 * handle -- valid while the signature lives and is not modified, never to be freed -- and params[i][8] = {ksize as stored,
 * hash_function, seed, max_hash, num, track_abundance, number of hashes, 0} in one call.  smgpu_minhashes_params: the same
 * parameter rows for sketch handles. */
void smgpu_signatures_sketch_views(const SourmashSignature *const *sigs, uintptr_t n, const SourmashKmerMinHash **out_mhs,
                                   uint64_t *params);
void smgpu_minhashes_params(const SourmashKmerMinHash *const *mhs, uintptr_t n, uint64_t *params);
/* Page-locked host memory (hipHostMalloc through the library's arena, which caches released blocks up to SMG_PINNED_CACHE_MAX
 * bytes): result matrices placed there are filled by ONE device-to-host copy at the link's rate.  smgpu_host_free takes only
 * pointers smgpu_host_alloc returned (others are ignored). */
void *smgpu_host_alloc(uintptr_t bytes);
void smgpu_host_free(void *ptr);
/* Bytes and nanoseconds the host-pointer entry points spent moving large pageable buffers (csrc/hostxfer.hpp) since the last reset:
 * out5 = {H2D bytes, D2H bytes, H2D ns, D2H ns, calls}.  Diagnostics for bench.py's API-level lines. */
void smgpu_xfer_stats(uint64_t *out5, bool reset);
/* The two directions of that path by themselves (tests): the n host pieces (pieces[i], lens[i] bytes, pageable or pinned, empty
 * ones allowed) are packed into one device buffer through the pinned ring, and that buffer comes back into out[0, out_bytes)
 * (out_bytes = the sum of lens) through the ring -- or directly when out is pinned (smgpu_host_alloc). */
void smgpu_xfer_roundtrip(const void *const *pieces, const uint64_t *lens, uintptr_t n, void *out, uint64_t out_bytes);
/* All pairs of BOTTOM-K (num) sketches in one launch (csrc/compare_ext.hip) -- what src/sourmash/compare.py:36-54 asks
 * kmerminhash_similarity for pair by pair: common_out[i][j] = |A ∩ B ∩ merged|, union_out[i][j] = |merged| with merged = the
 * num smallest hashes of A ∪ B, num taken from the sketch with the lower index (src/core/src/sketch/minhash.rs:593-621: the
 * merged sketch is built with self.num), jaccard_out = common / max(1, union) (minhash.rs:624-631), 1.0 on the diagonal.  n x n
 * host matrices, any may be NULL.  Errors: the compatibility error of the first sketch that does not match sketch 0. */
void smgpu_compare_num_all_pairs(const SourmashKmerMinHash *const *mhs, uintptr_t n, uint32_t *common_out, uint32_t *union_out,
                                 double *jaccard_out);
/* All pairs of ABUNDANCE-TRACKING sketches: sims_out[i][j] = angular similarity 1 - 2 acos(min(sum a_i b_j / (|a| |b|), 1)) / pi
 * (src/core/src/sketch/minhash.rs:635-680; 0 when a norm is 0; 1.0 on the diagonal as compare.py:33 has it).  The integer sums
 * come from one launch (prod_out[i][j] = sum over the common hashes of abund_i x abund_j, sumsq_out[i] = sum of squares; u64,
 * wrapping; optional), sqrt / acos from the host's libm.  Errors: compatibility as above; NeedsAbundanceTracking. */
void smgpu_compare_angular_all_pairs(const SourmashKmerMinHash *const *mhs, uintptr_t n, double *sims_out, uint64_t *prod_out,
                                     uint64_t *sumsq_out);
/* the host half of the above on its own: prod [n][n], sumsq [n] -> out [n][n] (n_threads 0: every host core) */
void smgpu_host_angular_f64(const uint64_t *prod, const uint64_t *sumsq, uintptr_t n, double *out, uint32_t n_threads);
/* The same kernels on caller-owned device buffers (CSR of sorted rows; d_nums[i] = num of sketch i; d_abunds parallel to
 * d_hashes; narrow = every abundance fits 32 bits).  Full n x n device matrices; d_union / d_jaccard may be NULL. */
void smgpu_compare_num_raw(const uint64_t *d_hashes, const uint64_t *d_offsets, const uint32_t *d_nums, uint32_t n,
                           uint32_t *d_common, uint32_t *d_union, double *d_jaccard, void *stream);
void smgpu_compare_abund_raw(const uint64_t *d_hashes, const uint64_t *d_abunds, const uint64_t *d_offsets, uint32_t n, bool narrow,
                             uint32_t *d_common, uint64_t *d_prod, uint64_t *d_sumsq, void *stream);
/* The same with the number of hashes (= d_offsets[n]) given by the caller: only enqueues work.  The form above reads d_offsets[n]
 * back first and so blocks the caller until `stream` has drained. */
void smgpu_compare_abund_raw_n(const uint64_t *d_hashes, const uint64_t *d_abunds, const uint64_t *d_offsets, uint32_t n,
                               uint64_t total_hashes, bool narrow, uint32_t *d_common, uint64_t *d_prod, uint64_t *d_sumsq, void *stream);

/* Device-resident sketch collection + gather counters: the batched form of CounterGather
 * (src/sourmash/index/__init__.py:735-909).  A SketchSet is a CSR of n sorted sketches in HBM;
 * a Counter holds c[d] = |Q ∩ D_d| for every member d (CounterGather.add, :783-789), finds the
 * best match with the reference tie-break (highest count, then first inserted = lowest index;
 * :856-857) and applies consume (c[d] -= |I ∩ D_d| for every live d; :897-909) in one kernel. */
typedef struct SmgpuSketchSet SmgpuSketchSet;
typedef struct SmgpuCounter SmgpuCounter;
SmgpuSketchSet *smgpu_sketchset_new(const SourmashKmerMinHash *const *mhs, uintptr_t n);
void smgpu_sketchset_free(SmgpuSketchSet *ptr);
uintptr_t smgpu_sketchset_len(const SmgpuSketchSet *ptr);
/* Bulk loading: signature files -> one CSR in HBM, with no per-sketch host object.  Replaces the per-signature
 * loops of src/sourmash/save_load.py:218-234,448-549 / src/core/src/signature.rs:569-659 for the collection side
 * of compare / search / gather.  `paths`: .sig, .sig.gz, .zip (members chosen through SOURMASH-MANIFEST.csv when
 * present, src/sourmash/manifest.py:15-387), directories (walked for *.sig / *.sig.gz) or text files listing one
 * path per line.  Selection: ksize (0 = any), moltype ("DNA", "protein", "dayhoff", "hp"; NULL = any), scaled
 * (0 = as stored; otherwise sketches with scaled <= this, downsampled to it).  Rows keep the input order; all
 * selected sketches must share ksize / moltype / seed / scaled (the reference's compatibility errors otherwise).
 * n_threads = 0 uses every host core. */
typedef struct SmgpuCollection SmgpuCollection;
/* The host half on its own (no GPU needed): the CSR and manifest of what sketchset_load would put in HBM. */
SmgpuCollection *smgpu_collection_load(const char *const *paths, uintptr_t n_paths, uint32_t ksize, const char *moltype,
                                       uint64_t scaled, uint32_t n_threads);
void smgpu_collection_free(SmgpuCollection *ptr);
uintptr_t smgpu_collection_len(const SmgpuCollection *ptr);
uint64_t smgpu_collection_total_hashes(const SmgpuCollection *ptr);
uint64_t smgpu_collection_skipped(const SmgpuCollection *ptr);
const uint64_t *smgpu_collection_hashes(const SmgpuCollection *ptr);   /* borrowed, valid until free */
const uint64_t *smgpu_collection_offsets(const SmgpuCollection *ptr);  /* borrowed, len + 1 entries */
SourmashStr smgpu_collection_manifest(const SmgpuCollection *ptr);
void smgpu_collection_params(const SmgpuCollection *ptr, uint32_t *ksize, uint32_t *hash_function, uint64_t *seed,
                             uint64_t *max_hash, uint64_t *num);
SmgpuSketchSet *smgpu_sketchset_from_collection(const SmgpuCollection *ptr);
SmgpuSketchSet *smgpu_sketchset_load(const char *const *paths, uintptr_t n_paths, uint32_t ksize, const char *moltype,
                                     uint64_t scaled, uint32_t n_threads);
/* rows[0..n) of a loaded set as a new set: row gather on the device, manifest rows follow.  Replaces the object loop of
 * Index.counter_gather (src/sourmash/index/__init__.py:302-320: `for result in self.prefetch(...): counter.add(
 * result.signature, ...)`): the rows that pass the prefetch become the counter's database without leaving HBM. */
SmgpuSketchSet *smgpu_sketchset_subset(const SmgpuSketchSet *set, const uint64_t *rows, uintptr_t n);
uint64_t smgpu_sketchset_total_hashes(const SmgpuSketchSet *ptr);
uint64_t smgpu_sketchset_skipped(const SmgpuSketchSet *ptr);
/* One manifest row per CSR row, in the reference's CSV manifest format (manifest.py:29-41,128-146). */
SourmashStr smgpu_sketchset_manifest(const SmgpuSketchSet *ptr);
void smgpu_sketchset_params(const SmgpuSketchSet *ptr, uint32_t *ksize, uint32_t *hash_function, uint64_t *seed,
                            uint64_t *max_hash, uint64_t *num);
void smgpu_sketchset_sizes(const SmgpuSketchSet *ptr, uint64_t *sizes_out);
SourmashKmerMinHash *smgpu_sketchset_get(const SmgpuSketchSet *ptr, uint64_t index);
void smgpu_sketchset_device_csr(const SmgpuSketchSet *ptr, const uint64_t **d_hashes, const uint64_t **d_offsets);
/* counts_out[row] = |query ∩ row| for every row of the set (Index.find's shared sizes, index/__init__.py:115-170) */
void smgpu_sketchset_overlaps(const SmgpuSketchSet *ptr, const SourmashKmerMinHash *query, uint64_t *counts_out);
/* n x n matrices of a loaded set on the host (either may be NULL); same kernels as smgpu_compare_all_pairs. */
void smgpu_sketchset_compare(const SmgpuSketchSet *ptr, uint32_t *common_out, double *jaccard_out);
SmgpuCounter *smgpu_counter_new(const SmgpuSketchSet *set, const SourmashKmerMinHash *query);
void smgpu_counter_free(SmgpuCounter *ptr);
void smgpu_counter_get(const SmgpuCounter *ptr, uint64_t *counts_out);
void smgpu_counter_set(SmgpuCounter *ptr, uint64_t index, uint64_t value);
bool smgpu_counter_best(const SmgpuCounter *ptr, uint64_t *index, uint64_t *count);
void smgpu_counter_consume(SmgpuCounter *ptr, const SourmashKmerMinHash *intersect);
/* The whole min-set-cover loop of GatherDatabases (src/sourmash/search.py:877-949) from the counter's current
 * state, every round on the GPU: winner index (order of smgpu_sketchset_new) and |intersect| per round into the
 * host arrays; -> number of rounds.  threshold_hashes = ceil(threshold_bp / scaled) (search.py:15-37). */
uint64_t smgpu_counter_gather(SmgpuCounter *ptr, uint64_t threshold_hashes, uint64_t *out_index, uint64_t *out_isect,
                              uint64_t cap);

/* The same loop over caller-owned device buffers (query: sorted unique u64; database: CSR shard whose rows have
 * global indices index_base, index_base+1, ...).  new_raw inverts the shard against the query once (query hash ->
 * rows holding it), so a round costs |I| postings walks instead of a pass over the database.
 *   single GPU:  begin, run.
 *   sharded:     begin, then per exchange  topk_export_raw (this shard's k best rows as records
 *                [key, bound, len, hashes...] of `stride` u64 words, key = (count << 32) | ~global index, bound = the
 *                best key the shard keeps back) -> ONE all-gather of the records -> cands_load_raw on every rank
 *                (at most 64 records in all) -> replay_raw(rounds): each round takes the best candidate, which is
 *                the global arg-max as long as its key is not below any kept-back key (counters only decrease), and
 *                applies it to the local postings and to the candidates' counters; rounds turn into no-ops once that
 *                test fails (the next exchange decides) or a stop rule fired.  poll every few exchanges.  Nothing
 *                between two polls needs the host; every rank replays the same rounds.
 *                Replaces the per-round walk of CounterGather.peek / consume, src/sourmash/index/__init__.py:817-909. */
typedef struct SmgpuGather SmgpuGather;
SmgpuGather *smgpu_gather_new_raw(const uint64_t *d_query, uint64_t nq, const uint64_t *d_hashes,
                                  const uint64_t *d_offsets, uint64_t ndb, uint64_t index_base, void *stream);
void smgpu_gather_free(SmgpuGather *ptr);
uint64_t smgpu_gather_postings(const SmgpuGather *ptr);
/* What the build and the last smgpu_gather_run cost, so that one benchmark line separates kernels from host effects
 * (waits for the build's kernels): out[0] build kernel span in ms (HIP events on the build's stream), out[1] build host
 * wall clock ms (the smgpu_gather_new_raw call), out[2] ms inside the driver's allocator during the build (0 once the
 * arena is warm), out[3] driver allocations made, out[4] host synchronisations of the build, out[5] ms the host waited in
 * them, out[6] GPU span in ms of the last run's rounds (events around the loop), out[7] host wall clock ms of that run,
 * out[8] times the resident loop kernel gave up because its grid was not resident as a whole (another kernel or process held
 * CUs) and the two-kernel rounds / the record protocol took over from the untouched state. */
void smgpu_gather_stats(SmgpuGather *ptr, double *out9);
/* The library's device arena (csrc/arena.hpp): every index / scratch block comes from it and is cached on release.
 * out[0] driver allocations, out[1] driver frees, out[2] ns inside the driver, out[3] reuse hits, out[4] live bytes,
 * out[5] cached bytes, out[6] peak bytes held, out[7] reuses that waited on another stream's event. */
void smgpu_arena_stats(uint64_t *out8);
void smgpu_arena_trim(uint64_t keep_bytes);     /* give cached blocks back to the driver (0: all of them) */
/* The gzip reader of the ingest path on its own (host threads, no device): inflates `path`, returns the decompressed length,
 * the CRC-32 of the bytes it produced and whether the many-thread form (csrc/pargz.hpp: block starts found by search,
 * window references resolved afterwards) carried the whole file; up to `cap` bytes are copied to `out` (may be NULL).
 * threads 0: the CPUs this process may use; span_bytes 0: 1 MiB of compressed data per work unit.
 * Replaces the single zlib stream behind screed / niffler (src/sourmash/command_sketch.py:697, src/core/benches/compute.rs:35-38). */
uint64_t smgpu_gunzip_file(const char *path, uint32_t threads, uint64_t span_bytes, uint8_t *out, uint64_t cap, uint32_t *crc32_out,
                           bool *parallel_used);
/* The 15-bit code the reader's marker bytes carry for window position `p` (an involution on 0..32767): the 128 codes whose two
 * marker bytes coincide -- and so read like a data byte >= 0x80 -- belong to positions 0..127 (tests/test_pargz_cpu.py). */
uint32_t smgpu_gunzip_position_code(uint32_t p);
void smgpu_gather_counters_get(const SmgpuGather *ptr, uint64_t *counts_out, void *stream);
void smgpu_gather_begin(SmgpuGather *ptr, uint64_t threshold_hashes, uint64_t max_rounds, void *stream);
uint64_t smgpu_gather_run(SmgpuGather *ptr, uint64_t *out_index, uint64_t *out_isect, uint64_t cap, void *stream);
/* Several ranks, one shard each, running the SAME rounds inside their resident loop kernels: the local winners meet in
 * host-visible memory every rank can reach (POSIX shared memory registered with HIP when the ranks are processes on one node;
 * private pinned memory when one process drives all of them), the best is the round's winner everywhere and its query
 * positions travel through the same memory -- no host collective inside the loop.  xchg_new: shm_name NULL/"" = private
 * memory; else rank 0 creates (create = true) and the others open it after a barrier.  rowcap >= the longest row of any shard
 * (smgpu_gather_longest_row, MAX over ranks).  launch_shared enqueues the armed loop (after smgpu_gather_begin) and returns
 * false when this index cannot run the resident loop (every rank must check smgpu_gather_loop_eligible first and all agree);
 * run_id: the same on every rank, different from the previous run on this memory; n_wg: workgroups (0 = one per CU).
 * smgpu_gather_results then waits and reads the picks (identical on every rank).  Replaces the per-round walk of
 * CounterGather.peek / consume across shards, src/sourmash/index/__init__.py:817-909. */
typedef struct SmgpuGatherXchg SmgpuGatherXchg;
SmgpuGatherXchg *smgpu_gather_xchg_new(const char *shm_name, uint32_t world, uint64_t rowcap, bool create);
void smgpu_gather_xchg_free(SmgpuGatherXchg *ptr);
bool smgpu_gather_loop_eligible(const SmgpuGather *ptr, uint32_t n_wg);
/* (one process driving several ranks: reserve every rank's loop memory BEFORE the first launch -- an allocation may synchronise
 *  the device while a loop that already runs waits for its peers) */
void smgpu_gather_loop_reserve(SmgpuGather *ptr, uint32_t n_wg, uint64_t rowcap, void *stream);
/* The exchange in DEVICE memory instead (BASELINE north_star: the gather exchange "over xGMI"): rank `rank` allocates ITS area
 * (fine-grained device memory), exports it (handle_out: smgpu_gather_xchg_ipc_handle_size() bytes, a hipIpcMemHandle_t), the
 * handles travel by one all-gather, and every rank opens its peers' (world <= 16, the ranks of one node).  A rank's loop kernel
 * writes its own area and polls the others'.  smgpu_gather_launch_shared takes either kind. */
SmgpuGatherXchg *smgpu_gather_xchg_new_device(uint32_t world, uint32_t rank, uint64_t rowcap);
uintptr_t smgpu_gather_xchg_ipc_handle_size(void);
void smgpu_gather_xchg_ipc_export(const SmgpuGatherXchg *xchg, uint8_t *handle_out);
void smgpu_gather_xchg_ipc_open(SmgpuGatherXchg *xchg, uint32_t peer_rank, const uint8_t *handle);
bool smgpu_gather_xchg_is_device(const SmgpuGatherXchg *xchg);
bool smgpu_gather_launch_shared(SmgpuGather *ptr, SmgpuGatherXchg *xchg, uint32_t rank, uint32_t run_id, uint32_t n_wg, void *stream);
/* Test support: `n_wg` workgroups that keep `lds_bytes` of LDS each and spin for `micros` microseconds on `stream` -- "somebody
 * else's kernel holds CUs of this device" (the resident gather loop must step aside for the two-kernel rounds, not fail). */
void smgpu_debug_hold_cus(uint32_t n_wg, uint32_t lds_bytes, uint64_t micros, void *stream);
uint64_t smgpu_gather_longest_row(const SmgpuGather *ptr);   /* hashes in the shard's longest row: stride >= 3 + the longest row of any shard */
void smgpu_gather_topk_export_raw(SmgpuGather *ptr, uint64_t *d_records, uint32_t k, uint64_t stride, void *stream);
void smgpu_gather_cands_load_raw(SmgpuGather *ptr, const uint64_t *d_records, uint32_t n_records, uint64_t stride,
                                 void *stream);
void smgpu_gather_replay_raw(SmgpuGather *ptr, uint32_t rounds, void *stream);
uint64_t smgpu_gather_poll(SmgpuGather *ptr, bool *done, void *stream);
uint64_t smgpu_gather_results(SmgpuGather *ptr, uint64_t *out_index, uint64_t *out_isect, uint64_t cap, void *stream);

/* Gather primitives (src/sourmash/index/__init__.py:735-909):
 *   overlap:  d_overlap[d] = |Q ∩ D_d| (op 0, CounterGather.add) or
 *             d_overlap[d] -= |Q ∩ D_d| saturating (op 1, CounterGather.consume)
 *   argmax:   *d_best = max(*d_best, (count << 32) | ~(index_base + d)): highest count,
 *             ties to the lowest index (Counter.most_common order) -- one u64 MAX
 *             all-reduce across GPUs picks the global winner.
 *   intersect: sorted list Q ∩ R into d_out, size into *d_n (device u64). */
void smgpu_overlap_raw(const uint64_t *d_query, uint64_t nq, const uint64_t *d_hashes, const uint64_t *d_offsets,
                       uint64_t ndb, uint64_t *d_overlap, int32_t op, void *stream);
void smgpu_argmax_raw(const uint64_t *d_overlap, uint64_t ndb, uint64_t index_base, uint64_t *d_best, void *stream);
uint64_t smgpu_intersect_workspace_bytes(uint64_t n);
void smgpu_intersect_raw(const uint64_t *d_a, uint64_t na, const uint64_t *d_b, uint64_t nb, uint64_t *d_out,
                         uint64_t *d_n, void *d_workspace, uint64_t workspace_bytes, void *stream);
/* query <- query minus match (src/sourmash/search.py:915-919): sorted set difference */
void smgpu_subtract_raw(const uint64_t *d_a, uint64_t na, const uint64_t *d_b, uint64_t nb, uint64_t *d_out,
                        uint64_t *d_n, void *d_workspace, uint64_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SOURMASH_AMD_H_INCLUDED */
