// Internal C++ interface between the HIP translation units and the C-ABI layer.
// Everything here takes raw device pointers and a stream and never allocates.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace smg {

// ---- sketch.hip ---------------------------------------------------------------
// Append every hash h of a canonical DNA k-mer of d_seq[0,len) with 1 <= h <= thr
// to d_out (unordered, duplicates kept); *d_count += number appended (keeps
// counting past `cap`, entries past cap are dropped).  d_seq may have any alignment
// (an unaligned start costs nothing: the kernel realigns and blanks the prefix).  Any byte outside ACGTacgt kills the k-mers covering it.
hipError_t sketch_dna_launch(const uint8_t* d_seq, uint64_t len, uint32_t k, uint64_t seed, uint64_t thr,
                             uint64_t* d_out, unsigned long long* d_count, uint64_t cap, hipStream_t stream);
// d_out[i] = hash of the canonical k-mer starting at i for i in [0, n_kmers) (0 for k-mers
// covering a byte outside ACGTacgt).  d_out must be zeroed by the caller.
// k-mers longer than the register-window kernel holds (sketch_words.hip): k = 129 .. sketch_dna_max_k(), any k >= 16 accepted
// several ksizes of one stretch in one pass (sketch_multi.hip): the standard ksize sets only
struct SketchMultiOut { uint64_t thr; uint64_t* out; unsigned long long* count; uint64_t cap; };
bool sketch_dna_multi_supported(const uint32_t* ks, int n);
hipError_t sketch_dna_multi_launch(const uint8_t* d_new, uint64_t n_new, const uint32_t* ks, int n, uint64_t seed, const SketchMultiOut* outs,
                                   hipStream_t stream);
hipError_t sketch_dna_words_launch(const uint8_t* d_seq, uint64_t len, uint32_t k, uint64_t seed, uint64_t thr, uint64_t* d_out,
                                   unsigned long long* d_count, uint64_t cap, bool dense, hipStream_t stream);
uint32_t sketch_dna_max_k();
hipError_t kmer_hashes_launch(const uint8_t* d_seq, uint64_t len, uint32_t k, uint64_t seed, uint64_t* d_out,
                              uint64_t n_kmers, hipStream_t stream);
// *d_first = min(*d_first, position of the first byte outside ACGTacgt)
hipError_t first_invalid_launch(const uint8_t* d_seq, uint64_t len, unsigned long long* d_first, hipStream_t stream);

// ---- protein.hip (protein / dayhoff / hp sketches) ---------------------------------------------
// d_aa[i] = alphabet(upper(d_seq[i]))  (hash_function 2 protein, 3 dayhoff, 4 hp)
hipError_t residues_launch(const uint8_t* d_seq, uint64_t len, uint32_t hash_function, uint8_t* d_aa, hipStream_t stream);
// six-frame translation of DNA: segments frame0 fwd, frame0 rc, frame1 fwd, ... each followed by a 0xFF separator;
// d_aa must hold translated_bytes(len) bytes
uint64_t translated_bytes(uint64_t len);
hipError_t translate_launch(const uint8_t* d_seq, uint64_t len, uint32_t hash_function, uint8_t* d_aa, hipStream_t stream);
// hashes of the windows of k residues that do not touch a separator: appended (1 <= h <= thr; *d_count += n) or,
// dense, d_out[i] = hash of the window starting at i (d_out pre-zeroed, cap = number of entries)
hipError_t residue_windows_launch(const uint8_t* d_aa, uint64_t n, uint32_t k, uint64_t seed, uint64_t thr, uint64_t* d_out,
                                  unsigned long long* d_count, uint64_t cap, bool dense, hipStream_t stream);

// ---- device_sort.hip ------------------------------------------------------------
size_t sort_unique_temp_bytes(uint64_t n);
// many small lists sorted as one (device_sort.hip): dst[seg.dst + i] = src[seg.src + i] | (list number << hbits)
struct TagSegment { uint64_t src, dst, n; };
hipError_t tag_gather_launch(const uint64_t* d_src, const TagSegment* d_segs, uint32_t n_segs, uint64_t* d_dst, int hbits, hipStream_t stream);
// keys[0,n) -> sorted unique in out[0,*d_n_out); if d_counts != nullptr also the
// multiplicity of every unique key.  keys is clobbered.  `bits` = significant key bits.
hipError_t sort_unique(uint64_t* d_keys, uint64_t n, uint64_t* d_out, uint64_t* d_counts, uint64_t* d_n_out,
                       void* d_temp, size_t temp_bytes, int bits, hipStream_t stream);

// ---- synth.hip --------------------------------------------------------------------
hipError_t synth_dna_launch(uint8_t* d_out, uint64_t start, uint64_t n, uint64_t seed, uint64_t record_len,
                            hipStream_t stream);

// ---- compare.hip ------------------------------------------------------------------
// CSR of sorted unique u64 rows on device -> common[i][j] for i in [row_lo,row_hi), all j.
hipError_t compare_counts_launch(const uint64_t* d_hashes, const uint64_t* d_offsets, uint32_t n,
                                 uint32_t row_lo, uint32_t row_hi, uint32_t* d_common /*[(row_hi-row_lo)][n]*/,
                                 hipStream_t stream);
// Sharded all-pairs: row tiles (16 rows each) rb_first, rb_first + rb_stride, ... of the n x n problem,
// upper-triangle tiles only; d_common holds the owned tiles back to back ([rb_count * 16][n]).
hipError_t compare_blocks_launch(const uint64_t* d_hashes, const uint64_t* d_offsets, uint32_t n, uint32_t rb_first,
                                 uint32_t rb_stride, uint32_t rb_count, uint32_t* d_common, hipStream_t stream);
// m[j][i] = m[i][j] for i < j on a full n x n matrix
hipError_t symmetrize_launch(uint32_t* d_common, uint32_t n, hipStream_t stream);
hipError_t jaccard_from_counts_launch(const uint32_t* d_common, const uint64_t* d_offsets, uint32_t n,
                                      uint32_t row_lo, uint32_t row_hi, double* d_out, hipStream_t stream);

// ---- compare_ext.hip (bottom-k and abundance-weighted all-pairs) --------------------------------
// Bottom-k sketches (CSR of sorted rows, d_nums[i] = num of sketch i): d_common[i][j] = |A ∩ B ∩ bottom_num(A ∪ B)| with num of
// the lower index (minhash.rs:593-621; the diagonal holds the row sizes), d_union[i][j] = |bottom_num(A ∪ B)| and
// d_jaccard = common / max(1, union) with 1.0 on the diagonal (either may be null).  Full n x n matrices.
hipError_t compare_num_launch(const uint64_t* d_hashes, const uint64_t* d_offsets, const uint32_t* d_nums, uint32_t n,
                              uint32_t* d_common, uint32_t* d_union, double* d_jaccard, hipStream_t stream);
// Abundance-tracking sketches: d_prod[i][j] = sum over common hashes of abund_i * abund_j (u64, wrapping; the diagonal holds the
// row's sum of squares), d_sumsq[i] the same sums of squares, d_common the plain intersection sizes (minhash.rs:635-680).
// narrow: every abundance fits 32 bits (one v_mad_u64_u32 per common hash instead of a 64 x 64 product).
hipError_t compare_abund_launch(const uint64_t* d_hashes, const uint64_t* d_abunds, const uint64_t* d_offsets, uint32_t n,
                                bool narrow, uint32_t* d_common, unsigned long long* d_prod, unsigned long long* d_sumsq,
                                hipStream_t stream, uint64_t total = 0);

// abund_pairs.hip: the same two matrices (diagonals excluded: the caller's row kernel writes them) from joins of per-block lists
// sorted by hash; scratch from the library's arena.  hipErrorNotSupported: 2^32 elements or more -- the caller keeps the walk.
hipError_t abund_pairs_launch(const uint64_t* d_hashes, const uint64_t* d_abunds, const uint64_t* d_offsets, uint32_t n, uint64_t total,
                              bool narrow, uint32_t* d_common, unsigned long long* d_prod, hipStream_t stream);
// Lists with several scaled values: gather the first new_off[r + 1] - new_off[r] hashes of the rows starting at src_start[r] into
// a CSR of their own; then move the entries sub[a][b] of an m x m matrix over rows[] whose pair class max(class_of[rows[a]],
// class_of[rows[b]]) equals `cls` to out[rows[a]][rows[b]] of the n x n matrix.
hipError_t csr_prefix_gather_launch(const uint64_t* d_hashes, const uint64_t* d_src_start, const uint64_t* d_new_off, uint32_t m,
                                    uint64_t* d_out, hipStream_t stream);
hipError_t class_scatter_launch(const uint32_t* d_sub, uint32_t m, const uint32_t* d_rows, const uint32_t* d_class_of, uint32_t cls,
                                uint32_t n, uint32_t* d_out, hipStream_t stream);

// ---- bitindex.hip (dense compare path) ---------------------------------------------------------
hipError_t bitmap_build_launch(const uint64_t* d_hashes, const uint64_t* d_offsets, uint32_t n, const uint64_t* d_dict,
                               uint64_t U, uint32_t* d_bits, uint32_t words_per_row, hipStream_t stream);
// rows: 16-row tiles rb_first, rb_first + rb_stride, ... (rb_count); d_common [rb_count*16][n]; every column, or with
// upper_only every entry on or above the diagonal (entries below it are left as they were: the caller mirrors)
hipError_t bitmatrix_launch(const uint32_t* d_bits, uint32_t words_per_row, uint32_t n, uint32_t rb_first,
                            uint32_t rb_stride, uint32_t rb_count, uint32_t* d_common, hipStream_t stream, bool upper_only = false);

// ---- dictindex.hip (the compare index without a sort) -------------------------------------------
constexpr int DICT_BUCKETS = 512;          // hash-space buckets, one workgroup each
constexpr int DICT_MAX_DISTINCT = 1024;    // distinct hashes a bucket can hold; fuller collections use the sort below
size_t dict_scratch_bytes(uint32_t n);
// slice bounds, pass 1 (distinct hashes + holders per bucket, classified against `threshold`), scan.  The caller zeroes the
// first 256 bytes of d_scratch (the builder's parameters live there; d_out may point at byte 64 of it).  d_out (5 values,
// zero on entry): distinct hashes, frequent ones, rare pair increments, rare elements, buckets that overflowed
hipError_t dict_count_launch(const uint64_t* d_hashes, const uint64_t* d_offsets, uint32_t n, uint32_t threshold, void* d_scratch,
                             unsigned long long* d_out, hipStream_t stream);
// pass 2: bit rows (zeroed by the caller) and the rows of every rare hash (d_rare_end[p] = end of p's run)
hipError_t dict_emit_launch(const uint64_t* d_hashes, const uint64_t* d_offsets, uint32_t n, void* d_scratch, uint32_t* d_bits,
                            uint32_t words_per_row, uint32_t* d_rare_rows, uint32_t* d_rare_end, hipStream_t stream);

// ---- sparse_pairs.hip (inverted compare path) ---------------------------------------------------
size_t inverted_temp_bytes(uint64_t total);
hipError_t inverted_sort_launch(const uint64_t* d_hashes, const uint64_t* d_offsets, uint32_t n, uint64_t total,
                                uint64_t* d_keys_a, uint64_t* d_keys_b, uint32_t* d_rows_tmp, uint32_t* d_rows_sorted,
                                uint32_t* d_counts, uint64_t* d_n_runs, void* d_temp, size_t temp_bytes, hipStream_t stream);
// one read-back for the host: d_out[0] distinct hashes, d_out[1] frequent ones, d_out[2] rare pair increments (zeroed by
// the caller); flags / run offsets / frequent ranks sized for max_runs (= total elements) + 1 entries
hipError_t inverted_classify_launch(const uint32_t* d_counts, const uint64_t* d_n_runs, uint64_t max_runs, uint32_t threshold,
                                    uint32_t* d_freq_flag, uint64_t* d_run_off, uint64_t* d_freq_rank, unsigned long long* d_out,
                                    void* d_temp, size_t temp_bytes, hipStream_t stream);
hipError_t inverted_apply_launch(const uint64_t* d_run_off, const uint32_t* d_freq_flag, const uint64_t* d_freq_rank,
                                 uint64_t n_runs, uint64_t total, const uint32_t* d_rows_sorted, uint32_t* d_run_end,
                                 uint32_t* d_bits, uint32_t words_per_row, hipStream_t stream);
// common[local row][col] += rare-hash contributions, for the rows of the 16-row tiles rb_first, rb_first + rb_stride, ...
hipError_t rare_pairs_launch(const uint32_t* d_rows_sorted, const uint32_t* d_run_end, uint64_t total, uint32_t n,
                             uint32_t rb_first, uint32_t rb_stride, uint32_t rb_count, uint32_t* d_common, hipStream_t stream,
                             bool upper_only = false);

}  // namespace smg
