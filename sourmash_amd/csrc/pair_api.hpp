// Internal launchers of pair_ops.hip (raw device pointers, no allocation).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace smg {

// sums layout (device, 4 x u64, zeroed by the caller):
//   [0] |A ∩ B|   [1] sum abundA*abundB over matches   [2] sum abundA^2   [3] sum abundB^2
hipError_t pair_match_launch(const uint64_t* A, uint64_t na, const uint64_t* B, uint64_t nb, const uint64_t* abA,
                             const uint64_t* abB, uint8_t* flags, unsigned long long* sums, int invert,
                             hipStream_t stream);  // invert: flag hashes of A NOT in B
// |A ∩ B| of two device-resident sorted sketches in ONE small launch whose result goes straight to a host-visible slot:
// slot[1] = count, then slot[0] = seq with system-scope release (the host polls slot[0]).  nb <= PAIR_SMALL_MAX: B is
// staged in LDS and A's hashes are binary-searched there.
constexpr uint64_t PAIR_SMALL_MAX = 8000;
hipError_t pair_count_small_launch(const uint64_t* A, uint64_t na, const uint64_t* B, uint64_t nb,
                                   unsigned long long* host_slot, unsigned long long seq, hipStream_t stream);
hipError_t sumsq_launch(const uint64_t* a, uint64_t n, unsigned long long* dst, hipStream_t stream);
hipError_t num_rank_launch(const uint64_t* I, uint64_t ni, const uint64_t* A, uint64_t na, const uint64_t* B,
                           uint64_t nb, uint64_t num, unsigned long long* dst, hipStream_t stream);
size_t select_temp_bytes(uint64_t n);
hipError_t select_flagged(const uint64_t* in, const uint8_t* flags, uint64_t n, uint64_t* out, uint64_t* d_n_out,
                          void* temp, size_t temp_bytes, hipStream_t stream);
// op 0: overlap[d] = |Q ∩ D_d| ; op 1: overlap[d] -= |Q ∩ D_d| (saturating, rows at 0 skipped)
hipError_t overlap_vector_launch(const uint64_t* Q, uint64_t nq, const uint64_t* hashes, const uint64_t* offsets,
                                 uint64_t ndb, unsigned long long* overlap, int op, hipStream_t stream);
// CSR row gather: row i of the destination = row rows[i] of the source (dst_off = prefix sums of the row lengths)
hipError_t copy_rows_launch(const uint64_t* src, const uint64_t* src_off, const uint64_t* rows, uint64_t n_rows,
                            const uint64_t* dst_off, uint64_t* dst, hipStream_t stream);
// *best = max(*best, (count << 32) | ~(index_base + d)) over rows with count > 0
hipError_t argmax_launch(const unsigned long long* overlap, uint64_t ndb, uint64_t index_base,
                         unsigned long long* best, hipStream_t stream);

}  // namespace smg
