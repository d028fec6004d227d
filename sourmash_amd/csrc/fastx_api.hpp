// Internal launchers of fastx.hip (raw device pointers, caller-provided scratch).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace smg {

// scratch for one fastx_compact_launch over at most max_chunk bytes
size_t fastx_temp_bytes(uint64_t max_chunk);
// d_raw[0,n): the next piece of a FASTA (fastq == 0) or 4-line FASTQ (fastq == 1) file.  Sequence bytes and one
// separator byte per record go to d_out in order, their number to *d_n_out; *d_n_records += header lines seen.
// d_carry: 4 bytes chaining consecutive pieces (initialise to {1, 1, 0, 0} for FASTA, {3, 1, 0, 0} for FASTQ);
// d_state: unused since round 6 (may be null).  last_piece: nothing follows this piece -- the carry is left in d_carry[2, 4) and
// not copied forward (one small copy less in the stream: it matters when a file is a single piece of a few megabytes).
hipError_t fastx_compact_launch(const uint8_t* d_raw, uint64_t n, int fastq, uint8_t* d_carry, uint8_t* d_state,
                                uint8_t* d_out, unsigned long long* d_n_out, unsigned long long* d_n_records,
                                void* d_temp, size_t temp_bytes, hipStream_t stream, bool last_piece = false);
// d_dst[0,halo) = the last `halo` bytes of (d_src[-halo,0) ++ d_src[0,*d_n_new)): the k-1 bytes the next piece
// must see in front of its own (halo <= 256)
hipError_t fastx_halo_launch(const uint8_t* d_src, const unsigned long long* d_n_new, int halo, uint8_t* d_dst,
                             hipStream_t stream);

}  // namespace smg
