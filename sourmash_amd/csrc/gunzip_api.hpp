// Internal launchers of gunzip.hip: gzip members inflated on the device (raw device pointers, caller-provided tables).
// The scheme is described in inflate_core.hpp; gunzip.hpp is the host side that strings the launches together.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace smg {

constexpr size_t GZ_COUNTS = 8 + 256;

struct GzCand {                 // a block start to decode from (pass 1)
    uint64_t bit, limit_bit;    // limit: where its member's trailer begins
    uint64_t rec_cap;           // records it may write (at d_rec + bit): up to the next candidate's first bit
};
struct GzRunResult {
    uint64_t end_bit, out_len;
    uint32_t status, n_records; // inf::RUN_*
};
struct GzRunDesc {              // a run of the chain (pass 2, tails)
    uint64_t bit;               // its records begin at d_rec + bit
    uint64_t out_off;           // first symbol / byte of the run in the symbol and output buffers
    uint64_t out_len;
    uint32_t n_records;
    uint32_t first_of_member, member, pad;
};
struct GzMemberDesc {
    uint64_t base;              // first byte of the member in the output buffer
    uint32_t run0, n_runs;
    uint32_t group0, n_groups;
};
struct GzGroupDesc {            // consecutive runs of one member (tails)
    uint64_t base;              // the member's
    uint64_t start;             // first position of the group, relative to base
    uint32_t run0, n_runs;
};
struct GzPiece {                // positions [from, to) whose symbols reference the 32 KB in front of run_start (resolve)
    uint64_t base, run_start, from, to;   // base: the member's; the others relative to it
    uint32_t member, pad;
};
struct GzChunk {                // up to 64 KB of output (crc)
    uint64_t off;
    uint32_t len, pad;
};

// words[0, n_bytes / 4 + 256): the files' bytes, zero-padded.  Survivors of the cheap test go to d_surv (capacity cap, a multiple
// of 256: 256 lists of cap / 256), block starts that pass the full test to d_valid (capacity cap).  d_counts: GZ_COUNTS u64,
// zeroed by the caller; afterwards [0] survivors, [1] block starts, [2] the fullest survivor list: above cap / 256 (or [1] above
// cap) the lists are incomplete.
hipError_t gz_scan_launch(const uint32_t* words, uint64_t n_bytes, uint64_t* d_surv, uint64_t* d_valid, unsigned long long* d_counts,
                          uint64_t cap, hipStream_t stream);
// pass 1: candidate i decodes from its bit and leaves its symbols as records at d_rec + bit (d_rec: one u32 per BIT of the buffer)
hipError_t gz_pass1_launch(const uint32_t* words, const GzCand* d_cands, uint32_t n, uint32_t* d_rec, GzRunResult* d_res, hipStream_t stream);
// pass 2: the records of every run of the chain -> 16-bit symbols at the run's place (d_res[i].out_len = symbols written, status)
hipError_t gz_pass2_launch(const uint32_t* words, const uint32_t* d_rec, const GzRunDesc* d_runs, uint32_t n, uint16_t* d_sym, GzRunResult* d_res,
                           hipStream_t stream);
// d_err[member] |= 1: a symbol points in front of its member.  Rewrites the tail symbols of d_sym (see gunzip.hip).
hipError_t gz_tails_launch(uint16_t* d_sym, uint8_t* d_out, const GzRunDesc* d_runs, const GzGroupDesc* d_groups, uint32_t n_groups,
                           const GzMemberDesc* d_members, uint32_t n_members, uint32_t* d_err, hipStream_t stream);
hipError_t gz_resolve_launch(const uint16_t* d_sym, uint8_t* d_out, const GzPiece* d_pieces, uint32_t n_pieces, uint32_t* d_err, hipStream_t stream);
// d_first[i] = the first byte of member i's output (a member of no bytes: whatever lies at its base)
hipError_t gz_first_bytes_launch(const uint8_t* d_out, const GzMemberDesc* d_members, uint32_t n, uint8_t* d_first, hipStream_t stream);
// CRC-32 (as in the gzip trailer) of every chunk
hipError_t gz_crc_launch(const uint8_t* d_out, const GzChunk* d_chunks, uint32_t n_chunks, uint32_t* d_crc, hipStream_t stream);

}  // namespace smg
