// Internal launchers of sigjson.hip: the hash arrays of signature JSON parsed on the device (sigload.hpp is the host side).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace smg {

constexpr uint32_t SJ_MAX_SPANS = 8;        // arrays looked at per document; a document with more is the host's
constexpr uint32_t SJ_MINS = 0, SJ_ABUND = 1;
constexpr uint32_t SJ_SPAN_ODD = 1;         // an array holds something the device does not parse
constexpr uint32_t SJ_DOC_ODD = 0x80000000u;   // doc_flags: this bit, or the number of arrays found in the low bits

struct SjDoc { uint64_t off, len; };        // a document's text inside the text block
struct SjSpan {
    uint64_t begin, end;                    // the array's bytes between '[' and ']', relative to the document
    uint32_t n_values, kind, flags, pad;
};
struct SjParse {                            // one `mins` array to turn into numbers
    uint64_t text_off, len;                 // its bytes (inside the text block)
    uint64_t value_off;                     // where its values go
    uint64_t n_values;
};
struct SjParsed { uint32_t n_kept, flags; };    // values <= keep_max (they come first: the array ascends); SJ_SPAN_ODD: not ascending / not plain numbers
struct SjPiece { uint64_t src, dst, n; };

hipError_t sj_spans_launch(const uint8_t* d_base, const SjDoc* d_docs, uint32_t n_docs, SjSpan* d_spans, uint32_t* d_doc_flags, hipStream_t stream);
hipError_t sj_parse_launch(const uint8_t* d_base, const SjParse* d_jobs, uint32_t n_jobs, uint64_t* d_values, SjParsed* d_results, uint64_t keep_max,
                           hipStream_t stream);
hipError_t sj_take_bytes_launch(const uint8_t* d_base, const SjPiece* d_pieces, uint32_t n, uint8_t* d_out, hipStream_t stream);
hipError_t sj_take_u64_launch(const uint64_t* d_values, const SjPiece* d_pieces, uint32_t n, uint64_t* d_out, hipStream_t stream);

}  // namespace smg
