// sigjson.hip -- the hash arrays of signature JSON turned into u64 on the device.
//
// What the reference does here: serde_json deserialises every signature into a struct, `mins` into a Vec<u64>
// (src/core/src/signature.rs:569-659, src/core/src/sketch/minhash.rs:103-184); loading a collection of 10^4 - 10^5 sketches is
// digit parsing (~85 KB of decimal text per sketch of 5,000 hashes) on the host's cores.  Here the inflated JSON text stays in
// HBM (gunzip.hip leaves it there) and
//   sj_spans_kernel   a wavefront per document finds the `"mins":[ ... ]` and `"abundances":[ ... ]` arrays: where they begin and end,
//                     how many commas they hold, whether they hold anything but digits, commas and white space
//   sj_parse_kernel   a wavefront per `mins` array: the array is cut into 64 pieces, a prefix sum of the pieces' comma counts
//                     gives every number its index, every lane parses the numbers that begin in its piece; the values are
//                     checked for strictly ascending order (minhash.rs:161-171 sorts on load: such a document goes to the host) and
//                     counted against the down-sampling threshold
//   sj_take_bytes     everything OUTSIDE the arrays (a few hundred bytes a document) packed for the host, which reads the
//                     metadata with the same scanner as before (collection.hpp) -- the arrays replaced by their index
//   sj_take_u64       the kept prefix of every selected array -> its row of the CSR
// Anything unusual in an array (a float, a minus sign, unsorted or repeated values, more arrays than slots) flags the document
// and the host parses it as before: the result is the host's result or an error, never something else.
#include <hip/hip_runtime.h>
#include "sigjson_api.hpp"

namespace smg {

namespace {

__device__ __forceinline__ bool is_ws(uint32_t c) { return c == ' ' || c == '\n' || c == '\r' || c == '\t'; }

// first position >= from (< len) whose byte satisfies pred, by 64-byte tiles; len if none.  Uniform result.
template <class Pred>
__device__ __forceinline__ uint64_t find_first(const uint8_t* t, uint64_t from, uint64_t len, Pred pred) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t p = from; p < len; p += 64) {
        const uint64_t i = p + lane;
        const bool hit = i < len && pred(i);
        const uint64_t m = __builtin_amdgcn_ballot_w64(hit);
        if (m) return p + (uint64_t)__builtin_ctzll(m);
    }
    return len;
}

__device__ __forceinline__ bool starts_with(const uint8_t* t, uint64_t i, uint64_t len, const char* key, uint32_t klen) {
    if (i + klen > len) return false;
    for (uint32_t k = 0; k < klen; ++k) if (t[i + k] != (uint8_t)key[k]) return false;
    return true;
}

__global__ __launch_bounds__(64) void sj_spans_kernel(const uint8_t* __restrict__ base, const SjDoc* __restrict__ docs, uint32_t n_docs,
                                                      SjSpan* __restrict__ spans, uint32_t* __restrict__ doc_flags) {
    const uint32_t d = blockIdx.x;
    if (d >= n_docs) return;
    const SjDoc doc = docs[d];
    const uint8_t* t = base + doc.off;
    const uint64_t len = doc.len;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t n = 0, flags = 0;
    uint64_t pos = 0;
    while (pos < len) {
        // the next key that opens an array we take: "mins" or "abundances"
        const uint64_t m = find_first(t, pos, len, [&](uint64_t i) {
            return t[i] == '"' && (starts_with(t, i, len, "\"mins\"", 6) || starts_with(t, i, len, "\"abundances\"", 12));
        });
        if (m >= len) break;
        const uint32_t kind = t[m + 1] == 'm' ? SJ_MINS : SJ_ABUND;
        uint64_t q = m + (kind == SJ_MINS ? 6 : 12);
        while (q < len && is_ws(t[q])) ++q;
        if (q >= len || t[q] != ':') { pos = m + 1; continue; }
        ++q;
        while (q < len && is_ws(t[q])) ++q;
        if (q >= len || t[q] != '[') { pos = m + 1; continue; }       // (abundances may be null)
        const uint64_t s = q + 1;
        // its end, its commas, and whether it holds anything but digits, commas and white space
        uint64_t e = len;
        uint32_t commas = 0, odd = 0, digits = 0;
        for (uint64_t p = s; p < len; p += 64) {
            const uint64_t i = p + lane;
            const uint32_t c = i < len ? t[i] : (uint32_t)']';
            const uint64_t close = __builtin_amdgcn_ballot_w64(c == ']');
            const uint64_t upto = close ? (1ull << __builtin_ctzll(close)) - 1ull : ~0ull;     // lanes in front of the first ']'
            commas += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(c == ',') & upto);
            digits |= (__builtin_amdgcn_ballot_w64(c >= '0' && c <= '9') & upto) != 0ull;
            odd |= (__builtin_amdgcn_ballot_w64(!(c == ',' || (c >= '0' && c <= '9') || is_ws(c) || c == ']')) & upto) != 0ull;
            if (close) { e = p + (uint64_t)__builtin_ctzll(close); break; }
        }
        if (e >= len) { flags |= SJ_DOC_ODD; break; }                 // no closing bracket
        if (n >= SJ_MAX_SPANS) { flags |= SJ_DOC_ODD; break; }
        if (lane == 0) {
            SjSpan sp;
            sp.begin = s; sp.end = e;
            sp.n_values = digits ? commas + 1u : 0u;
            sp.kind = kind;
            sp.flags = (odd ? SJ_SPAN_ODD : 0u) | (!digits && commas ? SJ_SPAN_ODD : 0u);
            sp.pad = 0;
            spans[(uint64_t)d * SJ_MAX_SPANS + n] = sp;
        }
        ++n;
        pos = e + 1;
    }
    if (lane == 0) doc_flags[d] = flags | n;
}

// A wavefront per array.  The text goes through LDS 4 KB at a time (coalesced 16-byte loads; a lane reading its own stretch of
// the array byte by byte from HBM was one dependent trip to memory per byte: 17 ms for 6,000 arrays): lane l owns bytes
// [64 l, 64 l + 64) of the chunk, counts its commas, a prefix sum gives every number its index, and the
// lane parses the numbers that begin behind its commas -- they may run on into the next lane's bytes or the 64 bytes read
// beyond the chunk.  Order and the down-sampling count are taken from the written values afterwards.
constexpr uint32_t SJ_CHUNK = 4096, SJ_AHEAD = 64;

__global__ __launch_bounds__(64) void sj_parse_kernel(const uint8_t* __restrict__ base, const SjParse* __restrict__ jobs, uint32_t n_jobs,
                                                      uint64_t* __restrict__ values, SjParsed* __restrict__ results, uint64_t keep_max) {
    __shared__ uint4 buf4[(SJ_CHUNK + SJ_AHEAD + 16) / 16 + 1];
    uint8_t* buf = reinterpret_cast<uint8_t*>(buf4);
    const uint32_t j = blockIdx.x;
    if (j >= n_jobs) return;
    const SjParse job = jobs[j];
    const uint8_t* t = base + job.text_off;
    const uint64_t len = job.len;                                     // bytes between '[' and ']'
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t* out = values + job.value_off;
    uint32_t bad = 0;
    uint64_t index = 0;                                               // numbers in front of the chunk
    for (uint64_t c0 = 0; c0 < len; c0 += SJ_CHUNK) {
        // bytes [c0, c0 + SJ_CHUNK + SJ_AHEAD) of the array -> buf[shift ...], from the 16-byte line they begin in
        const uintptr_t addr = reinterpret_cast<uintptr_t>(t + c0);
        const uint32_t shift = (uint32_t)(addr & 15u);
        const uint4* src = reinterpret_cast<const uint4*>(addr - shift);
        const uint64_t avail = len - c0;                              // bytes of the array from c0 on
        const uint32_t want = (uint32_t)(avail < SJ_CHUNK + SJ_AHEAD ? avail : SJ_CHUNK + SJ_AHEAD);
        const uint32_t lines = (shift + want + 15u) / 16u;
        __syncthreads();
        for (uint32_t l = lane; l < lines; l += 64u) buf4[l] = src[l];
        __syncthreads();
        const uint8_t* b = buf + shift;                               // b[i] = byte c0 + i of the array, i < want
        const uint32_t in_chunk = (uint32_t)(avail < SJ_CHUNK ? avail : SJ_CHUNK);
        const uint32_t p0 = lane * 64u, p1 = p0 + 64u < in_chunk ? p0 + 64u : in_chunk;
        uint32_t commas = 0;
        for (uint32_t i = p0; i < p1; ++i) commas += b[i] == ',';
        uint32_t incl = commas;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(incl, o); if (lane >= (uint32_t)o) incl += u; }
        const uint32_t chunk_commas = __shfl(incl, 63);
        uint64_t k = index + (incl - commas);                         // commas in front of this lane's bytes = index of the number open there
        auto number = [&](uint32_t from, uint64_t idx) {              // the number that begins at b[from] (white space allowed around it)
            uint32_t i = from;
            while (i < want && is_ws(b[i])) ++i;
            uint64_t v = 0;
            uint32_t nd = 0;
            while (i < want && b[i] >= '0' && b[i] <= '9') {
                if (nd >= 19 && (v > 1844674407370955161ull || (v == 1844674407370955161ull && b[i] > '5'))) bad = 1;   // beyond 2^64 - 1
                v = v * 10 + (uint64_t)(b[i] - '0');
                ++nd;
                ++i;
            }
            while (i < want && is_ws(b[i])) ++i;
            if (nd == 0 || nd > 20) bad = 1;
            if (i < want ? b[i] != ',' : c0 + i < len) bad = 1;       // ends at a comma, or at the end of the array (not at the end of what was read)
            if (idx < job.n_values) out[idx] = v;
        };
        if (c0 == 0 && lane == 0 && job.n_values) number(0, 0);
        for (uint32_t i = p0; i < p1; ++i)
            if (b[i] == ',') { ++k; number(i + 1, k); }
        index += chunk_commas;
    }
    // order (ascending, no repeats: minhash.rs:161-171 would sort -- such an array is the host's) and the down-sampling count
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __syncthreads();
    uint32_t kept = 0;
    for (uint64_t i = lane; i < job.n_values; i += 64u) {
        const uint64_t v = out[i];
        kept += v <= keep_max;
        if (i + 1 < job.n_values && out[i + 1] <= v) bad = 1;
    }
    for (int o = 32; o; o >>= 1) { kept += __shfl_xor(kept, o); bad |= __shfl_xor(bad, o); }
    if (lane == 0) {
        SjParsed r;
        r.n_kept = kept;
        r.flags = bad ? SJ_SPAN_ODD : 0u;
        results[j] = r;
    }
}

__global__ __launch_bounds__(256) void sj_take_bytes_kernel(const uint8_t* __restrict__ base, const SjPiece* __restrict__ pieces, uint8_t* __restrict__ out) {
    const SjPiece p = pieces[blockIdx.x];
    for (uint64_t i = threadIdx.x; i < p.n; i += 256u) out[p.dst + i] = base[p.src + i];
}

__global__ __launch_bounds__(256) void sj_take_u64_kernel(const uint64_t* __restrict__ values, const SjPiece* __restrict__ pieces, uint64_t* __restrict__ out) {
    const SjPiece p = pieces[blockIdx.x];
    for (uint64_t i = threadIdx.x; i < p.n; i += 256u) out[p.dst + i] = values[p.src + i];
}

}  // namespace

hipError_t sj_spans_launch(const uint8_t* d_base, const SjDoc* d_docs, uint32_t n_docs, SjSpan* d_spans, uint32_t* d_doc_flags, hipStream_t stream) {
    if (n_docs == 0) return hipSuccess;
    hipLaunchKernelGGL(sj_spans_kernel, dim3(n_docs), dim3(64), 0, stream, d_base, d_docs, n_docs, d_spans, d_doc_flags);
    return hipGetLastError();
}

hipError_t sj_parse_launch(const uint8_t* d_base, const SjParse* d_jobs, uint32_t n_jobs, uint64_t* d_values, SjParsed* d_results, uint64_t keep_max,
                           hipStream_t stream) {
    if (n_jobs == 0) return hipSuccess;
    hipLaunchKernelGGL(sj_parse_kernel, dim3(n_jobs), dim3(64), 0, stream, d_base, d_jobs, n_jobs, d_values, d_results, keep_max);
    return hipGetLastError();
}

hipError_t sj_take_bytes_launch(const uint8_t* d_base, const SjPiece* d_pieces, uint32_t n, uint8_t* d_out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(sj_take_bytes_kernel, dim3(n), dim3(256), 0, stream, d_base, d_pieces, d_out);
    return hipGetLastError();
}

hipError_t sj_take_u64_launch(const uint64_t* d_values, const SjPiece* d_pieces, uint32_t n, uint64_t* d_out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(sj_take_u64_kernel, dim3(n), dim3(256), 0, stream, d_values, d_pieces, d_out);
    return hipGetLastError();
}

}  // namespace smg
