// gunzip.hip -- gzip members inflated on the device: the kernels of the scheme in inflate_core.hpp.
//
// What the reference does here: screed / needletail inflate the .fna.gz on one host thread and hand records to the sketcher
// (src/sourmash/command_sketch.py:697,746-768; src/core/benches/compute.rs:35-38).  Here the compressed bytes go over PCIe
// (a third of the inflated size for DNA) and every block of the member is decoded at once:
//   gz_scan_kernel      a lane per input byte tests its 8 bit positions for a block header       (VALU: ~25 instructions per position)
//   gz_validate_kernel  a lane per survivor reads the whole header                                (a few thousand lanes)
//   gz_pass1_kernel     a wavefront per candidate decodes the blocks of its run into 32-bit records (issue-bound: ~40 instructions a symbol, every CU full)
//   gz_pass2_kernel     a wavefront per run of the chain expands its records, 64 output positions at a time (gathers from its own output)
//   gz_tails_a/b_kernel the last 32 KB of every run: rewritten per group of runs, then the groups' windows in order (latency: ~2 sqrt(runs) steps)
//   gz_resolve_kernel   symbols -> bytes against a 32 KB window in LDS                            (HBM: 2 B in, 1 B out per byte)
//   gz_crc_kernel       CRC-32 of 64 KB chunks, a kilobyte per lane, joined by polynomial shifts  (LDS table lookups)
// The decoders are the shared header's code run uniformly by all 64 lanes: the tables sit in LDS (broadcast reads), the lanes
// differ only in the sink's gathers and stores.
#include <hip/hip_runtime.h>
#include "gunzip_api.hpp"
#include "inflate_core.hpp"

namespace smg {

using namespace inf;

namespace {

constexpr uint32_t GZ_SEGMENTS = 256;

// Every bit position of the input: is it the start of a dynamic-Huffman, non-final block?  The cheap part (what
// inf::plausible_prefix states): header bits, then the Kraft sum of the code-length code -- its up to 19 lengths of 3 bits
// looked up three at a time in a 512-entry table of partial sums.  One lane per input byte, its 8 bit positions in turn.
// (survivors are appended to one of GZ_SEGMENTS lists, by block number: half a million appends to ONE counter take 5 ms of
// the L2's atomic unit; spread over 256 counters they vanish)
__global__ __launch_bounds__(256) void gz_scan_kernel(const uint32_t* __restrict__ words, uint64_t n_bytes, uint64_t* __restrict__ surv,
                                                      unsigned long long* counts, uint64_t seg_cap) {
    __shared__ uint8_t kraft3[512];
    for (uint32_t i = threadIdx.x; i < 512u; i += 256u) {
        uint32_t k = 0;
        for (uint32_t f = 0; f < 3u; ++f) {
            const uint32_t l = (i >> (3u * f)) & 7u;
            if (l) k += 128u >> l;
        }
        kraft3[i] = (uint8_t)k;                                       // <= 192
    }
    __syncthreads();
    const uint64_t b = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (b >= n_bytes) return;
    const uint64_t w = b >> 2;
    const uint32_t sh0 = (uint32_t)(b & 3u) * 8u;
    const uint32_t w0 = words[w], w1 = words[w + 1], w2 = words[w + 2], w3 = words[w + 3];
#pragma unroll
    for (uint32_t s = 0; s < 8; ++s) {
        const uint32_t sh = sh0 + s;                                  // 0 .. 31
        const uint32_t s0 = __builtin_amdgcn_alignbit(w1, w0, sh);    // bits [0, 32) of the candidate header
        if ((s0 & 7u) != 4u) continue;                                // BFINAL = 0, BTYPE = 10b
        if (((s0 >> 3) & 31u) > 29u || ((s0 >> 8) & 31u) > 29u) continue;
        const uint32_t hclen = ((s0 >> 13) & 15u) + 4u;
        const uint32_t s1 = __builtin_amdgcn_alignbit(w2, w1, sh), s2 = __builtin_amdgcn_alignbit(w3, w2, sh);
        uint64_t f = ((uint64_t)(s0 >> 17)) | ((uint64_t)s1 << 15) | ((uint64_t)s2 << 47);     // the lengths: bits [17, 74)
        f &= (1ull << (3u * hclen)) - 1ull;                           // (12 .. 57 bits)
        uint32_t k = 0;
#pragma unroll
        for (uint32_t c = 0; c < 7u; ++c) k += kraft3[(uint32_t)(f >> (9u * c)) & 511u];
        if (k != 128u) continue;
        const uint32_t seg = blockIdx.x % GZ_SEGMENTS;
        const unsigned long long at = atomicAdd(&counts[8 + seg], 1ull);
        if (at < seg_cap) surv[(uint64_t)seg * seg_cap + at] = b * 8 + s;
    }
}

// a lane's 128-byte table inside a tile shared by the 64 lanes of the block: byte i of lane l at [i][l]
struct TileTab {
    uint8_t* col;
    __device__ __forceinline__ uint32_t get(uint32_t i) const { return col[i * 64u]; }
    __device__ __forceinline__ void set(uint32_t i, uint32_t v) { col[i * 64u] = (uint8_t)v; }
};

__global__ __launch_bounds__(64) void gz_validate_kernel(const uint32_t* __restrict__ words, uint64_t n_bytes, const uint64_t* __restrict__ surv,
                                                         uint64_t* __restrict__ valid, unsigned long long* counts, uint64_t cap) {
    __shared__ uint8_t tile[128 * 64];
    const uint64_t i = (uint64_t)blockIdx.x * 64u + threadIdx.x;
    const uint64_t seg_cap = cap / GZ_SEGMENTS;
    const uint64_t seg = i / seg_cap, slot = i % seg_cap;
    if (seg >= GZ_SEGMENTS) return;
    const unsigned long long filled = counts[8 + seg];
    if (slot == 0) {                                                  // [0]: survivors in all, [2]: the fullest list (beyond seg_cap: some were lost)
        atomicAdd(&counts[0], filled);
        atomicMax(&counts[2], filled);
    }
    if (slot >= (filled < seg_cap ? filled : seg_cap)) return;
    TileTab tab{tile + threadIdx.x};
    const uint64_t bit = surv[seg * seg_cap + slot];
    if (!valid_dynamic_header(words, bit, n_bytes * 8, tab)) return;
    const unsigned long long at = atomicAdd(&counts[1], 1ull);
    if (at < cap) valid[at] = bit;
}

__global__ __launch_bounds__(64, 6) void gz_pass1_kernel(const uint32_t* __restrict__ words, const GzCand* __restrict__ cands, uint32_t n,
                                                      uint32_t* __restrict__ rec, GzRunResult* __restrict__ res) {
    __shared__ Scratch S;
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    const GzCand c = cands[i];
    RecordSink sink;
    sink.rec = rec + c.bit;
    sink.cap = c.rec_cap;
    const RunResult r = decode_run(words, c.bit, c.limit_bit, S, sink, MAX_RUN_BYTES);
    if (threadIdx.x == 0) {
        GzRunResult o;
        o.end_bit = r.end_bit; o.out_len = r.out_len; o.status = r.status; o.n_records = r.n_records;
        res[i] = o;
    }
}

__global__ __launch_bounds__(64) void gz_pass2_kernel(const uint32_t* __restrict__ words, const uint32_t* __restrict__ rec,
                                                   const GzRunDesc* __restrict__ runs, uint32_t n, uint16_t* sym, GzRunResult* __restrict__ res) {
    __shared__ ExpandScratch X;
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    const GzRunDesc d = runs[i];
    Expander ex;
    ex.out = sym + d.out_off;
    ex.cap = d.out_len;
    ex.bytes = reinterpret_cast<const uint8_t*>(words);
    ex.no_window = d.first_of_member != 0;
    ex.run(rec + d.bit, d.n_records, X);
    if (threadIdx.x == 0) {
        GzRunResult o;
        o.end_bit = 0; o.out_len = ex.g0; o.status = ex.bad ? RUN_BAD_DISTANCE : RUN_OK; o.n_records = d.n_records;
        res[i] = o;
    }
}

// The last 32 KB of every run, in two levels (a member of thousands of runs would otherwise be thousands of dependent steps).
// A member's runs are dealt into groups of consecutive runs.
//   gz_tails_a_kernel  one workgroup per GROUP, its runs in order: the tail symbols are rewritten so that what they still
//                      reference lies in the 32 KB in front of the GROUP (MARK | m' = position group_start - 32768 + m') --
//                      ring[p & 32767] holds the rewritten symbol of position p
//   gz_tails_b_kernel  one workgroup per MEMBER, its groups in order: the 32 KB in front of every group become bytes
// after which every other tail symbol is one lookup away from a byte (gz_resolve_kernel against the group's window) and so
// is, after that, everything in front of the tails (gz_resolve_kernel against the run's window).
__global__ __launch_bounds__(1024) void gz_tails_a_kernel(uint16_t* sym, const GzRunDesc* __restrict__ runs, const GzGroupDesc* __restrict__ groups) {
    __shared__ uint16_t ring[WIN];
    const GzGroupDesc g = groups[blockIdx.x];
    const uint32_t t = threadIdx.x;
    for (uint32_t i = t; i < WIN; i += 1024u) ring[(g.start + i) & (WIN - 1)] = (uint16_t)(MARK | i);
    __syncthreads();
    for (uint32_t r = 0; r < g.n_runs; ++r) {
        const GzRunDesc d = runs[g.run0 + r];
        const uint64_t S = d.out_off - g.base, E = S + d.out_len;
        const uint64_t from = d.out_len > WIN ? E - WIN : S;
        uint16_t val[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const uint64_t p = from + t + 1024u * (uint32_t)j;
            uint16_t v = 0;
            if (p < E) {
                v = sym[g.base + p];
                if (v & MARK) v = ring[(S + (uint64_t)(v & 0x7fffu)) & (WIN - 1)];
            }
            val[j] = v;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const uint64_t p = from + t + 1024u * (uint32_t)j;
            if (p < E) {
                ring[p & (WIN - 1)] = val[j];
                sym[g.base + p] = val[j];
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void gz_tails_b_kernel(const uint16_t* __restrict__ sym, uint8_t* out, const GzGroupDesc* __restrict__ groups,
                                                          const GzMemberDesc* __restrict__ members, uint32_t* err) {
    __shared__ uint8_t ring[WIN];
    const GzMemberDesc m = members[blockIdx.x];
    const uint32_t t = threadIdx.x;
    bool wrong = false;
    for (uint32_t gi = 0; gi + 1 < m.n_groups; ++gi) {
        const GzGroupDesc g = groups[m.group0 + gi];
        const uint64_t next = groups[m.group0 + gi + 1].start;        // the 32 KB in front of the next group ...
        const uint64_t lo = next > WIN ? next - WIN : 0;
        const uint64_t from = lo > g.start ? lo : g.start;            // ... as far as they are this group's
        uint8_t val[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const uint64_t p = from + t + 1024u * (uint32_t)j;
            uint8_t b = 0;
            if (p < next) {
                const uint16_t v = sym[m.base + p];
                if (v & MARK) {
                    const uint64_t q = g.start + (uint64_t)(v & 0x7fffu);   // position + 32768
                    if (q < WIN) wrong = true;
                    b = ring[q & (WIN - 1)];
                } else b = (uint8_t)v;
            }
            val[j] = b;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const uint64_t p = from + t + 1024u * (uint32_t)j;
            if (p < next) {
                ring[p & (WIN - 1)] = val[j];
                out[m.base + p] = val[j];
            }
        }
        __syncthreads();
    }
    if (wrong) atomicOr(&err[blockIdx.x], 1u);
}

__global__ __launch_bounds__(256) void gz_resolve_kernel(const uint16_t* __restrict__ sym, uint8_t* out, const GzPiece* __restrict__ pieces, uint32_t* err) {
    __shared__ uint32_t ring32[WIN / 4];
    uint8_t* ring = reinterpret_cast<uint8_t*>(ring32);
    const GzPiece pc = pieces[blockIdx.x];
    const uint32_t t = threadIdx.x;
    const uint64_t S = pc.run_start;
    const uint64_t lo = S >= WIN ? S - WIN : 0;
    {   // the window [lo, S): whole dwords, then the ragged ends
        const uint64_t lo4 = (lo + 3) & ~3ull, s4 = S & ~3ull;
        for (uint64_t p = lo4 + 4ull * t; p < s4; p += 4ull * 256u)
            ring32[(p & (WIN - 1)) >> 2] = *reinterpret_cast<const uint32_t*>(out + pc.base + p);
        if (t < 4) {
            const uint64_t p = lo + t;
            if (p < lo4 && p < S) ring[p & (WIN - 1)] = out[pc.base + p];
            const uint64_t q = (s4 > lo4 ? s4 : lo4) + t;
            if (q < S) ring[q & (WIN - 1)] = out[pc.base + q];
        }
    }
    __syncthreads();
    bool wrong = false;
    auto one = [&](uint64_t p) {
        const uint16_t v = sym[pc.base + p];
        uint8_t b;
        if (v & MARK) {
            const uint64_t q = S + (uint64_t)(v & 0x7fffu);
            if (q < WIN) wrong = true;
            b = ring[q & (WIN - 1)];
        } else b = (uint8_t)v;
        out[pc.base + p] = b;
    };
    const uint64_t f8 = (pc.from + 7) & ~7ull, t8 = pc.to & ~7ull;
    if (f8 >= t8) {
        for (uint64_t p = pc.from + t; p < pc.to; p += 256u) one(p);
    } else {
        if (t < 8) {
            if (pc.from + t < f8) one(pc.from + t);
            if (t8 + t < pc.to) one(t8 + t);
        }
        for (uint64_t p = f8 + 8ull * t; p < t8; p += 8ull * 256u) {
            const uint4 v4 = *reinterpret_cast<const uint4*>(sym + pc.base + p);
            const uint32_t vs[4] = {v4.x, v4.y, v4.z, v4.w};
            uint32_t o2[2] = {0, 0};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t v = (vs[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
                uint32_t b;
                if (v & MARK) {
                    const uint64_t q = S + (uint64_t)(v & 0x7fffu);
                    if (q < WIN) wrong = true;
                    b = ring[q & (WIN - 1)];
                } else b = v & 0xffu;
                o2[k >> 2] |= b << ((k & 3) * 8);
            }
            *reinterpret_cast<uint2*>(out + pc.base + p) = make_uint2(o2[0], o2[1]);
        }
    }
    if (wrong) atomicOr(&err[pc.member], 1u);
}

struct CrcShifts { uint32_t xp[64]; };                               // xp[k] = x^(8 * 1024 * k) mod p

__device__ __forceinline__ uint32_t crc_mul_dev(uint32_t a, uint32_t b) {
    uint32_t p = 0;
#pragma unroll 4
    for (uint32_t m = 1u << 31; m; m >>= 1) {
        if (a & m) p ^= b;
        b = b & 1u ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}

__global__ __launch_bounds__(64) void gz_crc_kernel(const uint8_t* __restrict__ data, const GzChunk* __restrict__ chunks, uint32_t* __restrict__ crcs,
                                                    CrcShifts sh) {
    __shared__ uint32_t T[256];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 256; i += 64) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = c & 1u ? (c >> 1) ^ 0xEDB88320u : c >> 1;
        T[i] = c;
    }
    __syncthreads();
    const GzChunk ch = chunks[blockIdx.x];
    const uint32_t lo = lane * 1024u < ch.len ? lane * 1024u : ch.len;
    const uint32_t hi = (lane + 1) * 1024u < ch.len ? (lane + 1) * 1024u : ch.len;
    uint32_t c = 0xffffffffu;
    const uint8_t* p = data + ch.off;
    uint32_t i = lo;
    for (; i + 16 <= hi; i += 16) {
        const uint4 v = *reinterpret_cast<const uint4*>(p + i);
        const uint32_t ws[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t x = ws[k];
#pragma unroll
            for (int b = 0; b < 4; ++b, x >>= 8) c = T[(c ^ x) & 0xffu] ^ (c >> 8);
        }
    }
    for (; i < hi; ++i) c = T[(c ^ p[i]) & 0xffu] ^ (c >> 8);
    c = hi > lo ? c ^ 0xffffffffu : 0u;                              // (the CRC of no bytes is 0)
    // shift by the bytes behind this lane's piece and fold
    uint32_t part;
    if (ch.len == 65536u) part = crc_mul_dev(sh.xp[63 - lane], c);
    else {
        uint32_t r = 1u << 31, sq = 1u << 23;
        for (uint32_t n = ch.len - hi; n; n >>= 1) {
            if (n & 1u) r = crc_mul_dev(r, sq);
            sq = crc_mul_dev(sq, sq);
        }
        part = crc_mul_dev(r, c);
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) part ^= __shfl_xor(part, o);
    if (lane == 0) crcs[blockIdx.x] = part;
}

__global__ __launch_bounds__(256) void gz_first_bytes_kernel(const uint8_t* __restrict__ out, const GzMemberDesc* __restrict__ members, uint32_t n,
                                                             uint8_t* __restrict__ first) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) first[i] = out[members[i].base];
}

}  // namespace

hipError_t gz_first_bytes_launch(const uint8_t* d_out, const GzMemberDesc* d_members, uint32_t n, uint8_t* d_first, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(gz_first_bytes_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_out, d_members, n, d_first);
    return hipGetLastError();
}

hipError_t gz_scan_launch(const uint32_t* words, uint64_t n_bytes, uint64_t* d_surv, uint64_t* d_valid, unsigned long long* d_counts,
                          uint64_t cap, hipStream_t stream) {
    if (n_bytes == 0) return hipSuccess;
    const uint64_t blocks = (n_bytes + 255) / 256;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gz_scan_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, words, n_bytes, d_surv, d_counts, cap / GZ_SEGMENTS);
    const uint64_t vblocks = (cap + 63) / 64;
    hipLaunchKernelGGL(gz_validate_kernel, dim3((unsigned)vblocks), dim3(64), 0, stream, words, n_bytes, (const uint64_t*)d_surv, d_valid, d_counts, cap);
    return hipGetLastError();
}

hipError_t gz_pass1_launch(const uint32_t* words, const GzCand* d_cands, uint32_t n, uint32_t* d_rec, GzRunResult* d_res, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(gz_pass1_kernel, dim3(n), dim3(64), 0, stream, words, d_cands, n, d_rec, d_res);
    return hipGetLastError();
}

hipError_t gz_pass2_launch(const uint32_t* words, const uint32_t* d_rec, const GzRunDesc* d_runs, uint32_t n, uint16_t* d_sym, GzRunResult* d_res,
                           hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(gz_pass2_kernel, dim3(n), dim3(64), 0, stream, words, d_rec, d_runs, n, d_sym, d_res);
    return hipGetLastError();
}

hipError_t gz_tails_launch(uint16_t* d_sym, uint8_t* d_out, const GzRunDesc* d_runs, const GzGroupDesc* d_groups, uint32_t n_groups,
                           const GzMemberDesc* d_members, uint32_t n_members, uint32_t* d_err, hipStream_t stream) {
    if (n_members == 0 || n_groups == 0) return hipSuccess;
    hipLaunchKernelGGL(gz_tails_a_kernel, dim3(n_groups), dim3(1024), 0, stream, d_sym, d_runs, d_groups);
    hipLaunchKernelGGL(gz_tails_b_kernel, dim3(n_members), dim3(1024), 0, stream, (const uint16_t*)d_sym, d_out, d_groups, d_members, d_err);
    return hipGetLastError();
}

hipError_t gz_resolve_launch(const uint16_t* d_sym, uint8_t* d_out, const GzPiece* d_pieces, uint32_t n_pieces, uint32_t* d_err, hipStream_t stream) {
    if (n_pieces == 0) return hipSuccess;
    hipLaunchKernelGGL(gz_resolve_kernel, dim3(n_pieces), dim3(256), 0, stream, d_sym, d_out, d_pieces, d_err);
    return hipGetLastError();
}

hipError_t gz_crc_launch(const uint8_t* d_out, const GzChunk* d_chunks, uint32_t n_chunks, uint32_t* d_crc, hipStream_t stream) {
    if (n_chunks == 0) return hipSuccess;
    static const CrcShifts shifts = [] {
        CrcShifts s;
        for (int k = 0; k < 64; ++k) s.xp[k] = crc_xpow8((uint64_t)1024 * (uint64_t)k);
        return s;
    }();
    hipLaunchKernelGGL(gz_crc_kernel, dim3(n_chunks), dim3(64), 0, stream, d_out, d_chunks, d_crc, shifts);
    return hipGetLastError();
}

}  // namespace smg
