// fastx.hip -- FASTA / FASTQ record structure resolved on the GPU.
//
// The reference parses sequence files on the host, one record at a time (screed / needletail:
// src/sourmash/command_sketch.py:697,746-768, src/core/benches/compute.rs:35-38).  Here the host only moves raw
// file bytes into HBM; which bytes are sequence is decided on the device at memory bandwidth:
//   state      per byte, the kind of line it is on.  FASTA: the kind of the most recent line start (1 = sequence line, 2 = header
//              line: it starts with '>'); FASTQ: the number of line starts so far mod 4 (line 0 of every 4 is the header, line 1 the
//              sequence).
//   keep       a byte stays if it is on a sequence line and is not CR/LF, or if it is the first byte of a header
//              line ('>' / '@'): that one remains in the stream as the record separator -- it is outside ACGT, so
//              it kills exactly the k-mers that would span two records, which is what one add_sequence call per
//              record achieves
//   compact    order preserving, over 8 KiB blocks.
// Round 6: three launches, none of them per byte.  Rounds 1-5 ran a byte-wise rocPRIM scan for the states (one state byte written
// and read twice per input byte: 46 us of the 87 us a 4.6 MB genome's parse took).  A block's states depend on what lies in front
// of it only through ONE value -- the state of the byte before the block -- so
//   fx_summary_kernel   per block: what it does to that value (FASTA: the kind of its last line start, if any; FASTQ: its number of
//                       line starts) and its kept bytes and header lines as a function of it (FASTA: the part before the block's
//                       first line start counts only on a sequence line; FASTQ: one count per phase 0 .. 3);
//   fx_offsets_kernel   one workgroup walks the summaries: every block's entry state, kept bytes -> output offset, the records, the
//                       carry for the next piece;
//   fx_scatter_kernel   every block recomputes its flags from its entry state, compacts itself in LDS and writes its bytes out
//                       contiguously.
// Chunks are chained through a 4-byte carry (state of the last byte, "ended on a newline"), so a file streams through in pieces
// of any size.
#include <hip/hip_runtime.h>
#include <cstring>
#include "fastx_api.hpp"

namespace smg {

namespace {

constexpr int FX_THREADS = 256;
constexpr int FX_PER_THREAD = 32;                              // consecutive bytes per lane (two 16-byte loads)
constexpr int FX_BLOCK_BYTES = FX_THREADS * FX_PER_THREAD;     // 8 KiB per workgroup

// what a block does to the state that enters it, and what it keeps as a function of that state
struct BlockSum {
    uint32_t cnt[4];     // FASTA: [0] kept bytes whatever enters, [1] more if a sequence line enters; FASTQ: kept bytes by entry phase
    uint32_t hdr[4];     // FASTA: [0] header lines; FASTQ: header lines by entry phase
    uint32_t starts;     // FASTQ: line starts in the block
    uint32_t last_kind;  // FASTA: kind of the block's last line start (0: none)
};

// A lane's 32 bytes.  ls / nl / gt: bit j = byte j starts a line / is CR or LF / is '>'; bits at and above the lane's valid bytes are 0.
struct LaneBits { uint32_t ls, nl, gt, valid; };
__device__ __forceinline__ LaneBits lane_bits(const uint8_t* __restrict__ raw, uint64_t base, uint64_t n, uint8_t prev_nl_at_0, uint8_t* bytes) {
    LaneBits b{0, 0, 0, 0};
    if (base >= n) return b;
    if (base + FX_PER_THREAD <= n) {
        const uint4 r0 = *reinterpret_cast<const uint4*>(raw + base), r1 = *reinterpret_cast<const uint4*>(raw + base + 16);
        memcpy(bytes, &r0, 16); memcpy(bytes + 16, &r1, 16);
        b.valid = 0xffffffffu;
    } else {
        const int lim = (int)(n - base);
        for (int j = 0; j < FX_PER_THREAD; ++j) bytes[j] = j < lim ? raw[base + j] : (uint8_t)'\n';
        b.valid = (1u << lim) - 1u;                              // lim in 1 .. 31
    }
    bool prev_nl = base ? raw[base - 1] == '\n' : prev_nl_at_0 != 0;
#pragma unroll
    for (int j = 0; j < FX_PER_THREAD; ++j) {
        const uint8_t c = bytes[j];
        b.ls |= prev_nl ? (1u << j) : 0u;
        b.nl |= (c == '\n' || c == '\r') ? (1u << j) : 0u;
        b.gt |= c == '>' ? (1u << j) : 0u;
        prev_nl = c == '\n';
    }
    b.ls &= b.valid; b.nl &= b.valid; b.gt &= b.valid;
    return b;
}

// FASTA: keep masks of a lane if a header line (m2) / a sequence line (m1) enters it; kind of its last line start (0: none)
__device__ __forceinline__ void fasta_masks(const LaneBits& b, uint32_t* m1, uint32_t* m2, uint32_t* last_kind) {
    uint32_t keep = 0, on_seq = 0;
    bool seq = false;                                              // entering on a header line: nothing kept before the first line start
#pragma unroll
    for (int j = 0; j < FX_PER_THREAD; ++j) {
        const uint32_t bit = 1u << j;
        if (b.ls & bit) seq = !(b.gt & bit);
        on_seq |= seq ? bit : 0u;
    }
    const uint32_t hs = b.ls & b.gt;
    keep = hs | (on_seq & ~b.nl & b.valid);
    const uint32_t before_first = b.ls ? ((b.ls & (0u - b.ls)) - 1u) : 0xffffffffu;    // bits below the first line start
    *m2 = keep;
    *m1 = keep | (before_first & ~b.nl & b.valid);
    *last_kind = b.ls ? ((hs >> (31 - __clz(b.ls))) & 1u ? 2u : 1u) : 0u;
}

// FASTQ: M[t] = bytes whose line number within the lane is t mod 4 (counting the lane's own line starts up to and including the byte)
__device__ __forceinline__ void fastq_classes(const LaneBits& b, uint32_t (&M)[4]) {
    M[0] = M[1] = M[2] = M[3] = 0;
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < FX_PER_THREAD; ++j) {
        const uint32_t bit = 1u << j;
        c += (b.ls >> j) & 1u;
        const uint32_t t = c & 3u;
        M[0] |= t == 0 ? bit : 0u; M[1] |= t == 1 ? bit : 0u; M[2] |= t == 2 ? bit : 0u; M[3] |= t == 3 ? bit : 0u;
    }
}
// keep mask / header starts of a lane entered in phase q (the state of the byte in front of it, mod 4)
__device__ __forceinline__ uint32_t fastq_keep(const LaneBits& b, const uint32_t (&M)[4], uint32_t q, uint32_t* headers) {
    const uint32_t hl = M[(4u - q) & 3u], sl = M[(5u - q) & 3u];       // header line: q + t = 0, sequence line: q + t = 1 (mod 4)
    *headers = hl & b.ls;
    return (hl & b.ls) | (sl & ~b.nl & b.valid);
}

// exclusive scan over the workgroup's lanes: FASTA "last non-zero", FASTQ sum.  *total: the inclusive value of the last lane.
template <bool LASTNZ>
__device__ __forceinline__ uint32_t block_excl(uint32_t v, uint32_t* s_wave, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl = LASTNZ ? (incl ? incl : o) : incl + o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < FX_THREADS / 64; ++w) {
        const uint32_t x = s_wave[w];
        if (w < wave) before = LASTNZ ? (x ? x : before) : before + x;
        all = LASTNZ ? (x ? x : all) : all + x;
    }
    *total = all;
    uint32_t prev = __shfl_up(incl, 1);                          // inclusive value of the lane in front
    if (lane == 0) prev = 0u;
    return LASTNZ ? (prev ? prev : before) : before + prev;
}

__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t* s_red) {        // -> thread 0 holds the sum
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t t = 0;
    if (threadIdx.x == 0) for (int w = 0; w < FX_THREADS / 64; ++w) t += s_red[w];
    return t;
}

__global__ __launch_bounds__(FX_THREADS) void fx_summary_kernel(const uint8_t* __restrict__ raw, uint64_t n, int fastq,
                                                                 const uint8_t* __restrict__ carry, BlockSum* __restrict__ sums) {
    __shared__ uint32_t s_wave[FX_THREADS / 64], s_red[FX_THREADS / 64];
    const uint64_t base = (uint64_t)blockIdx.x * FX_BLOCK_BYTES + (uint64_t)threadIdx.x * FX_PER_THREAD;
    uint8_t bytes[FX_PER_THREAD];
    const LaneBits b = lane_bits(raw, base, n, carry[1], bytes);
    BlockSum out{};
    if (!fastq) {
        uint32_t m1, m2, kind, last;
        fasta_masks(b, &m1, &m2, &kind);
        const uint32_t inh = block_excl<true>(kind, s_wave, &last);
        const uint32_t local = inh == 1 ? __popc(m1) : __popc(m2);
        const uint32_t pre = inh == 0 ? __popc(m1 & ~m2) : 0u;
        const uint32_t c0 = block_sum(local, s_red), c1 = block_sum(pre, s_red), h = block_sum(__popc(b.ls & b.gt), s_red);
        out.cnt[0] = c0; out.cnt[1] = c1; out.hdr[0] = h; out.last_kind = last;
    } else {
        uint32_t M[4], all;
        fastq_classes(b, M);
        const uint32_t r = block_excl<false>(__popc(b.ls), s_wave, &all);
        uint32_t cq[4], hq[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) { uint32_t hm; cq[q] = __popc(fastq_keep(b, M, q, &hm)); hq[q] = __popc(hm); }
#pragma unroll
        for (uint32_t p = 0; p < 4; ++p) {
            const uint32_t q = (p + r) & 3u;
            const uint32_t c = q == 0 ? cq[0] : q == 1 ? cq[1] : q == 2 ? cq[2] : cq[3];
            const uint32_t h = q == 0 ? hq[0] : q == 1 ? hq[1] : q == 2 ? hq[2] : hq[3];
            out.cnt[p] = block_sum(c, s_red);
            out.hdr[p] = block_sum(h, s_red);
        }
        out.starts = all;
    }
    if (threadIdx.x == 0) sums[blockIdx.x] = out;
}

// one workgroup: entry state and output offset of every block, totals, the next piece's carry
__global__ __launch_bounds__(1024) void fx_offsets_kernel(const BlockSum* __restrict__ sums, unsigned n_blocks, int fastq,
                                                          const uint8_t* __restrict__ raw, uint64_t n, const uint8_t* __restrict__ carry,
                                                          uint8_t* __restrict__ carry_out, uint8_t* __restrict__ entry,
                                                          unsigned long long* __restrict__ block_off, unsigned long long* __restrict__ total,
                                                          unsigned long long* __restrict__ n_records) {
    __shared__ unsigned long long part[1024];
    __shared__ uint32_t st[1024];
    const unsigned per = (n_blocks + 1023) / 1024;
    const unsigned lo = threadIdx.x * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
    // what this thread's span does to the state
    uint32_t eff = 0;
    for (unsigned i = lo; i < hi; ++i) {
        if (fastq) eff += sums[i].starts;
        else { const uint32_t k = sums[i].last_kind; eff = k ? k : eff; }
    }
    st[threadIdx.x] = eff;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                         // Hillis-Steele inclusive scan of the spans' effects
        const uint32_t mine = st[threadIdx.x];
        const uint32_t v = threadIdx.x >= (unsigned)d ? st[threadIdx.x - d] : 0;
        __syncthreads();
        st[threadIdx.x] = fastq ? mine + v : (mine ? mine : v);
        __syncthreads();
    }
    const uint32_t before = threadIdx.x ? st[threadIdx.x - 1] : 0u;
    uint32_t e = fastq ? (uint32_t)((carry[0] + before) & 3u) : (before ? before : (uint32_t)carry[0]);
    unsigned long long kept = 0, recs = 0;
    for (unsigned i = lo; i < hi; ++i) {                          // (the summary is indexed in memory: a local copy indexed by the phase lands in scratch)
        const BlockSum* s = sums + i;
        entry[i] = (uint8_t)e;
        if (fastq) { kept += s->cnt[e & 3u]; recs += s->hdr[e & 3u]; e = (e + s->starts) & 3u; }
        else { const uint32_t lk = s->last_kind; kept += s->cnt[0] + (e == 1u ? s->cnt[1] : 0u); recs += s->hdr[0]; e = lk ? lk : e; }
    }
    if (hi == n_blocks && lo < hi) {                             // the thread that owns the last block: the carry
        carry_out[0] = (uint8_t)e;
        carry_out[1] = raw[n - 1] == '\n';
    }
    part[threadIdx.x] = kept;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const unsigned long long v = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned long long run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    for (unsigned i = lo; i < hi; ++i) {
        const BlockSum* s = sums + i;
        const uint32_t ei = entry[i];
        block_off[i] = run;
        run += fastq ? s->cnt[ei & 3u] : s->cnt[0] + (ei == 1u ? s->cnt[1] : 0u);
    }
    if (threadIdx.x == 1023) *total = part[1023];
    // records: a sum over the threads through the same scratch
    __syncthreads();
    part[threadIdx.x] = recs;
    __syncthreads();
    for (int d = 512; d > 0; d >>= 1) {
        if (threadIdx.x < (unsigned)d) part[threadIdx.x] += part[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0 && part[0]) atomicAdd(n_records, part[0]);
}

// every block: flags from its entry state, compacted in LDS, written out contiguously
__global__ __launch_bounds__(FX_THREADS) void fx_scatter_kernel(const uint8_t* __restrict__ raw, uint64_t n, int fastq,
                                                                 const uint8_t* __restrict__ carry, const uint8_t* __restrict__ entry,
                                                                 const unsigned long long* __restrict__ block_off, uint8_t* __restrict__ out) {
    __shared__ uint8_t s_out[FX_BLOCK_BYTES];
    __shared__ uint32_t s_wave[FX_THREADS / 64];
    const uint64_t base = (uint64_t)blockIdx.x * FX_BLOCK_BYTES + (uint64_t)threadIdx.x * FX_PER_THREAD;
    uint8_t bytes[FX_PER_THREAD];
    const LaneBits b = lane_bits(raw, base, n, carry[1], bytes);
    const uint32_t e = entry[blockIdx.x];
    uint32_t mask, dummy;
    if (!fastq) {
        uint32_t m1, m2, kind;
        fasta_masks(b, &m1, &m2, &kind);
        const uint32_t inh = block_excl<true>(kind, s_wave, &dummy);
        mask = (inh ? inh : e) == 1u ? m1 : m2;
    } else {
        uint32_t M[4], hm;
        fastq_classes(b, M);
        const uint32_t r = block_excl<false>(__popc(b.ls), s_wave, &dummy);
        mask = fastq_keep(b, M, (e + r) & 3u, &hm);
    }
    const uint32_t cnt = __popc(mask);
    __syncthreads();                                             // s_wave is reused
    uint32_t total;
    unsigned pos = block_excl<false>(cnt, s_wave, &total);
#pragma unroll
    for (int j = 0; j < FX_PER_THREAD; ++j)
        if (mask & (1u << j)) s_out[pos++] = bytes[j];
    __syncthreads();
    uint8_t* dst = out + block_off[blockIdx.x];
    for (unsigned i = threadIdx.x; i < total; i += FX_THREADS) dst[i] = s_out[i];
}

// dst[j] = last H bytes of (old halo ++ the n new bytes at src); src[-H .. -1] is the old halo
__global__ void halo_kernel(const uint8_t* __restrict__ src, const unsigned long long* __restrict__ n_new, int H,
                            uint8_t* __restrict__ dst) {
    const long long n = (long long)*n_new;
    for (int j = threadIdx.x; j < H; j += blockDim.x) dst[j] = src[n + j - H];
}

size_t align256(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace

// temp layout: [block summaries][entry states u8][block offsets u64]
size_t fastx_temp_bytes(uint64_t max_chunk) {
    const uint64_t n_blocks = (max_chunk + FX_BLOCK_BYTES - 1) / FX_BLOCK_BYTES + 1;
    return align256(n_blocks * sizeof(BlockSum)) + align256(n_blocks) + align256(n_blocks * 8) + 256;
}

// d_state: unused since round 6 (the per-byte states are never materialised); kept in the signature for the callers' scratch layout
hipError_t fastx_compact_launch(const uint8_t* d_raw, uint64_t n, int fastq, uint8_t* d_carry, uint8_t* d_state,
                                uint8_t* d_out, unsigned long long* d_n_out, unsigned long long* d_n_records,
                                void* d_temp, size_t temp_bytes, hipStream_t stream, bool last_piece) {
    (void)d_state;
    if (n == 0) return hipMemsetAsync(d_n_out, 0, 8, stream);
    if (temp_bytes < fastx_temp_bytes(n)) return hipErrorInvalidValue;
    const uint64_t n_blocks = (n + FX_BLOCK_BYTES - 1) / FX_BLOCK_BYTES;
    if (n_blocks > 0x7fffffffull) return hipErrorInvalidValue;
    BlockSum* sums = reinterpret_cast<BlockSum*>(d_temp);
    uint8_t* entry = reinterpret_cast<uint8_t*>((char*)d_temp + align256((n_blocks + 1) * sizeof(BlockSum)));
    unsigned long long* block_off = reinterpret_cast<unsigned long long*>((char*)entry + align256(n_blocks + 1));
    // the carry is read (first byte) and rewritten (last byte) by the same launches: go through a second slot
    hipLaunchKernelGGL(fx_summary_kernel, dim3((unsigned)n_blocks), dim3(FX_THREADS), 0, stream, d_raw, n, fastq, d_carry, sums);
    hipLaunchKernelGGL(fx_offsets_kernel, dim3(1), dim3(1024), 0, stream, (const BlockSum*)sums, (unsigned)n_blocks, fastq, d_raw, n, d_carry,
                       d_carry + 2, entry, block_off, d_n_out, d_n_records);
    hipLaunchKernelGGL(fx_scatter_kernel, dim3((unsigned)n_blocks), dim3(FX_THREADS), 0, stream, d_raw, n, fastq, d_carry, (const uint8_t*)entry,
                       (const unsigned long long*)block_off, d_out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess || last_piece) return e;
    return hipMemcpyAsync(d_carry, d_carry + 2, 2, hipMemcpyDeviceToDevice, stream);
}

hipError_t fastx_halo_launch(const uint8_t* d_src, const unsigned long long* d_n_new, int halo, uint8_t* d_dst,
                             hipStream_t stream) {
    if (halo <= 0) return hipSuccess;
    hipLaunchKernelGGL(halo_kernel, dim3(1), dim3(256), 0, stream, d_src, d_n_new, halo, d_dst);
    return hipGetLastError();
}

}  // namespace smg
