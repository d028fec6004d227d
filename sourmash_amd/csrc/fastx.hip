// fastx.hip -- FASTA / FASTQ record structure resolved on the GPU.
//
// The reference parses sequence files on the host, one record at a time (screed / needletail:
// src/sourmash/command_sketch.py:697,746-768, src/core/benches/compute.rs:35-38).  Here the host only moves raw
// file bytes into HBM; which bytes are sequence is decided on the device at memory bandwidth:
//   scan       per byte, the kind of line it is on.  FASTA: "kind of the most recent line start" (last-non-zero
//              scan over 0 / 1 = sequence line starts here / 2 = header line starts here); FASTQ: line number mod 4
//              (wrapping u8 sum of line starts; line 1 of every 4 is sequence).  The scan input is generated from
//              the raw bytes on the fly (rocPRIM scan over a transform iterator).
//   keep       a byte stays if it is on a sequence line and is not CR/LF, or if it is the first byte of a header
//              line ('>' / '@'): that one remains in the stream as the record separator -- it is outside ACGT, so
//              it kills exactly the k-mers that would span two records, which is what one add_sequence call per
//              record achieves
//   compact    order preserving, two passes over 8 KiB blocks: kept bytes per block -> exclusive scan of the
//              block counts -> every block compacts itself in LDS and writes its bytes out contiguously
// Chunks are chained through a 4-byte carry (state of the last byte, "ended on a newline", line number), so a
// file streams through in pieces of any size.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include "fastx_api.hpp"

namespace smg {

namespace {

struct LastNonZero {
    __host__ __device__ uint8_t operator()(uint8_t a, uint8_t b) const { return b ? b : a; }
};
struct WrapSum {
    __host__ __device__ uint8_t operator()(uint8_t a, uint8_t b) const { return (uint8_t)(a + b); }
};

// carry[0] state of the last byte of the previous piece (FASTA: 1 | 2; FASTQ: line number & 3),
// carry[1] that byte was '\n'.  The scan input is generated on the fly from the raw bytes.
struct Classify {
    const uint8_t* raw;
    const uint8_t* carry;
    int fastq;
    __device__ uint8_t operator()(size_t i) const {
        const bool line_start = i ? raw[i - 1] == '\n' : carry[1] != 0;
        if (fastq) return (uint8_t)((line_start ? 1 : 0) + (i ? 0 : carry[0]));
        return line_start ? (raw[i] == '>' ? 2 : 1) : (i ? 0 : carry[0]);
    }
};

constexpr int FX_THREADS = 256;
constexpr int FX_PER_THREAD = 32;                              // consecutive bytes per lane (two 16-byte loads)
constexpr int FX_BLOCK_BYTES = FX_THREADS * FX_PER_THREAD;     // 8 KiB per workgroup

// keep mask (bit j = byte j of this lane's run is kept) and number of header lines starting in the run
__device__ __forceinline__ uint32_t lane_flags(const uint8_t* __restrict__ raw, const uint8_t* __restrict__ state,
                                               uint64_t base, uint64_t n, int fastq, uint8_t prev_nl_at_0,
                                               uint8_t* bytes, unsigned* headers) {
    uint32_t mask = 0;
    unsigned hdr = 0;
    if (base >= n) { *headers = 0; return 0; }
    uint8_t st[FX_PER_THREAD];
    if (base + FX_PER_THREAD <= n) {
        const uint4 r0 = *reinterpret_cast<const uint4*>(raw + base), r1 = *reinterpret_cast<const uint4*>(raw + base + 16);
        const uint4 s0 = *reinterpret_cast<const uint4*>(state + base), s1 = *reinterpret_cast<const uint4*>(state + base + 16);
        memcpy(bytes, &r0, 16); memcpy(bytes + 16, &r1, 16);
        memcpy(st, &s0, 16); memcpy(st + 16, &s1, 16);
    } else {
        for (int j = 0; j < FX_PER_THREAD; ++j) {
            bytes[j] = base + j < n ? raw[base + j] : (uint8_t)'\n';
            st[j] = base + j < n ? state[base + j] : 0;
        }
    }
    bool prev_nl = base ? raw[base - 1] == '\n' : prev_nl_at_0 != 0;
    const int lim = base + FX_PER_THREAD <= n ? FX_PER_THREAD : (int)(n - base);
#pragma unroll
    for (int j = 0; j < FX_PER_THREAD; ++j) {
        const uint8_t c = bytes[j], s = st[j];
        const bool header_line = fastq ? (s & 3) == 0 : s == 2;
        const bool seq_line = fastq ? (s & 3) == 1 : s == 1;
        const bool header_start = header_line && prev_nl;
        const bool keep = j < lim && (header_start || (seq_line && c != '\n' && c != '\r'));
        hdr += (j < lim && header_start) ? 1 : 0;
        mask |= keep ? (1u << j) : 0u;
        prev_nl = c == '\n';
    }
    *headers = hdr;
    return mask;
}

// pass A: kept bytes per 8 KiB block, header lines, and the carry the next piece starts from
__global__ __launch_bounds__(FX_THREADS) void count_kernel(const uint8_t* __restrict__ raw, const uint8_t* __restrict__ state,
                                                            uint64_t n, int fastq, const uint8_t* __restrict__ carry,
                                                            uint8_t* __restrict__ carry_out, unsigned int* __restrict__ block_count,
                                                            unsigned long long* n_records) {
    __shared__ unsigned int red[FX_THREADS / 64][2];
    const uint64_t base = (uint64_t)blockIdx.x * FX_BLOCK_BYTES + (uint64_t)threadIdx.x * FX_PER_THREAD;
    uint8_t bytes[FX_PER_THREAD];
    unsigned hdr = 0;
    const uint32_t mask = lane_flags(raw, state, base, n, fastq, carry[1], bytes, &hdr);
    unsigned cnt = __popc(mask);
    if (base < n && n - base <= FX_PER_THREAD) {                 // this lane owns the last byte of the piece
        const uint8_t s = state[n - 1];
        carry_out[0] = fastq ? (uint8_t)(s & 3) : s;
        carry_out[1] = raw[n - 1] == '\n';
    }
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_down(cnt, off);
        hdr += __shfl_down(hdr, off);
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = cnt; red[threadIdx.x >> 6][1] = hdr; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned c = 0, h = 0;
        for (int w = 0; w < FX_THREADS / 64; ++w) { c += red[w][0]; h += red[w][1]; }
        block_count[blockIdx.x] = c;
        if (h) atomicAdd(n_records, (unsigned long long)h);
    }
}

// exclusive scan of the block counts (one workgroup; a piece has at most a few thousand blocks)
__global__ __launch_bounds__(1024) void offsets_kernel(const unsigned int* __restrict__ block_count, unsigned n_blocks,
                                                       unsigned long long* __restrict__ block_off,
                                                       unsigned long long* __restrict__ total) {
    __shared__ unsigned long long part[1024];
    const unsigned per = (n_blocks + 1023) / 1024;
    const unsigned lo = threadIdx.x * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
    unsigned long long s = 0;
    for (unsigned i = lo; i < hi; ++i) s += block_count[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                         // Hillis-Steele inclusive scan of the partials
        const unsigned long long v = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned long long run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    for (unsigned i = lo; i < hi; ++i) { block_off[i] = run; run += block_count[i]; }
    if (threadIdx.x == 1023) *total = part[1023];
}

// pass B: recompute the flags, compact the block in LDS, write it out contiguously
__global__ __launch_bounds__(FX_THREADS) void scatter_kernel(const uint8_t* __restrict__ raw, const uint8_t* __restrict__ state,
                                                              uint64_t n, int fastq, const uint8_t* __restrict__ carry,
                                                              const unsigned long long* __restrict__ block_off,
                                                              uint8_t* __restrict__ out) {
    __shared__ uint8_t s_out[FX_BLOCK_BYTES];
    __shared__ unsigned int s_wave[FX_THREADS / 64];
    const uint64_t base = (uint64_t)blockIdx.x * FX_BLOCK_BYTES + (uint64_t)threadIdx.x * FX_PER_THREAD;
    uint8_t bytes[FX_PER_THREAD];
    unsigned hdr = 0;
    const uint32_t mask = lane_flags(raw, state, base, n, fastq, carry[1], bytes, &hdr);
    const unsigned cnt = __popc(mask);
    // exclusive prefix of cnt across the workgroup: wave scan + wave totals
    unsigned incl = cnt;
    const int lane = threadIdx.x & 63;
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    if (lane == 63) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned wave_base = 0, total = 0;
    for (int w = 0; w < FX_THREADS / 64; ++w) {
        if (w < (int)(threadIdx.x >> 6)) wave_base += s_wave[w];
        total += s_wave[w];
    }
    unsigned pos = wave_base + incl - cnt;
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
size_t scan_temp_bytes(uint64_t n) {
    size_t a = 0, b = 0;
    auto in = rocprim::make_transform_iterator(rocprim::counting_iterator<size_t>(0), Classify{nullptr, nullptr, 0});
    (void)rocprim::inclusive_scan(nullptr, a, in, (uint8_t*)nullptr, (size_t)n, LastNonZero(), (hipStream_t)0);
    (void)rocprim::inclusive_scan(nullptr, b, in, (uint8_t*)nullptr, (size_t)n, WrapSum(), (hipStream_t)0);
    return align256(a > b ? a : b);
}

}  // namespace

// temp layout: [scan scratch][block counts u32][block offsets u64]
size_t fastx_temp_bytes(uint64_t max_chunk) {
    const uint64_t n_blocks = (max_chunk + FX_BLOCK_BYTES - 1) / FX_BLOCK_BYTES + 1;
    return scan_temp_bytes(max_chunk) + align256(n_blocks * 4) + align256(n_blocks * 8) + 256;
}

hipError_t fastx_compact_launch(const uint8_t* d_raw, uint64_t n, int fastq, uint8_t* d_carry, uint8_t* d_state,
                                uint8_t* d_out, unsigned long long* d_n_out, unsigned long long* d_n_records,
                                void* d_temp, size_t temp_bytes, hipStream_t stream) {
    if (n == 0) return hipMemsetAsync(d_n_out, 0, 8, stream);
    if (temp_bytes < fastx_temp_bytes(n)) return hipErrorInvalidValue;
    const uint64_t n_blocks = (n + FX_BLOCK_BYTES - 1) / FX_BLOCK_BYTES;
    size_t scan_bytes = scan_temp_bytes(n);
    unsigned int* block_count = reinterpret_cast<unsigned int*>((char*)d_temp + scan_bytes);
    unsigned long long* block_off = reinterpret_cast<unsigned long long*>((char*)block_count + align256((n_blocks + 1) * 4));
    auto in = rocprim::make_transform_iterator(rocprim::counting_iterator<size_t>(0), Classify{d_raw, d_carry, fastq});
    hipError_t e;
    if (fastq) e = rocprim::inclusive_scan(d_temp, scan_bytes, in, d_state, (size_t)n, WrapSum(), stream);
    else e = rocprim::inclusive_scan(d_temp, scan_bytes, in, d_state, (size_t)n, LastNonZero(), stream);
    if (e != hipSuccess) return e;
    // the carry is read (first byte) and rewritten (last byte) by the same launch: go through a second slot
    hipLaunchKernelGGL(count_kernel, dim3((unsigned)n_blocks), dim3(FX_THREADS), 0, stream, d_raw, d_state, n, fastq, d_carry,
                       d_carry + 2, block_count, d_n_records);
    hipLaunchKernelGGL(offsets_kernel, dim3(1), dim3(1024), 0, stream, block_count, (unsigned)n_blocks, block_off, d_n_out);
    hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)n_blocks), dim3(FX_THREADS), 0, stream, d_raw, d_state, n, fastq, d_carry,
                       block_off, d_out);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return hipMemcpyAsync(d_carry, d_carry + 2, 2, hipMemcpyDeviceToDevice, stream);
}

hipError_t fastx_halo_launch(const uint8_t* d_src, const unsigned long long* d_n_new, int halo, uint8_t* d_dst,
                             hipStream_t stream) {
    if (halo <= 0) return hipSuccess;
    hipLaunchKernelGGL(halo_kernel, dim3(1), dim3(256), 0, stream, d_src, d_n_new, halo, d_dst);
    return hipGetLastError();
}

}  // namespace smg
