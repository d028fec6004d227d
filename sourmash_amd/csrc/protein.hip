// protein.hip -- protein / dayhoff / hp k-mers on the GPU.
//
// Reference: src/core/src/encodings.rs:103-368 (codon table with its third-position-N entries, dayhoff and hp
// alphabets, unknown residue -> 'X', to_aa dropping a trailing partial codon) and src/core/src/signature.rs:
// 307-393 (DNA into a protein sketch: frames 0..2, forward strand then reverse complement per frame, EVERY window
// of ksize/3 residues hashed -- this mode has no validity test; protein input: windows of the upper-cased residues).
//
//   residues_kernel   protein input: upper-case + alphabet mapping, one lane per byte
//   translate_kernel  DNA input: one lane per residue of the six translations, written as six segments separated
//                     by a 0xFF byte (no residue maps to it), so that one window kernel serves both inputs
//   window_fast_kernel<NB>  (round 5; k <= 79 residues) a lane owns the 8 window starts of one aligned word of the residue
//                     buffer and hashes its windows from registers (residue_core.hpp); kept hashes are staged in LDS and leave
//                     in batches (one global atomic per batch -- one per HIT, as the kernel below does it, is 5 million
//                     returning atomics on one word per 10^9 windows at scaled = 200: that alone was most of its 53 ms)
//   window_kernel     longer windows: one lane per window start, the k bytes copied out and hashed byte by byte
// both: skip windows touching a separator; keep 1 <= h <= thr (append) or write per position (dense).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include "device_api.hpp"
#include "murmur3.hpp"
#include "residues.hpp"
#include "residue_core.hpp"
#include "translate_core.hpp"

namespace smg {

namespace {

constexpr uint8_t SEP = 0xff;

// Both kernels map bytes through small tables in LDS that every block fills from the scalar functions of residues.hpp at its
// start (the functions ARE the definition: encodings.rs:103-347); round 4 evaluated those functions -- compare chains of a few
// dozen instructions -- for every byte, which made the dayhoff / hp / translated sketches 2.5 x slower than protein ones.
//   residues: byte -> residue of the sketch's alphabet, 16 bytes per lane
__global__ __launch_bounds__(256) void residues_kernel(const uint8_t* __restrict__ seq, uint64_t len, uint32_t hf,
                                                       uint8_t* __restrict__ aa) {
    __shared__ uint8_t lut[256];
    lut[threadIdx.x] = residue_encode(ascii_upper((uint8_t)threadIdx.x), hf);
    __syncthreads();
    const uint64_t n16 = len / 16;
    const bool aligned = (((uintptr_t)seq | (uintptr_t)aa) & 15) == 0;
    if (aligned)
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
            const uint4 v = reinterpret_cast<const uint4*>(seq)[i];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            uint32_t o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                o[k] = (uint32_t)lut[w[k] & 0xffu] | ((uint32_t)lut[(w[k] >> 8) & 0xffu] << 8) | ((uint32_t)lut[(w[k] >> 16) & 0xffu] << 16) |
                       ((uint32_t)lut[w[k] >> 24] << 24);
            reinterpret_cast<uint4*>(aa)[i] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    for (uint64_t i = (aligned ? n16 * 16 : 0) + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (uint64_t)gridDim.x * blockDim.x)
        aa[i] = lut[seq[i]];
}

// segment s = 2 * frame + strand holds (len - frame) / 3 residues and is followed by one separator.
//   translate: byte -> nucleotide code (A C G T N other = 0 .. 5; on the reverse strand the code of the complement), three codes ->
//   the residue of the sketch's alphabet through a 6 x 6 x 6 table
__global__ __launch_bounds__(256) void translate_kernel(const uint8_t* __restrict__ seq, uint64_t len, uint32_t hf,
                                                        uint8_t* __restrict__ aa, uint64_t total) {
    __shared__ uint8_t code_f[256], code_r[256], codon[216];
    {
        const uint8_t c = ascii_upper((uint8_t)threadIdx.x);
        code_f[threadIdx.x] = (uint8_t)nt_code(c);
        code_r[threadIdx.x] = (uint8_t)nt_code(dna_complement_or_nul(c));
        if (threadIdx.x < 216) {
            const char* letters = "ACGTN?";                            // '?' stands for every byte that is no base
            const int x = threadIdx.x / 36, y = (threadIdx.x / 6) % 6, z = threadIdx.x % 6;
            codon[threadIdx.x] = residue_encode(translate_codon((uint8_t)letters[x], (uint8_t)letters[y], (uint8_t)letters[z]), hf);
        }
    }
    __syncthreads();
    uint64_t start[7];
    start[0] = 0;
    for (int s = 0; s < 6; ++s) start[s + 1] = start[s] + (len - (uint64_t)(s >> 1)) / 3 + 1;
    for (uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (uint64_t)gridDim.x * blockDim.x) {
        int s = 0;
        while (o >= start[s + 1]) ++s;
        const uint64_t i = o - start[s];
        const int frame = s >> 1;
        if (i == start[s + 1] - start[s] - 1) { aa[o] = SEP; continue; }
        const uint64_t p = (uint64_t)frame + 3 * i;
        uint32_t x, y, z;
        if (s & 1) {                                  // reverse complement: rc[p] = complement(seq[len - 1 - p])
            x = code_r[seq[len - 1 - p]]; y = code_r[seq[len - 2 - p]]; z = code_r[seq[len - 3 - p]];
        } else {
            x = code_f[seq[p]]; y = code_f[seq[p + 1]]; z = code_f[seq[p + 2]];
        }
        aa[o] = codon[x * 36 + y * 6 + z];
    }
}

// the same translation, one aligned output WORD (four residues of one segment) per lane: translate_core.hpp.  Needs 4-byte aligned
// input and output; the kernel above serves anything else.
__global__ __launch_bounds__(256) void translate_words_kernel(const uint8_t* __restrict__ seq, uint64_t len, uint32_t hf,
                                                              uint8_t* __restrict__ aa, uint64_t total) {
    __shared__ uint8_t code_f[256], code_r[256], codon[216];
    {
        const uint8_t c = ascii_upper((uint8_t)threadIdx.x);
        code_f[threadIdx.x] = (uint8_t)nt_code(c);
        code_r[threadIdx.x] = (uint8_t)nt_code(dna_complement_or_nul(c));
        if (threadIdx.x < 216) {
            const char* letters = "ACGTN?";                            // '?' stands for every byte that is no base
            const int x = threadIdx.x / 36, y = (threadIdx.x / 6) % 6, z = threadIdx.x % 6;
            codon[threadIdx.x] = residue_encode(translate_codon((uint8_t)letters[x], (uint8_t)letters[y], (uint8_t)letters[z]), hf);
        }
    }
    __syncthreads();
    const TranslateLayout L = translate_layout(len);
    const TranslateTables T{code_f, code_r, codon};
    const uint64_t n_words = (total + 3) / 4;
    for (uint64_t G = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; G < n_words; G += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t w = translate_word(seq, reinterpret_cast<const uint32_t*>(seq), L, T, G);
        if (4 * G + 4 <= total) reinterpret_cast<uint32_t*>(aa)[G] = w;
        else
            for (int j = 0; j < 4; ++j)
                if (4 * G + (uint64_t)j < total) aa[4 * G + (uint64_t)j] = (uint8_t)(w >> (8 * j));
    }
}

constexpr int MAX_RESIDUES = 256;

__global__ __launch_bounds__(256) void window_kernel(const uint8_t* __restrict__ aa, uint64_t n, uint32_t k, uint64_t seed,
                                                     uint64_t thr, uint64_t* __restrict__ out,
                                                     unsigned long long* out_count, uint64_t cap, int dense) {
    const uint64_t n_win = n - k + 1;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_win; i += (uint64_t)gridDim.x * blockDim.x) {
        uint8_t buf[MAX_RESIDUES];
        bool ok = true;
        for (uint32_t j = 0; j < k; ++j) {
            const uint8_t c = aa[i + j];
            ok = ok && c != SEP;
            buf[j] = c;
        }
        if (!ok) continue;
        const uint64_t h = mmh3_h1_bytes(buf, k, seed);
        if (dense) {
            if (i < cap) out[i] = h;
        } else if ((h - 1) < thr) {
            const unsigned long long g = atomicAdd(out_count, 1ull);
            if (g < cap) out[g] = h;
        }
    }
}

constexpr int RW_BLOCK = 256;
constexpr int RW_OUT_CAP = 2048;     // LDS staging entries for kept hashes (16 KiB)

template <int NB, bool DENSE>
__global__ __launch_bounds__(RW_BLOCK) void window_fast_kernel(const uint64_t* __restrict__ aa64, uint64_t n, ResidueTail t, uint64_t seed,
                                                               uint64_t thr, uint64_t* __restrict__ out,
                                                               unsigned long long* __restrict__ out_count, uint64_t cap, uint64_t n_lanes) {
    __shared__ uint64_t s_out[RW_OUT_CAP];
    __shared__ unsigned int s_cnt;
    __shared__ unsigned long long s_base;
    const int tid = threadIdx.x;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    auto flush = [&](unsigned int at_least) {                            // (called by every thread of the block)
        __syncthreads();
        const unsigned int cnt = s_cnt;
        if (cnt >= at_least && cnt) {
            const unsigned int m = cnt < (unsigned)RW_OUT_CAP ? cnt : (unsigned)RW_OUT_CAP;
            if (tid == 0) s_base = atomicAdd(out_count, (unsigned long long)m);
            __syncthreads();
            const unsigned long long b = s_base;
            for (unsigned int i = tid; i < m; i += RW_BLOCK)
                if (b + i < cap) out[b + i] = s_out[i];
            __syncthreads();
            if (tid == 0) s_cnt = 0;
        }
        __syncthreads();
    };
    const uint64_t rounds = (n_lanes + (uint64_t)gridDim.x * RW_BLOCK - 1) / ((uint64_t)gridDim.x * RW_BLOCK);
    for (uint64_t r = 0; r < rounds; ++r) {                              // (uniform trip count: the flush has barriers)
        const uint64_t g = (r * gridDim.x + blockIdx.x) * RW_BLOCK + tid;
        if (g < n_lanes)
            residue_windows_lane<NB>(aa64, n, g, t, seed, [&](uint64_t start, uint64_t h) {
                if constexpr (DENSE) {
                    if (start < cap) out[start] = h;
                } else if ((h - 1) < thr) {
                    const unsigned int idx = atomicAdd(&s_cnt, 1u);
                    if (idx < (unsigned)RW_OUT_CAP) {
                        s_out[idx] = h;
                    } else {                                             // pathological density: straight to HBM
                        const unsigned long long gi = atomicAdd(out_count, 1ull);
                        if (gi < cap) out[gi] = h;
                    }
                }
            });
        if constexpr (!DENSE) flush(RW_OUT_CAP / 2);
    }
    if constexpr (!DENSE) flush(1);
}

template <int NB>
hipError_t window_fast_launch(const uint8_t* d_aa, uint64_t n, uint32_t k, uint64_t seed, uint64_t thr, uint64_t* d_out,
                              unsigned long long* d_count, uint64_t cap, bool dense, hipStream_t stream) {
    const uint64_t n_lanes = (n - k + 1 + RW_P - 1) / RW_P;              // lanes that own at least one window start
    const uint64_t blocks = (n_lanes + RW_BLOCK - 1) / RW_BLOCK;
    const unsigned grid = (unsigned)(blocks < 1 ? 1 : (blocks > 256 * 8 ? 256 * 8 : blocks));
    const ResidueTail t = residue_tail(k);
    if (dense)
        hipLaunchKernelGGL((window_fast_kernel<NB, true>), dim3(grid), dim3(RW_BLOCK), 0, stream, (const uint64_t*)d_aa, n, t, seed, thr, d_out,
                           d_count, cap, n_lanes);
    else
        hipLaunchKernelGGL((window_fast_kernel<NB, false>), dim3(grid), dim3(RW_BLOCK), 0, stream, (const uint64_t*)d_aa, n, t, seed, thr, d_out,
                           d_count, cap, n_lanes);
    return hipGetLastError();
}

unsigned grid_for(uint64_t n) {
    const uint64_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

uint64_t translated_bytes(uint64_t len) {
    uint64_t t = 0;
    for (int s = 0; s < 6; ++s) t += (len - (uint64_t)(s >> 1)) / 3 + 1;
    return t;
}

hipError_t residues_launch(const uint8_t* d_seq, uint64_t len, uint32_t hash_function, uint8_t* d_aa, hipStream_t stream) {
    if (len == 0) return hipSuccess;
    hipLaunchKernelGGL(residues_kernel, dim3(grid_for(len)), dim3(256), 0, stream, d_seq, len, hash_function, d_aa);
    return hipGetLastError();
}

hipError_t translate_launch(const uint8_t* d_seq, uint64_t len, uint32_t hash_function, uint8_t* d_aa, hipStream_t stream) {
    if (len < 3) return hipErrorInvalidValue;
    const uint64_t total = translated_bytes(len);
    if ((((uintptr_t)d_seq | (uintptr_t)d_aa) & 3) == 0)
        hipLaunchKernelGGL(translate_words_kernel, dim3(grid_for((total + 3) / 4)), dim3(256), 0, stream, d_seq, len, hash_function, d_aa, total);
    else
        hipLaunchKernelGGL(translate_kernel, dim3(grid_for(total)), dim3(256), 0, stream, d_seq, len, hash_function, d_aa, total);
    return hipGetLastError();
}

hipError_t residue_windows_launch(const uint8_t* d_aa, uint64_t n, uint32_t k, uint64_t seed, uint64_t thr, uint64_t* d_out,
                                  unsigned long long* d_count, uint64_t cap, bool dense, hipStream_t stream) {
    if (k == 0 || k > (uint32_t)MAX_RESIDUES) return hipErrorInvalidValue;
    if (n < k) return hipSuccess;
    // the register-window form: windows of up to 79 residues over an 8-byte aligned buffer (every buffer the library makes is);
    // longer windows and unaligned caller buffers take the byte-wise kernel below
    if (k / 16 <= (uint32_t)RW_MAX_NB && ((uintptr_t)d_aa & 7) == 0) {
        switch (k / 16) {
        case 0: return window_fast_launch<0>(d_aa, n, k, seed, thr, d_out, d_count, cap, dense, stream);
        case 1: return window_fast_launch<1>(d_aa, n, k, seed, thr, d_out, d_count, cap, dense, stream);
        case 2: return window_fast_launch<2>(d_aa, n, k, seed, thr, d_out, d_count, cap, dense, stream);
        case 3: return window_fast_launch<3>(d_aa, n, k, seed, thr, d_out, d_count, cap, dense, stream);
        default: return window_fast_launch<4>(d_aa, n, k, seed, thr, d_out, d_count, cap, dense, stream);
        }
    }
    hipLaunchKernelGGL(window_kernel, dim3(grid_for(n - k + 1)), dim3(256), 0, stream, d_aa, n, k, seed, thr, d_out, d_count,
                       cap, dense ? 1 : 0);
    return hipGetLastError();
}

}  // namespace smg
