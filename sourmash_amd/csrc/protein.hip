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
//   window_kernel     one lane per window start: skip windows touching a separator, MurmurHash3 of the k bytes,
//                     keep 1 <= h <= thr (append) or write per position (dense)
// Residue k-mers are short keys of arbitrary length (7 ... 60 bytes); the byte-wise hash of murmur3.hpp is used.
#include <hip/hip_runtime.h>
#include "device_api.hpp"
#include "murmur3.hpp"
#include "residues.hpp"

namespace smg {

namespace {

constexpr uint8_t SEP = 0xff;

__global__ __launch_bounds__(256) void residues_kernel(const uint8_t* __restrict__ seq, uint64_t len, uint32_t hf,
                                                       uint8_t* __restrict__ aa) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (uint64_t)gridDim.x * blockDim.x)
        aa[i] = residue_encode(ascii_upper(seq[i]), hf);
}

// segment s = 2 * frame + strand holds (len - frame) / 3 residues and is followed by one separator
__global__ __launch_bounds__(256) void translate_kernel(const uint8_t* __restrict__ seq, uint64_t len, uint32_t hf,
                                                        uint8_t* __restrict__ aa, uint64_t total) {
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
        uint8_t a, b, c;
        if (s & 1) {                                  // reverse complement: rc[p] = complement(seq[len - 1 - p])
            a = dna_complement_or_nul(ascii_upper(seq[len - 1 - p]));
            b = dna_complement_or_nul(ascii_upper(seq[len - 2 - p]));
            c = dna_complement_or_nul(ascii_upper(seq[len - 3 - p]));
        } else {
            a = ascii_upper(seq[p]); b = ascii_upper(seq[p + 1]); c = ascii_upper(seq[p + 2]);
        }
        aa[o] = residue_encode(translate_codon(a, b, c), hf);
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
    hipLaunchKernelGGL(translate_kernel, dim3(grid_for(total)), dim3(256), 0, stream, d_seq, len, hash_function, d_aa, total);
    return hipGetLastError();
}

hipError_t residue_windows_launch(const uint8_t* d_aa, uint64_t n, uint32_t k, uint64_t seed, uint64_t thr, uint64_t* d_out,
                                  unsigned long long* d_count, uint64_t cap, bool dense, hipStream_t stream) {
    if (k == 0 || k > (uint32_t)MAX_RESIDUES) return hipErrorInvalidValue;
    if (n < k) return hipSuccess;
    hipLaunchKernelGGL(window_kernel, dim3(grid_for(n - k + 1)), dim3(256), 0, stream, d_aa, n, k, seed, thr, d_out, d_count,
                       cap, dense ? 1 : 0);
    return hipGetLastError();
}

}  // namespace smg
