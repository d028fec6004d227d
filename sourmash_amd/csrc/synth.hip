// synth.hip -- synthetic random DNA generated directly in HBM.
//
// BASELINE.json config C2 ("sketch 10 GB synthetic random-DNA") with the
// generator of SURVEY.md section 8(d): base p = "ACGT"[(splitmix64(seed ^ (p >> 5))
// >> (2 * (p & 31))) & 3]; every position p with (p + 1) % (record_len + 1) == 0
// holds the record separator '\n' (records are record_len bases; 0 = none).
// oracle/oracle.c:orc_synth_dna is the CPU twin used by the parity tests.
#include <hip/hip_runtime.h>
#include "device_api.hpp"

namespace smg {

__device__ __forceinline__ uint64_t splitmix64_dev(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

// one thread = 32 bases = one splitmix64 word = two 16-byte stores
__global__ __launch_bounds__(256) void synth_dna_kernel(uint8_t* __restrict__ out, uint64_t start, uint64_t n,
                                                        uint64_t seed, uint64_t record_len) {
    const uint64_t first_word = start >> 5;
    const uint64_t n_words = ((start + n + 31) >> 5) - first_word;
    for (uint64_t wi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; wi < n_words;
         wi += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t word = first_word + wi;
        const uint64_t bits = splitmix64_dev(seed ^ word);
        const uint64_t p0 = word << 5;
        uint32_t dw[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int i = 4 * d + b;
                const uint32_t code = (uint32_t)(bits >> (2 * i)) & 3u;
                uint32_t ch = (0x54474341u >> (8 * code)) & 0xffu;   // "ACGT"
                if (record_len && (p0 + i + 1) % (record_len + 1) == 0) ch = '\n';
                v |= ch << (8 * b);
            }
            dw[d] = v;
        }
        if (p0 >= start && p0 + 32 <= start + n && (((uintptr_t)(out + (p0 - start))) & 15) == 0) {
            uint4* o = reinterpret_cast<uint4*>(out + (p0 - start));
            o[0] = make_uint4(dw[0], dw[1], dw[2], dw[3]);
            o[1] = make_uint4(dw[4], dw[5], dw[6], dw[7]);
        } else {
            for (int i = 0; i < 32; ++i) {
                const uint64_t p = p0 + i;
                if (p >= start && p < start + n) out[p - start] = (uint8_t)(dw[i >> 2] >> (8 * (i & 3)));
            }
        }
    }
}

hipError_t synth_dna_launch(uint8_t* d_out, uint64_t start, uint64_t n, uint64_t seed, uint64_t record_len,
                            hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const uint64_t n_words = ((start + n + 31) >> 5) - (start >> 5);
    const uint64_t nb = (n_words + 255) / 256;
    const unsigned grid = (unsigned)(nb < 8192 ? nb : 8192);
    hipLaunchKernelGGL(synth_dna_kernel, dim3(grid), dim3(256), 0, stream, d_out, start, n, seed, record_len);
    return hipGetLastError();
}

}  // namespace smg
