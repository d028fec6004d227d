// sketch_multi.hip -- several ksizes of one DNA stretch in ONE pass (round 6).
//
// What the reference does here: Signature::add_sequence hands every k-mer window to each of the signature's sketches in turn
// (src/core/src/signature.rs:661-677: one SeqToHashes walk per ksize).  sketch.hip runs one launch per ksize, which costs nothing
// on long inputs -- everything after the staging of the bytes depends on k (both strands' words, the canonical compare,
// MurmurHash3: 112 of the 116 instructions a k-mer costs at k = 31) -- but a single genome is 4.6 MB: three launches of ~30-50 us
// each where the work is ~55 us.  Here a tile's bytes are staged once, every lane pulls the window of the LARGEST ksize into
// registers, and process_lane<K> (kmer_core.hpp: the same code as the per-ksize kernels) runs for each ksize on the window's
// prefix, each with its own LDS list of kept hashes and its own output.  Instantiated for the ksize sets listed at the end
// (21 / 31 / 51: the standard set of `sourmash sketch dna -p k=21,k=31,k=51`); any other set takes the per-ksize launches.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sketch_kernel.hpp"
#include "device_api.hpp"

namespace smg {

namespace {

constexpr int SM_N = 3;              // ksizes per pass
constexpr int SM_OUT_CAP = 1024;     // LDS staging entries for kept hashes, per ksize (8 KiB each)

struct MultiArgs {
    uint64_t thr[SM_N];
    uint64_t* out[SM_N];
    unsigned long long* count[SM_N];
    uint64_t cap[SM_N];
    uint32_t start[SM_N];            // k-mers of ksize j start at positions >= start[j] of the (aligned) buffer: in front of that lies
                                     // the halo of a LONGER ksize, whose k-mers of this size the previous piece has hashed already
};

template <int KA, int KB, int KC>
__global__ __launch_bounds__(SK_BLOCK) __attribute__((amdgpu_waves_per_eu(3, 8))) void sketch_dna_multi_kernel(const uint8_t* __restrict__ seq, uint64_t len, uint64_t seed, MultiArgs a,
                                                                    uint64_t n_tiles, uint32_t skip) {
    static_assert(KA < KB && KB < KC, "ascending ksizes: the window of the last one holds the others'");
    constexpr int P = 16;
    using G = LaneGeom<KC, P>;
    constexpr int TILE = SK_BLOCK * P;
    constexpr int LANE_RD = ((G::NW + 3) / 4) * 4;
    constexpr int IN_DW = (SK_BLOCK - 1) * (P / 4) + LANE_RD;
    constexpr int IN_CHUNKS = (IN_DW + 3) / 4;
    __shared__ __attribute__((aligned(16))) uint32_t s_in[IN_CHUNKS * 4];
    __shared__ uint64_t s_out[SM_N][SM_OUT_CAP];
    __shared__ unsigned int s_cnt[SM_N];
    __shared__ unsigned long long s_base[SM_N];
    const int tid = threadIdx.x;
    if (tid < SM_N) s_cnt[tid] = 0;

    auto flush = [&](bool always) {                    // (called by every thread, after a barrier)
#pragma unroll
        for (int j = 0; j < SM_N; ++j) {
            const unsigned int cnt = s_cnt[j];
            const bool go = always ? cnt != 0 : cnt >= (unsigned)SM_OUT_CAP / 2;       // workgroup-uniform
            if (!go) continue;
            const unsigned int n = cnt < (unsigned)SM_OUT_CAP ? cnt : (unsigned)SM_OUT_CAP;
            if (tid == 0) s_base[j] = atomicAdd(a.count[j], (unsigned long long)n);
            __syncthreads();
            const unsigned long long b = s_base[j];
            for (unsigned int i = tid; i < n; i += SK_BLOCK)
                if (b + i < a.cap[j]) a.out[j][b + i] = s_out[j][i];
            __syncthreads();
            if (tid == 0) s_cnt[j] = 0;
        }
    };

    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t base = tile * (uint64_t)TILE;
        __syncthreads();
        for (int c = tid; c < IN_CHUNKS; c += SK_BLOCK) {
            const uint64_t off = base + (uint64_t)c * 16;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (off + 16 <= len) {
                v = *reinterpret_cast<const uint4*>(seq + off);
            } else if (off < len) {
                uint32_t w[4] = {0, 0, 0, 0};
                for (uint64_t b = off; b < len; ++b) w[(b - off) >> 2] |= (uint32_t)seq[b] << (8 * ((b - off) & 3));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            if (off == 0 && skip) {                      // blank the alignment prefix
                uint32_t w[4] = {v.x, v.y, v.z, v.w};
                for (uint32_t b = 0; b < skip; ++b) w[b >> 2] &= ~(0xffu << (8 * (b & 3)));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            *reinterpret_cast<uint4*>(&s_in[c * 4]) = v;
        }
        __syncthreads();
        uint32_t raw[LANE_RD];
        const uint4* wp = reinterpret_cast<const uint4*>(&s_in[tid * (P / 4)]);
#pragma unroll
        for (int i = 0; i < LANE_RD / 4; ++i) {
            const uint4 v = wp[i];
            raw[4 * i] = v.x; raw[4 * i + 1] = v.y; raw[4 * i + 2] = v.z; raw[4 * i + 3] = v.w;
        }
        const uint64_t lane0 = base + (uint64_t)tid * P;
        auto keep = [&](int j, int o, uint64_t h) {
            if (lane0 + (uint64_t)o < (uint64_t)a.start[j]) return;        // hashed with the previous piece (see MultiArgs)
            const unsigned int idx = atomicAdd(&s_cnt[j], 1u);
            if (idx < (unsigned)SM_OUT_CAP) {
                s_out[j][idx] = h;
            } else {                                     // pathological density: straight to HBM
                const unsigned long long g = atomicAdd(a.count[j], 1ull);
                if (g < a.cap[j]) a.out[j][g] = h;
            }
        };
        // one ksize after the other on the same registers (a fence keeps the scheduler from interleaving three hash pipelines)
        // (the window passes through an empty asm between the phases: without it the compiler shares the upper-cased and
        //  complemented words of the three phases and overlaps their hash pipelines -- 219 registers, two waves per SIMD)
        auto fence = [&]() {
#pragma unroll
            for (int i = 0; i < LANE_RD; ++i) asm volatile("" : "+v"(raw[i]));
            __builtin_amdgcn_sched_barrier(0);
        };
        process_lane<KA, P, true>(raw, seed, a.thr[0], [&](int o, uint64_t h) { keep(0, o, h); });
        fence();
        process_lane<KB, P, true>(raw, seed, a.thr[1], [&](int o, uint64_t h) { keep(1, o, h); });
        fence();
        process_lane<KC, P, true>(raw, seed, a.thr[2], [&](int o, uint64_t h) { keep(2, o, h); });
        __syncthreads();
        flush(false);
    }
    __syncthreads();
    flush(true);
}

template <int KA, int KB, int KC>
hipError_t launch_multi(const uint8_t* d_new, uint64_t n_new, uint64_t seed, const SketchMultiOut* o, hipStream_t stream) {
    constexpr uint64_t TILE = (uint64_t)SK_BLOCK * 16;
    const uint8_t* seq = d_new - (KC - 1);
    const uint32_t skip = (uint32_t)((uintptr_t)seq & 15);
    seq -= skip;
    const uint64_t len = (uint64_t)skip + (uint64_t)(KC - 1) + n_new;
    const uint64_t n_tiles = (len + TILE - 1) / TILE;
    if (n_tiles == 0) return hipSuccess;
    MultiArgs a;
    const int ks[SM_N] = {KA, KB, KC};
    for (int j = 0; j < SM_N; ++j) {
        a.thr[j] = o[j].thr; a.out[j] = o[j].out; a.count[j] = o[j].count; a.cap[j] = o[j].cap;
        a.start[j] = skip + (uint32_t)(KC - ks[j]);
    }
    const uint64_t max_blocks = 256ull * 8;
    const unsigned grid = (unsigned)(n_tiles < max_blocks ? n_tiles : max_blocks);
    hipLaunchKernelGGL((sketch_dna_multi_kernel<KA, KB, KC>), dim3(grid), dim3(SK_BLOCK), 0, stream, seq, len, seed, a, n_tiles, skip);
    return hipGetLastError();
}

}  // namespace

bool sketch_dna_multi_supported(const uint32_t* ks, int n) { return n == 3 && ks[0] == 21 && ks[1] == 31 && ks[2] == 51; }

// d_new: the first NEW byte; the (largest k) - 1 bytes in front of it are readable and hold the stream's previous bytes (or
// separators); outs[j] belongs to ks[j], ascending.  hipErrorNotSupported: the caller launches per ksize.
hipError_t sketch_dna_multi_launch(const uint8_t* d_new, uint64_t n_new, const uint32_t* ks, int n, uint64_t seed, const SketchMultiOut* outs,
                                   hipStream_t stream) {
    if (!sketch_dna_multi_supported(ks, n)) return hipErrorNotSupported;
    if (n_new == 0) return hipSuccess;
    return launch_multi<21, 31, 51>(d_new, n_new, seed, outs, stream);
}

}  // namespace smg
