// gather_parts.hpp -- what gather_build.hip (the index build), gather.hip (rounds, resident loop) and overlap.hip (the streaming walks over a
// collection: the overlap pass of search / prefetch, and pass 1 + 2a of the index build) share.  Internal to the three units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <chrono>
#include "gather_api.hpp"
#include "arena.hpp"
#include "qindex.hpp"

namespace smg {

#define SMG_TRY(expr)                      \
    do {                                   \
        hipError_t e_ = (expr);            \
        if (e_ != hipSuccess) return e_;   \
    } while (0)

// the final scatter of the index build works by windows of BR_SUB posting lists; a staged posting is (row << BR_SUB_BITS) | list
// within its window
constexpr int BR_SUB = 256;
constexpr int BR_SUB_BITS = 8;

// gather.hip: T[b] = first position of the sorted query with Q >= b << shift, b = 0 .. buckets (qindex.hpp: qindex_geometry)
hipError_t qtable_launch(const uint64_t* Q, uint64_t nq, uint32_t shift, uint32_t buckets, uint32_t* table, hipStream_t stream);

// overlap.hip: the builder's pass 1 + 2a through the lean streaming kernel's staging form
uint32_t build_stage_positions(uint64_t nq, uint32_t buckets, double mean_row);
uint32_t build_stage_buckets_max();
uint32_t build_stage_rows_max();
size_t build_stage_desc_bytes(uint32_t n_ranges);
hipError_t build_stage_plan(const uint64_t* Q, uint64_t nq, uint32_t shift, uint32_t n_buckets, uint32_t W, uint32_t n_ranges, void* desc,
                            unsigned int* max_nb, hipStream_t stream);
hipError_t build_stage_launch(const uint64_t* Q, uint64_t nq, const uint32_t* T, uint32_t n_buckets, uint32_t shift, const uint64_t* hashes,
                              const uint64_t* offsets, uint64_t ndb, uint32_t rows_per_wg, uint32_t n_ranges, const void* desc,
                              unsigned long long* counters, uint32_t* qpos, uint32_t* inter, uint32_t* dir_start, uint32_t* dir_len,
                              unsigned int* misc, hipStream_t stream);
void lean_table_geometry(uint64_t nq, uint64_t q_max, double mean_row, uint32_t* shift, uint32_t* buckets);
struct LeanPlan { uint32_t bpr, n_ranges, qcap, rows_cap; };
LeanPlan build_lean_plan(uint64_t nq, uint32_t buckets, double mean_row);

// ---- shared by gather_build.hip and gather.hip ----
__device__ __forceinline__ unsigned long long wave_max(unsigned long long k) {
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_down(k, off);
        k = o > k ? o : k;
    }
    return k;
}

static inline QIndex qindex_of(const GatherDev& g) { return QIndex{g.q_padded, g.nq, g.q_table, g.q_shift, g.q_max, g.q_rec}; }



// Owned buffers and build scratch come from the arena (arena.hpp): blocks the library keeps between builds, so that a
// rebuild of the same shape makes no driver call.  (Round 2 used hipMallocAsync with a raised release threshold; on the
// benchmark host that still cost 175 ms per build against 6.9 ms of kernels -- VERDICT r02.)
template <class T>
static inline hipError_t own_alloc(GatherDev& g, T** p, size_t bytes, hipStream_t user = nullptr) {
    return arena_alloc((void**)p, bytes, user ? user : g.stream);
}

static inline uint64_t host_ns() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static inline hipError_t timed_sync(GatherDev& g, hipStream_t stream) {
    const uint64_t t0 = host_ns();
    const hipError_t e = hipStreamSynchronize(stream);
    g.build_sync_wait_ns += host_ns() - t0;
    g.build_syncs++;
    return e;
}


}  // namespace smg
