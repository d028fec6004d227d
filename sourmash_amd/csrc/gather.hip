// gather.hip -- min-set-cover gather with every round resident on the GPU: the rounds (the index they run on: gather_build.hip).
//
// What the reference does per round (src/sourmash/index/__init__.py:735-909 CounterGather, driven by
// src/sourmash/search.py:877-949 GatherDatabases.__next__):
//   peek     pick the dataset with the largest remaining overlap (ties: first inserted), stop when the overlap is
//            below threshold_bp / scaled hashes (search.py:15-37) or the query is exhausted;
//   I        = current query ∩ match;     current query -= match;
//   consume  for every remaining dataset d: counter[d] -= |I ∩ D_d| (drop at 0).
// consume is the expensive line: the reference walks the whole database every round.  Here the database is
// inverted once against the query (gather_build.hip: hash position -> rows containing it, a CSR of u32 row ids), so a round touches
// only the postings of the hashes in I: total work over a whole gather is sum_d |Q ∩ D_d| counter decrements, the
// same number the reference spends on round 0 alone.  Since round 3 the whole loop is ONE resident kernel where the index
// allows it (gather_loop_kernel: a workgroup per CU owns a row range, counters and the uncovered set live in LDS, one granule
// all-gather per round); otherwise one round = two small kernels (arg-max with stop rules and bookkeeping in its last
// workgroup, apply), enqueued in batches by a host that only looks at a done flag.  Every kernel is a no-op once the flag is set.
//
// Multi-GPU (database sharded by dataset, query replicated): the ranks' resident loop kernels agree on every round's winner
// through host-visible memory shared by the ranks (gather_launch_loop with a GatherShared; sourmash_amd/parallel.py).  The
// older protocol, kept for indexes that cannot run the resident loop: rounds are replayed from exchanged candidates.  Every
// shard exports its K best rows (packed key, hashes) plus the best key it keeps back; ONE all-gather hands all of
// them to every rank, which inverts them against the query as well (one bit per candidate and query hash, cmask) so
// that apply keeps the candidates' counters as exact as the local ones.  The winner of a round is then the best
// candidate -- provably the global arg-max while its key is not below any kept-back key (counters only decrease) --
// and rounds run back to back with no exchange until that test fails.  See sourmash_amd/parallel.py.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <stdlib.h>
#include <rocprim/device/device_scan.hpp>
#include "gather_api.hpp"
#include "arena.hpp"
#include "wavemask.hpp"
#include <chrono>
#include "qindex.hpp"
#include "gather_parts.hpp"

namespace smg {

namespace {

__device__ __forceinline__ void record_pending(unsigned long long* state, uint64_t* out_idx, uint64_t* out_isect) {
    if (state[GS_PENDING]) {
        const unsigned long long r = state[GS_ROUNDS], acc = state[GS_ACC];
        out_idx[r] = 0xffffffffull & ~state[GS_KEY];
        out_isect[r] = acc;
        state[GS_QLEN] -= acc;
        state[GS_ACC] = 0;
        state[GS_ROUNDS] = r + 1;
        state[GS_PENDING] = 0;
        if (r + 1 >= state[GS_MAXR]) state[GS_DONE] = 1;
    }
}

// search.py:15-37 + index/__init__.py:817-861: stop when nothing overlaps, the query is exhausted, the threshold
// is unattainable (more hashes than the query has left) or the best overlap is below it
__device__ __forceinline__ void stop_rules(unsigned long long* state, unsigned long long key) {
    const unsigned long long count = key >> 32, qlen = state[GS_QLEN], thr = state[GS_THR];
    if (key == 0 || qlen == 0 || qlen < thr || count < thr) state[GS_DONE] = 1;
    else state[GS_PENDING] = 1;
}



// Packed arg-max with the reference tie-break (highest count, then lowest global index) over all counters, then --
// in the workgroup that finishes last (ticket counter) -- the bookkeeping of the previous round, the final reduction
// and, if asked, the stop rules.  (Measured: the same speed as a partial + a final launch -- a round is bound by the
// dependency between its kernels, not by their number -- but one kernel less to reason about.)
__global__ __launch_bounds__(256) void pick_kernel(const unsigned long long* __restrict__ counters, uint64_t ndb,
                                                   uint64_t index_base, unsigned long long* state,
                                                   unsigned long long* partials, uint64_t* out_idx, uint64_t* out_isect,
                                                   unsigned long long* key_out, int check_stop) {
    __shared__ unsigned long long red[4];
    __shared__ int s_last;
    if (state[GS_DONE]) {                                           // stable for the whole launch: only a last block sets it
        if (blockIdx.x == 0 && threadIdx.x == 0 && key_out) *key_out = 0;
        return;
    }
    unsigned long long k = 0;
    for (uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; d < ndb; d += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long c = counters[d];
        if (c) {
            const unsigned long long key = (c << 32) | (0xffffffffull & ~(unsigned long long)(index_base + d));
            k = key > k ? key : k;
        }
    }
    k = wave_max(k);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = k;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) k = red[w] > k ? red[w] : k;
        partials[blockIdx.x] = k;
        __threadfence();                                            // the partial is visible before the ticket is
        s_last = atomicAdd(&state[GS_TICKET], 1ull) == (unsigned long long)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) {
        state[GS_TICKET] = 0;
        record_pending(state, out_idx, out_isect);
    }
    __syncthreads();
    if (state[GS_DONE]) {
        if (threadIdx.x == 0 && key_out) *key_out = 0;
        return;
    }
    const volatile unsigned long long* vp = partials;
    k = 0;
    for (unsigned i = threadIdx.x; i < gridDim.x; i += blockDim.x) { const unsigned long long v = vp[i]; k = v > k ? v : k; }
    k = wave_max(k);
    __syncthreads();                                                // red[] is reused
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = k;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) k = red[w] > k ? red[w] : k;
        state[GS_KEY] = k;
        if (key_out) *key_out = k;
        if (check_stop) stop_rules(state, k);
    }
}

// ---- candidate replay ---------------------------------------------------------------------------------------------
// Top-K selection: every thread keeps the sorted list of its TOPK_LIST best keys, a wave merges its 64 lists by
// TOPK_LIST rounds of wave-max (the lane that owned the maximum shifts its list), wave 0 merges the 4 wave lists the
// same way; the last workgroup to finish (ticket) merges the per-workgroup lists.  Keys are distinct (the index is
// part of the key), zeros are "nothing".
constexpr int TOPK_LIST = (int)GATHER_TOPK_MAX + 1;

__device__ __forceinline__ void topk_insert(unsigned long long (&best)[TOPK_LIST], unsigned long long key) {
    if (key <= best[TOPK_LIST - 1]) return;
    best[TOPK_LIST - 1] = key;
#pragma unroll
    for (int i = TOPK_LIST - 1; i > 0; --i) {
        const unsigned long long a = best[i - 1], b = best[i];
        best[i - 1] = a > b ? a : b;
        best[i] = a > b ? b : a;
    }
}

// all lanes of the wave get the wave's `want` best keys in out[] (descending); best[] is consumed
__device__ __forceinline__ void topk_wave_merge(unsigned long long (&best)[TOPK_LIST], int want,
                                                unsigned long long (&out)[TOPK_LIST]) {
#pragma unroll
    for (int it = 0; it < TOPK_LIST; ++it) {
        unsigned long long m = 0;
        if (it < want) {
            m = best[0];
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(m, off);
                m = o > m ? o : m;
            }
            if (m != 0 && best[0] == m) {
#pragma unroll
                for (int i = 0; i + 1 < TOPK_LIST; ++i) best[i] = best[i + 1];
                best[TOPK_LIST - 1] = 0;
            }
        }
        out[it] = m;
    }
}

// workgroup-wide: thread lists -> s_list[0][0..want) (valid after the call for every thread)
__device__ __forceinline__ void topk_block_merge(unsigned long long (&best)[TOPK_LIST], int want,
                                                 unsigned long long (*s_list)[TOPK_LIST]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long out[TOPK_LIST];
    topk_wave_merge(best, want, out);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < TOPK_LIST; ++i) s_list[wave][i] = out[i];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < TOPK_LIST; ++i) best[i] = lane < 4 ? s_list[lane][i] : 0ull;
        topk_wave_merge(best, want, out);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < TOPK_LIST; ++i) s_list[0][i] = out[i];
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void topk_kernel(const unsigned long long* __restrict__ counters, uint64_t ndb,
                                                   uint64_t index_base, unsigned long long* state,
                                                   unsigned long long* partials, unsigned long long* sel, int want) {
    __shared__ unsigned long long s_list[4][TOPK_LIST];
    __shared__ int s_last;
    if (state[GS_DONE]) return;
    unsigned long long best[TOPK_LIST];
#pragma unroll
    for (int i = 0; i < TOPK_LIST; ++i) best[i] = 0;
    for (uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; d < ndb; d += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long c = counters[d];
        if (c) topk_insert(best, (c << 32) | (0xffffffffull & ~(unsigned long long)(index_base + d)));
    }
    topk_block_merge(best, want, s_list);
    if (threadIdx.x < TOPK_LIST)
        __hip_atomic_store(&partials[(uint64_t)blockIdx.x * TOPK_LIST + threadIdx.x], s_list[0][threadIdx.x],
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                            // the partial list is visible before the ticket is
        s_last = atomicAdd(&state[GS_TICKET], 1ull) == (unsigned long long)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) state[GS_TICKET] = 0;
#pragma unroll
    for (int i = 0; i < TOPK_LIST; ++i)                             // thread t adopts workgroup t's (sorted) list
        best[i] = threadIdx.x < gridDim.x
                      ? __hip_atomic_load(&partials[(uint64_t)threadIdx.x * TOPK_LIST + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                      : 0ull;
    topk_block_merge(best, want, s_list);
    if (threadIdx.x < TOPK_LIST) sel[threadIdx.x] = threadIdx.x < want ? s_list[0][threadIdx.x] : 0ull;
}

// records [key, bound, len, hashes...] of the K selected rows; bound = sel[K] (the best key kept back)
__global__ __launch_bounds__(256) void export_cands_kernel(const unsigned long long* state,
                                                           const unsigned long long* __restrict__ sel, uint32_t K,
                                                           const uint64_t* __restrict__ hashes,
                                                           const uint64_t* __restrict__ offsets, uint64_t ndb,
                                                           uint64_t index_base, uint64_t* __restrict__ out, uint64_t stride) {
    const uint32_t c = blockIdx.y;
    uint64_t* rec = out + (uint64_t)c * stride;
    const unsigned long long key = state[GS_DONE] ? 0ull : sel[c];
    uint64_t lo = 0, len = 0;
    if (key) {
        const uint64_t d = (0xffffffffull & ~key) - index_base;
        lo = offsets[d];
        len = offsets[d + 1] - lo;
        if (len + GATHER_CAND_HEAD > stride) len = stride - GATHER_CAND_HEAD;   // cannot happen when stride covers the longest row
    }
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len + GATHER_CAND_HEAD;
         i += (uint64_t)gridDim.x * blockDim.x)
        rec[i] = i == 0 ? key : i == 1 ? (state[GS_DONE] ? 0ull : sel[K]) : i == 2 ? len : hashes[lo + i - GATHER_CAND_HEAD];
}

// take the bits of the previous exchange's candidates out of cmask (one workgroup per candidate)
__global__ __launch_bounds__(256) void cands_clear_kernel(uint64_t* __restrict__ cmask, const uint32_t* __restrict__ cand_len,
                                                          const uint32_t* __restrict__ cand_qpos, uint64_t qstride) {
    const uint32_t c = blockIdx.x, len = cand_len[c];
    const uint32_t* pos = cand_qpos + (uint64_t)c * qstride;
    for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) {
        const uint32_t j = pos[i];
        if (j != NONE32) cmask[j] = 0;
    }
}

// adopt the gathered records: query positions of candidate c's hashes, bit c in cmask for the uncovered ones, counters
// from the keys, the largest kept-back key; re-arms the replay (GS_NEEDX = 0)
__global__ __launch_bounds__(256) void cands_load_kernel(QIndex qi, const uint8_t* __restrict__ alive,
                                                         const uint64_t* __restrict__ cands, uint32_t n_cand, uint64_t stride,
                                                         uint64_t* cmask, unsigned long long* __restrict__ cand_count,
                                                         unsigned long long* __restrict__ cand_key,
                                                         uint32_t* __restrict__ cand_len, uint32_t* __restrict__ cand_qpos,
                                                         uint64_t qstride, unsigned long long* state) {
    const uint32_t c = blockIdx.x;
    const uint64_t* rec = cands + (uint64_t)c * stride;
    const unsigned long long key = rec[0];
    const uint32_t len = key ? (uint32_t)rec[2] : 0u;
    if (threadIdx.x == 0) {
        cand_key[c] = key;
        cand_count[c] = key >> 32;
        cand_len[c] = len;
        if (c == 0) {
            unsigned long long bound = 0;
            for (uint32_t k = 0; k < n_cand; ++k) {
                const uint64_t* r = cands + (uint64_t)k * stride;
                if (r[0] && r[1] > bound) bound = r[1];
            }
            state[GS_BOUND] = bound;
            state[GS_NEEDX] = 0;
        }
    }
    uint32_t* pos = cand_qpos + (uint64_t)c * qstride;
    for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) {
        const uint32_t j = q_find(qi, rec[GATHER_CAND_HEAD + i]);
        pos[i] = j;
        if (j != NONE32 && alive[j]) atomicOr((unsigned long long*)&cmask[j], 1ull << c);
    }
}

// one wave: bookkeeping of the previous round, then the best candidate.  It is the global arg-max as long as its key
// is not below the largest key any shard kept back: rows outside the candidate set had smaller keys at the exchange
// and keys only decrease.  Otherwise the round is left to the next exchange (GS_NEEDX).
__global__ __launch_bounds__(64) void replay_pick_kernel(const unsigned long long* __restrict__ cand_count,
                                                         const unsigned long long* __restrict__ cand_key, uint32_t n_cand,
                                                         unsigned long long* state, uint64_t* out_idx, uint64_t* out_isect) {
    if (state[GS_DONE]) return;
    const int lane = threadIdx.x;
    if (lane == 0) record_pending(state, out_idx, out_isect);
    __syncthreads();
    if (state[GS_DONE] || state[GS_NEEDX]) return;
    unsigned long long key = 0;
    if ((uint32_t)lane < n_cand) {
        const unsigned long long cnt = cand_count[lane];
        if (cnt && cand_key[lane]) key = (cnt << 32) | (0xffffffffull & cand_key[lane]);
    }
    unsigned long long best = key;
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(best, off);
        best = o > best ? o : best;
    }
    const unsigned long long who = __ballot(key == best && key != 0);
    if (lane == 0) {
        if (best < state[GS_BOUND]) {
            state[GS_NEEDX] = 1;
        } else {
            state[GS_KEY] = best;
            state[GS_WSLOT] = who ? (unsigned long long)__ffsll((long long)who) - 1ull : 0ull;
            stop_rules(state, best);
        }
    }
}

// A wave takes APPLY_EPW hashes of the row: one lane each looks its hash up in the query; the postings of the
// wave's hits are then treated as ONE list (prefix sums over the hits) that all 64 lanes walk together, so short
// and long posting lists keep the lanes equally busy and every load / atomic of an iteration is independent.
// Few hashes per wave and a grid of 4096 waves: a round is one dependent chain (row -> table -> bucket -> postings ->
// counters), so its length is the postings one wave has to walk (C5: 33 us per round with 16 hashes per wave on 512
// waves, 29 us with 2 on 4096).
constexpr int APPLY_EPW = 2;

// The replay form (cands != nullptr) reads the winner from its candidate record, whose query positions cands_load
// left in cand_qpos, and keeps the candidates' counters exact through cmask.
struct CandView {
    const uint64_t* cands;          // records [key, bound, len, hashes...]; null: not a replay round
    uint64_t stride;
    const uint32_t* qpos;           // [n_cand][qstride]
    uint64_t qstride;
    const uint64_t* cmask;
    unsigned long long* count;
};

template <bool GATE>
__global__ __launch_bounds__(256) void apply_kernel(QIndex qi, uint8_t* alive, const uint64_t* __restrict__ post_off,
                                                    const uint32_t* __restrict__ post_rows,
                                                    unsigned long long* counters, unsigned long long* state,
                                                    const uint64_t* __restrict__ rowbuf,
                                                    const uint64_t* __restrict__ hashes,
                                                    const uint64_t* __restrict__ offsets, uint64_t index_base,
                                                    const uint32_t* __restrict__ qpos, CandView cv) {
    if (GATE && (state[GS_DONE] || state[GS_NEEDX])) return;
    const uint64_t* row;
    const uint32_t* row_pos = nullptr;                              // query positions of the row, when they are known
    uint64_t len;
    if (cv.cands) {
        const uint64_t slot = state[GS_WSLOT];
        const uint64_t* rec = cv.cands + slot * cv.stride;
        len = rec[2];
        row = rec + GATHER_CAND_HEAD;
        row_pos = cv.qpos + slot * cv.qstride;
    } else if (rowbuf) {
        len = rowbuf[0];
        row = rowbuf + 1;
    } else {
        const uint64_t d = (0xffffffffull & ~state[GS_KEY]) - index_base;
        row = hashes + offsets[d];
        row_pos = qpos + offsets[d];
        len = offsets[d + 1] - offsets[d];
    }
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    for (uint64_t base = wave * APPLY_EPW; base < len; base += n_waves * APPLY_EPW) {
        const uint64_t i = base + lane;
        uint32_t j = NONE32;
        uint64_t list_lo = 0, list_hi = 0;                         // posting list of the hash: asked for together with its alive byte
        if (lane < APPLY_EPW && i < len) {
            j = row_pos ? row_pos[i] : q_find(qi, row[i]);
            if (j != NONE32) {
                list_lo = post_off[j];
                list_hi = post_off[j + 1];
                // only hashes still uncovered count: they leave the set here, and counters[d] stays |row_d ∩ uncovered|
                // whatever the caller hands to consume (the same intersect twice, hashes it never peeked)
                if (alive[j]) alive[j] = 0;                    // row hashes are distinct: no two lanes share j
                else j = NONE32;
            }
            if (j != NONE32 && cv.cands) {                     // the candidates holding this hash lose it too
                uint64_t m = cv.cmask[j];
                while (m) {
                    atomicAdd(&cv.count[__ffsll((long long)m) - 1], ~0ull);
                    m &= m - 1;
                }
            }
        }
        const unsigned long long hits = __ballot(j != NONE32);
        if (lane == 0 && hits) {
            if (GATE) atomicAdd(&state[GS_ACC], (unsigned long long)__popcll(hits));
            else atomicAdd(&state[GS_QLEN], 0ull - (unsigned long long)__popcll(hits));   // protocol path: a later fused run starts from the truth
        }
        if (!hits) continue;
        // lane l < APPLY_EPW: posting list [lo, lo + n) of its hit; inclusive prefix sums over those lanes
        uint64_t lo = 0;
        uint32_t n = 0;
        if (j != NONE32) {
            lo = list_lo;
            n = (uint32_t)(list_hi - list_lo);
        }
        uint32_t incl = n;
#pragma unroll
        for (int d = 1; d < APPLY_EPW; d <<= 1) {
            const uint32_t v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        const uint32_t total = __shfl(incl, APPLY_EPW - 1);
        uint32_t bound[APPLY_EPW - 1];                          // wave-uniform: end of the lists of hits 0 .. 14
#pragma unroll
        for (int k = 0; k < APPLY_EPW - 1; ++k) bound[k] = __shfl(incl, k);
        const uint32_t excl = incl - n;
        const uint32_t lo_lo = (uint32_t)lo, lo_hi = (uint32_t)(lo >> 32);
        // wave-uniform trip count (the shuffles read lanes 0 .. 15, which must not have left the loop); four
        // independent loads are in flight before their atomics are issued
        constexpr int UNROLL = 4;
        for (uint32_t t0 = 0; t0 < total; t0 += 64 * UNROLL) {
            uint32_t target[UNROLL];
            bool live[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const uint32_t t = t0 + (uint32_t)(64 * u + lane);
                int h = 0;
#pragma unroll
                for (int k = 0; k < APPLY_EPW - 1; ++k) h += t >= bound[k];
                const uint64_t start = ((uint64_t)(uint32_t)__shfl((int)lo_hi, h) << 32) | (uint32_t)__shfl((int)lo_lo, h);
                const uint32_t first = (uint32_t)__shfl((int)excl, h);
                live[u] = t < total;
                target[u] = live[u] ? post_rows[start + (t - first)] : 0u;
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                if (!live[u]) continue;
                unsigned long long* c = &counters[target[u]];
                if (GATE) {
                    atomicAdd(c, ~0ull);                       // invariant: counters[d] = |row_d ∩ uncovered| >= 1 here
                } else {
                    // a caller may have overwritten counters (smgpu_counter_set): never step below zero
                    unsigned long long cur = *c;
                    while (cur != 0) {
                        const unsigned long long prev = atomicCAS(c, cur, cur - 1);
                        if (prev == cur) break;
                        cur = prev;
                    }
                }
            }
        }
    }
}

// ---- persistent loop: every round of the gather inside ONE launch --------------------------------------------------
// The two-kernel round above is a chain of dependent trips to memory (winner key -> row offset -> positions -> alive ->
// post_off -> postings -> counters, then the arg-max over all counters through a ticket) plus one device-scope atomic
// per posting: 28.7 us per round at C5, of which 7.5 are the atomics at the device's rate (profiles/r02_gather_loops.txt).
// Here one workgroup per CU stays resident for the whole loop and OWNS a contiguous range of rows:
//   * its rows' counters live in LDS (plain LDS atomics; nobody else touches them), so a round has no device-scope
//     atomic at all and the local arg-max needs no pass over global memory;
//   * the uncovered set of the query is replicated as a bitmap in every workgroup's LDS (10^6 hashes = 125 KB of the
//     160): test-and-clear is an LDS atomicAnd, deterministic and identical everywhere, so the set is never exchanged;
//   * every workgroup walks the whole winning row (its query positions, written by the build's pass 1) and, for each
//     newly covered hash, only the slice of the posting list that can hold ITS rows: pass 2b of the build writes a
//     list as consecutive runs, one per row block in ascending order, whose boundaries are the per-block prefixes of pass 1;
//   * the only exchange per round is an all-gather of one 32-byte record per workgroup -- local best key, start and
//     length of that row -- as four 8-byte {epoch, value} granules written by ONE write-through store each and swept
//     by every workgroup until all tags carry the round's epoch: barrier and data in one hop (MI355X_MICROARCH.md,
//     "allgather" / R2 granules; no fence needed because nothing else mutable is shared: counters and bitmap are
//     private, postings / positions / offsets are immutable during the loop).
// Rounds, stop rules (search.py:15-37) and bookkeeping are replicated arithmetic on replicated values, so every
// workgroup leaves the loop in the same round; workgroup 0 records the results.  At exit the counters and the bitmap
// go back to the global arrays, so that the protocol entry points (peek / consume, counters_get, a second begin + run)
// carry on from the same state.
constexpr int PL_THREADS = 1024;
constexpr int PL_ROW_PER = 6;                  // row elements per thread and chunk (chunk <= PL_ROW_PER * PL_THREADS): a C5 row (~5,000
                                               // hashes) is one chunk -- a second chunk costs another pair of barriers and LDS passes per round
constexpr int PL_LANES = 2;                    // lanes per newly covered position when its row block's run is read (2 x 2 x 4 entries;
                                               // a run is about 4 entries at C5: 250 per list over 64 row blocks)
constexpr int PL_STEPS = 3;                    // positions per lane group whose loads are in flight together (3 x 512 per batch)
constexpr uint32_t PL_SPIN_LIMIT = 1u << 22;   // sweeps before a workgroup gives up (a peer is not resident): seconds, not forever

struct LoopArgs {
    const uint64_t* offsets;
    const uint32_t* qpos;
    const uint32_t* post_rows;
    const uint32_t* block_pre;          // [nq][B + 1] absolute start of row block b's run in list j (entry B: the list's end)
    const uint64_t* post_off;
    unsigned long long* counters;
    uint8_t* alive;
    unsigned long long* state;
    uint64_t* out_idx;
    uint64_t* out_isect;
    unsigned long long* xchg;           // [2][n_wg * 4] granules, zeroed before the launch
    uint64_t ndb, nq, index_base;
    uint32_t rows_per_wg, block_rows, B, chunk, bitmap_words, prefetch;
    unsigned long long* dbg;            // SMG_GATHER_TRACE: [8] phase times of workgroup 0 in 10 ns ticks (null: not collected)
    // ---- several ranks, one database shard each (W > 0): the round's winner is agreed through host-visible memory ----
    uint32_t W, rank, epoch_base, rowcap;   // ranks, this rank, tag offset of this run, granules per row slot
    unsigned long long* x_rec;          // shared: [2][W][4] record granules {key hi, key lo, row length, -}
    unsigned long long* x_rows;         // shared: [2][W][rowcap] {tag, query position} granules: every rank's local best row
    // ... or (peers != 0) one area PER RANK in that rank's own device memory, mapped into every peer (hipIpc: over xGMI between the
    // GPUs of a node): a rank WRITES only its own area -- local, write-through -- and READS the others'.  Area of rank r:
    // [2][4] record granules, then [2][rowcap] row granules.
    uint32_t peers;
    unsigned long long* peer[GATHER_PEERS_MAX];
    unsigned long long* gwin;           // local:  [2][4] the global winner {key hi, key lo, length, owner rank}
    unsigned long long* gate;           // local:  [0] workgroups that have started, bit 63: somebody gave up waiting for the rest
    unsigned long long gate_ticks;      // how long a workgroup waits at the gate for the others to start (wall_clock64 ticks, 10 ns)
    unsigned long long* stage;          // local:  [2][rowcap] the winner's positions, copied in once per round by a few workgroups
};

__device__ __forceinline__ unsigned long long gran_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gran_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the same across devices / processes: host-visible (pinned, coherent) memory, system scope
__device__ __forceinline__ unsigned long long sys_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void sys_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
constexpr uint32_t PL_STAGERS = 8;             // workgroups that copy a remote winner's row into local memory
// where rank r's record / row granules of parity `par` live: in the one shared area, or in r's own area
__device__ __forceinline__ unsigned long long* xchg_rec(const LoopArgs& a, uint32_t par, uint32_t r) {
    return a.peers ? a.peer[r] + (uint64_t)par * 4 : a.x_rec + ((uint64_t)par * a.W + r) * 4;
}
__device__ __forceinline__ unsigned long long* xchg_row(const LoopArgs& a, uint32_t par, uint32_t r) {
    return a.peers ? a.peer[r] + 8 + (uint64_t)par * a.rowcap : a.x_rows + ((uint64_t)par * a.W + r) * a.rowcap;
}

__global__ __launch_bounds__(PL_THREADS) void gather_loop_kernel(LoopArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t pl_lds[];
    uint32_t* s_bits = pl_lds;                                   // [bitmap_words] bit p set: query hash p is uncovered
    uint32_t* s_cnt = s_bits + a.bitmap_words;                   // [rows_per_wg] |row ∩ uncovered| of the owned rows
    uint32_t* s_off = s_cnt + a.rows_per_wg;                     // [rows_per_wg + 1] element offsets of the owned rows
    uint32_t* s_I = s_off + a.rows_per_wg + 1;                   // [chunk] newly covered query positions of the chunk in flight
    __shared__ unsigned long long s_red[PL_THREADS / 64];
    __shared__ uint32_t s_nI, s_wstart, s_wlen;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t wg = blockIdx.x, n_wg = gridDim.x;
    // Which rows a workgroup owns: consecutive ranges go to workgroups on the SAME XCD (workgroup b runs on XCD b % 8 --
    // an observation that only buys speed): the 32 workgroups that read a group's runs then share one L2.
    const uint32_t own = (n_wg % 8u == 0u) ? (wg % 8u) * (n_wg / 8u) + wg / 8u : wg;
    const uint64_t r0 = (uint64_t)own * a.rows_per_wg < a.ndb ? (uint64_t)own * a.rows_per_wg : a.ndb;
    const uint64_t r1 = r0 + a.rows_per_wg < a.ndb ? r0 + a.rows_per_wg : a.ndb;
    const uint32_t n_own = (uint32_t)(r1 - r0);
    // the row blocks whose runs hold rows r0 .. r1 - 1 (one block when the ranges are aligned, else two)
    const uint32_t b0 = n_own ? (uint32_t)(r0 / a.block_rows) : 0u;
    const uint32_t b1 = n_own ? (uint32_t)((r1 - 1) / a.block_rows) : 0u;
    // ---- the gate: nothing is touched before ALL workgroups of the grid are running ----
    // The sweeps below wait for every workgroup, so all of them must be resident at once.  On an idle device a grid of one
    // workgroup per CU is; with another kernel or another process on the device some may only start when others leave.  One word
    // decides for the whole grid: it counts arrivals and reaches n_wg, or a workgroup that waited gate_ticks sets bit 63 by
    // compare-and-swap while the count is still short (no CAS can succeed once everybody is there).  Given up: every
    // workgroup -- those that start later included -- leaves at once with counters, uncovered set and results untouched, and
    // the host runs the two-kernel rounds on the same state (GS_ERR = 10; capi.cpp: gather_drain).
    {
        constexpr unsigned long long GAVE_UP = 1ull << 63;
        if (tid == 0) {
            unsigned long long v = atomicAdd(a.gate, 1ull) + 1ull;
            const unsigned long long t0 = wall_clock64();
            while (!(v & GAVE_UP) && v < (unsigned long long)n_wg) {
                if (wall_clock64() - t0 > a.gate_ticks) {
                    const unsigned long long seen = atomicCAS(a.gate, v, v | GAVE_UP);
                    if (seen == v) {                                  // this workgroup called it off: tell the host, and the other ranks
                        v |= GAVE_UP;
                        a.state[GS_ERR] = 10;
                        a.state[13] = 0; a.state[14] = wg;
                        if (a.W > 0) {                                // (their workgroup 0 polls this record in epoch 1)
                            const unsigned long long gtag = (unsigned long long)(a.epoch_base + 1u) << 32;
                            unsigned long long* rec = xchg_rec(a, 1u, a.rank);
                            sys_store(rec + 0, gtag); sys_store(rec + 1, gtag); sys_store(rec + 2, gtag | 0xffffffffull);
                        }
                    } else v = seen;
                    continue;
                }
                __builtin_amdgcn_s_sleep(8);
                v = gran_load(a.gate);
            }
            s_nI = (v & GAVE_UP) ? 1u : 0u;
        }
        __syncthreads();
        if (s_nI) return;
        __syncthreads();
    }
    // ---- load the state this loop starts from ----
    for (uint32_t w = tid; w < a.bitmap_words; w += PL_THREADS) {
        uint32_t bits = 0;
        const uint64_t p0 = (uint64_t)w * 32;
        if (p0 + 32 <= a.nq) {
            const uint4 lo = *reinterpret_cast<const uint4*>(a.alive + p0), hi = *reinterpret_cast<const uint4*>(a.alive + p0 + 16);
            const uint32_t v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int b = 0; b < 4; ++b) bits |= ((v[k] >> (8 * b)) & 0xffu ? 1u : 0u) << (4 * k + b);
        } else {
            for (uint32_t b = 0; b < 32 && p0 + b < a.nq; ++b) bits |= (a.alive[p0 + b] ? 1u : 0u) << b;
        }
        s_bits[w] = bits;
    }
    for (uint32_t i = tid; i < a.rows_per_wg; i += PL_THREADS) s_cnt[i] = i < n_own ? (uint32_t)a.counters[r0 + i] : 0u;
    for (uint32_t i = tid; i <= a.rows_per_wg; i += PL_THREADS) s_off[i] = (uint32_t)a.offsets[r0 + (i <= n_own ? i : n_own)];
    unsigned long long qlen = a.state[GS_QLEN], rounds = a.state[GS_ROUNDS];
    const unsigned long long thr = a.state[GS_THR], maxr = a.state[GS_MAXR];
    unsigned long long last_key = 0;
    bool failed = false;
    uint32_t fail_code = 1, fail_epoch = 0;
    __syncthreads();
    const bool timing = a.dbg != nullptr && wg == 0 && tid == 0;
    unsigned long long t_mark = timing ? wall_clock64() : 0, t_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PL_WAITLAP(slot) do { if (a.dbg != nullptr && wg == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PL_LAP(slot); } } while (0)
#define PL_LAP(slot) do { if (timing) { const unsigned long long t_ = wall_clock64(); t_acc[slot] += t_ - t_mark; t_mark = t_; } } while (0)
    if (timing) a.dbg[6] = t_mark;
    for (uint32_t epoch = 1;; ++epoch) {
        fail_epoch = epoch;
        // ---- local best of the owned rows -> this workgroup's record of the epoch ----
        unsigned long long k = 0;
        for (uint32_t i = tid; i < n_own; i += PL_THREADS) {
            const unsigned long long c = s_cnt[i];
            if (c) {
                const unsigned long long key = (c << 32) | (0xffffffffull & ~(unsigned long long)(a.index_base + r0 + i));
                k = key > k ? key : k;
            }
        }
        k = wave_max(k);
        if (lane == 0) s_red[wave] = k;
        __syncthreads();
        if (wave == 0) {                                             // the 16 wave maxima: one LDS read per lane + a wave reduction (was: 16 dependent reads by thread 0)
            k = lane < PL_THREADS / 64 ? s_red[lane] : 0ull;
            k = wave_max(k);
        }
        if (tid == 0) {
            uint32_t start = 0, len = 0;
            if (k) {
                const uint32_t i = (uint32_t)((0xffffffffull & ~k) - a.index_base - r0);
                start = s_off[i];
                len = s_off[i + 1] - start;
            }
            unsigned long long* rec = a.xchg + (uint64_t)(epoch & 1u) * n_wg * 4 + (uint64_t)wg * 4;
            const unsigned long long tag = (unsigned long long)epoch << 32;
            gran_store(rec + 0, tag | (k >> 32));
            gran_store(rec + 1, tag | (k & 0xffffffffull));
            gran_store(rec + 2, tag | start);
            gran_store(rec + 3, tag | len);
            s_wstart = start;                                        // (reused below once the winner is known)
            s_wlen = len;
        }
        __syncthreads();
        // While the records travel: touch the positions of this workgroup's own best row, one lane per 64-byte line.  The
        // round's winner is one of these rows, so its slice is in the memory-side cache (and one XCD's L2) when everybody asks.
        // (Round 5 tried the touches as plain loads issued by the waves that do not sweep, their values folded into a sink a phase
        //  later, so that nobody waits for a touch: 12.75 us per round against 11.8 with the volatile loads below, whose
        //  s_waitcnt vmcnt(0) sits in the shadow of the records' travel -- profiles/r05_gather_prefetch_ab.txt.  Also tried and
        //  dropped: touching the rows of the best records that LOST for the next round (17-20 us: the state carried across the
        //  round's phases cost 100 bytes of scratch per lane).)
        if (a.prefetch) {
            const uint32_t ps = s_wstart, pl = s_wlen;
            for (uint32_t i = (uint32_t)tid * 16u; i < pl; i += PL_THREADS * 16u)
                (void)*reinterpret_cast<const volatile uint32_t*>(a.qpos + (uint64_t)ps + i);   // volatile: issued, its value unused
        }
        PL_LAP(0);                                                   // local arg-max + publish
        // ---- sweep everyone's records until all of them carry this epoch: the winner of the round ----
        const unsigned long long* all = a.xchg + (uint64_t)(epoch & 1u) * n_wg * 4;
        unsigned long long best = 0;
        uint32_t b_start = 0, b_len = 0;
        for (uint32_t spins = 0;; ++spins) {
            bool ok = true;
            best = 0;
            for (uint32_t sidx = tid; sidx < n_wg; sidx += PL_THREADS) {
                const unsigned long long x0 = gran_load(all + (uint64_t)sidx * 4 + 0), x1 = gran_load(all + (uint64_t)sidx * 4 + 1),
                                         x2 = gran_load(all + (uint64_t)sidx * 4 + 2), x3 = gran_load(all + (uint64_t)sidx * 4 + 3);
                ok = ok && (x0 >> 32) == epoch && (x1 >> 32) == epoch && (x2 >> 32) == epoch && (x3 >> 32) == epoch;
                const unsigned long long key = ((x0 & 0xffffffffull) << 32) | (x1 & 0xffffffffull);
                if (key > best) { best = key; b_start = (uint32_t)x2; b_len = (uint32_t)x3; }
            }
            if (__syncthreads_and(ok ? 1 : 0)) break;
            if (spins >= PL_SPIN_LIMIT) { failed = true; fail_code = 11; break; }        // uniform: every thread counts the same sweeps
            __builtin_amdgcn_s_sleep(1);
            if (timing) t_acc[5] += 1;                                  // sweeps that found a record missing
        }
        if (failed) break;
        PL_LAP(1);                                                   // sweep
        const unsigned long long mine = best;
        best = wave_max(best);
        best = __shfl(best, 0);
        if (lane == 0) s_red[wave] = best;
        __syncthreads();
        unsigned long long top = 0;
        for (int w = 0; w < PL_THREADS / 64; ++w) top = s_red[w] > top ? s_red[w] : top;
        if (top != 0 && mine == top) { s_wstart = b_start; s_wlen = b_len; }   // keys are distinct: one writer
        __syncthreads();
        // ---- several ranks: the local winners meet in shared memory; the best of them is the round's winner everywhere ----
        const unsigned long long* stage_row = nullptr;             // non-null: the winner lives on another rank; its positions come from here
        if (a.W > 0) {
            const uint32_t par = epoch & 1u;
            const unsigned long long gtag = (unsigned long long)(a.epoch_base + epoch) << 32, ltag = (unsigned long long)epoch << 32;
            const uint32_t lstart = s_wstart, llen = top ? s_wlen : 0u;
            if (llen > a.rowcap) { if (tid == 0) a.state[GS_ERR] = 4; failed = true; break; }   // (the host sized the slots from the longest row)
            __syncthreads();                                        // (s_wstart / s_wlen are about to be reused)
            // every rank pushes its own best row (query positions, tagged): whoever wins, its row is already on its way
            unsigned long long* my_row = xchg_row(a, par, a.rank);
            for (uint32_t i = wg * PL_THREADS + (uint32_t)tid; i < llen; i += n_wg * PL_THREADS)
                sys_store(my_row + i, gtag | a.qpos[(uint64_t)lstart + i]);
            unsigned long long* gw = a.gwin + (uint64_t)par * 4;
            if (wg == 0) {
                if (tid == 0) {
                    unsigned long long* rec = xchg_rec(a, par, a.rank);
                    sys_store(rec + 0, gtag | (top >> 32));
                    sys_store(rec + 1, gtag | (top & 0xffffffffull));
                    sys_store(rec + 2, gtag | llen);
                }
                // one lane per rank polls that rank's record
                unsigned long long rk = 0;
                uint32_t rl = 0;
                for (uint32_t spins = 0;; ++spins) {
                    bool ok = true;
                    if ((uint32_t)tid < a.W) {
                        const unsigned long long* rec = xchg_rec(a, par, (uint32_t)tid);
                        const unsigned long long x0 = sys_load(rec + 0), x1 = sys_load(rec + 1), x2 = sys_load(rec + 2);
                        ok = (x0 >> 32) == (gtag >> 32) && (x1 >> 32) == (gtag >> 32) && (x2 >> 32) == (gtag >> 32);
                        rk = ((x0 & 0xffffffffull) << 32) | (x1 & 0xffffffffull);
                        rl = (uint32_t)x2;
                    }
                    if (__syncthreads_and(ok ? 1 : 0)) break;
                    if (spins >= PL_SPIN_LIMIT) { failed = true; fail_code = 12; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                // a rank whose grid did not become resident says so in the length field of its epoch-1 record: nobody applies anything
                if (!failed && __syncthreads_or((uint32_t)tid < a.W && rl == 0xffffffffu ? 1 : 0)) {
                    failed = true; fail_code = 14;
                    if (tid == 0) { gran_store(gw + 0, ltag); gran_store(gw + 1, ltag); gran_store(gw + 2, ltag); gran_store(gw + 3, ltag | 0xffffffffull); }
                }
                if (!failed) {
                    // the best of at most 1024 ranks' keys (distinct: the global index is part of the key)
                    unsigned long long m = (uint32_t)tid < a.W ? rk : 0ull;
                    m = wave_max(m);
                    m = __shfl(m, 0);
                    if (lane == 0) s_red[wave] = m;
                    __syncthreads();
                    unsigned long long g = 0;
                    for (int w = 0; w < PL_THREADS / 64; ++w) g = s_red[w] > g ? s_red[w] : g;
                    if (g == 0 && tid == 0) {                        // nobody has anything left
                        gran_store(gw + 0, ltag); gran_store(gw + 1, ltag); gran_store(gw + 2, ltag); gran_store(gw + 3, ltag);
                    } else if ((uint32_t)tid < a.W && rk == g && g != 0) {
                        gran_store(gw + 0, ltag | (g >> 32));
                        gran_store(gw + 1, ltag | (g & 0xffffffffull));
                        gran_store(gw + 2, ltag | rl);
                        gran_store(gw + 3, ltag | (uint32_t)tid);
                    }
                }
            }
            // every workgroup learns the global winner from local memory
            unsigned long long g0v = 0, g1v = 0, g2v = 0, g3v = 0;
            if (!failed)
                for (uint32_t spins = 0;; ++spins) {
                    g0v = gran_load(gw + 0); g1v = gran_load(gw + 1); g2v = gran_load(gw + 2); g3v = gran_load(gw + 3);
                    const bool ok = (g0v >> 32) == epoch && (g1v >> 32) == epoch && (g2v >> 32) == epoch && (g3v >> 32) == epoch;
                    if (__syncthreads_and(ok ? 1 : 0)) break;
                    if (spins >= PL_SPIN_LIMIT) { failed = true; fail_code = 13; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            if (failed) break;
            top = ((g0v & 0xffffffffull) << 32) | (g1v & 0xffffffffull);
            const uint32_t owner = (uint32_t)g3v, glen = (uint32_t)g2v;
            if (owner == 0xffffffffu) { failed = true; fail_code = 14; break; }   // workgroup 0 saw a rank call the run off
            if (tid == 0) { s_wlen = glen; if (owner == a.rank) s_wstart = lstart; }
            __syncthreads();
            if (top != 0 && owner != a.rank) {
                // the winner's positions: copied out of shared memory ONCE per rank by a few workgroups, read locally by all
                unsigned long long* st = a.stage + (uint64_t)par * a.rowcap;
                const unsigned long long* src = xchg_row(a, par, owner);
                if (wg < PL_STAGERS)
                    for (uint32_t i = wg * PL_THREADS + (uint32_t)tid; i < glen; i += PL_STAGERS * PL_THREADS) {
                        unsigned long long x = sys_load(src + i);
                        for (uint32_t spins = 0; (x >> 32) != (gtag >> 32) && spins < PL_SPIN_LIMIT; ++spins) {
                            __builtin_amdgcn_s_sleep(1);
                            x = sys_load(src + i);
                        }
                        if ((x >> 32) != (gtag >> 32)) a.state[GS_ERR] = 2;   // the owner's row never arrived: the run is void
                        gran_store(st + i, (x >> 32) == (gtag >> 32) ? (ltag | (x & 0xffffffffull)) : (ltag | 0xffffffffull));
                    }
                stage_row = st;
            }
        }
        // ---- stop rules on the round's winner (search.py:15-37; the same values in every workgroup) ----
        last_key = top;
        if (top == 0 || qlen == 0 || qlen < thr || (top >> 32) < thr) break;
        const uint32_t wstart = s_wstart, wlen = s_wlen;
        // a position of the winning row: from this rank's own index, or from the staged copy of another rank's row
        auto row_pos = [&](uint32_t i) -> uint32_t {
            if (!stage_row) return a.qpos[(uint64_t)wstart + i];
            unsigned long long x = gran_load(stage_row + i);
            for (uint32_t spins = 0; (x >> 32) != epoch && spins < PL_SPIN_LIMIT; ++spins) {
                __builtin_amdgcn_s_sleep(1);
                x = gran_load(stage_row + i);
            }
            if ((x >> 32) != epoch) a.state[GS_ERR] = 3;                // the staged copy never arrived: the run is void
            return (x >> 32) == epoch ? (uint32_t)x : NONE32;
        };
        // ---- apply: I = row ∩ uncovered leaves the set; the owned rows holding a hash of I lose it ----
        uint32_t isect = 0;
        // a thread's positions of a chunk are asked for together, and the NEXT chunk's before this chunk's postings are walked
        uint32_t pos[PL_ROW_PER];
#pragma unroll
        for (int u = 0; u < PL_ROW_PER; ++u) {
            const uint32_t i = (uint32_t)u * PL_THREADS + (uint32_t)tid;
            pos[u] = i < wlen && i < a.chunk ? row_pos(i) : NONE32;
        }
        const uint32_t grp = (uint32_t)tid / PL_LANES, gl = (uint32_t)tid % PL_LANES;
        constexpr uint32_t GROUPS = PL_THREADS / PL_LANES;
        for (uint32_t c0 = 0; c0 < wlen; c0 += a.chunk) {
            if (tid == 0) s_nI = 0;
            __syncthreads();
            PL_WAITLAP(6);                                           // (trace) waiting for the row's positions
            {
                // all of a thread's test-and-clears are issued before the first result is used (the loop this replaces made
                // PL_ROW_PER rounds of: returning LDS atomic -> ballot -> returning LDS atomic on the counter -> store), and a
                // wave reserves room for ALL its hits of the chunk with one atomic
                uint32_t old_w[PL_ROW_PER];
#pragma unroll
                for (int u = 0; u < PL_ROW_PER; ++u) {
                    old_w[u] = 0;
                    if (pos[u] != NONE32) old_w[u] = atomicAnd(&s_bits[pos[u] >> 5], ~(1u << (pos[u] & 31u)));   // row hashes are distinct: no two lanes share a bit
                }
                unsigned long long m[PL_ROW_PER];
                uint32_t total = 0;
#pragma unroll
                for (int u = 0; u < PL_ROW_PER; ++u) {
                    m[u] = __ballot(pos[u] != NONE32 && ((old_w[u] >> (pos[u] & 31u)) & 1u) != 0);
                    total += (uint32_t)__popcll(m[u]);
                }
                uint32_t base = 0;
                if (lane == 0 && total) base = atomicAdd(&s_nI, total);
                base = __shfl(base, 0);
#pragma unroll
                for (int u = 0; u < PL_ROW_PER; ++u) {
                    if ((m[u] >> lane) & 1ull) s_I[base + (uint32_t)__popcll(m[u] & ((1ull << lane) - 1ull))] = pos[u];
                    base += (uint32_t)__popcll(m[u]);
                }
            }
            {
                const uint32_t n0 = c0 + a.chunk, n1 = n0 + a.chunk < wlen ? n0 + a.chunk : wlen;
#pragma unroll
                for (int u = 0; u < PL_ROW_PER; ++u) {
                    const uint32_t i = n0 + (uint32_t)u * PL_THREADS + (uint32_t)tid;
                    pos[u] = i < n1 ? row_pos(i) : NONE32;
                }
            }
            __syncthreads();
            PL_LAP(2);                                               // row scan: positions + bitmap
            const uint32_t nI = s_nI;
            isect += nI;
            if (n_own && nI) {
                // PL_LANES lanes per newly covered position: the run of this workgroup's row block in that posting list is
                // read as 16-byte pieces, two per lane.  PL_STEPS positions per lane group
                // form a batch whose loads are in flight together; the bounds of the next batch are asked for before this
                // batch's entries are used, so a further batch costs one trip to memory, not two.
                uint32_t lo[PL_STEPS], hi[PL_STEPS];
#pragma unroll
                for (int st = 0; st < PL_STEPS; ++st) {
                    const uint32_t k = (uint32_t)st * GROUPS + grp;
                    lo[st] = hi[st] = 0;
                    if (k < nI) {
                        const uint64_t p = s_I[k];
                        const uint32_t* bt = a.block_pre + p * (a.B + 1);             // one row of the table: lo and hi are neighbours
                        lo[st] = bt[b0];
                        hi[st] = bt[b1 + 1];
                    }
                }
                PL_WAITLAP(7);                                       // (trace) waiting for the first batch's run bounds
                for (uint32_t k0 = 0; k0 < nI; k0 += GROUPS * PL_STEPS) {
                    uint4 ent[PL_STEPS][2];
#pragma unroll
                    for (int st = 0; st < PL_STEPS; ++st)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const uint32_t idx = lo[st] + ((uint32_t)h * PL_LANES + gl) * 4u;
                            ent[st][h] = make_uint4(NONE32, NONE32, NONE32, NONE32);
                            if (idx < hi[st]) __builtin_memcpy(&ent[st][h], a.post_rows + idx, 16);   // 4-byte aligned: one dwordx4 load
                        }
                    uint32_t lo_n[PL_STEPS], hi_n[PL_STEPS];
#pragma unroll
                    for (int st = 0; st < PL_STEPS; ++st) {
                        const uint32_t k = k0 + GROUPS * PL_STEPS + (uint32_t)st * GROUPS + grp;
                        lo_n[st] = hi_n[st] = 0;
                        if (k < nI) {
                            const uint64_t p = s_I[k];
                            const uint32_t* bt = a.block_pre + p * (a.B + 1);             // one row of the table: lo and hi are neighbours
                            lo_n[st] = bt[b0];
                            hi_n[st] = bt[b1 + 1];
                        }
                    }
                    PL_WAITLAP(8);                                   // (trace) waiting for the batch's entries (+ next bounds)
#pragma unroll
                    for (int st = 0; st < PL_STEPS; ++st) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const uint32_t idx = lo[st] + ((uint32_t)h * PL_LANES + gl) * 4u;
                            const uint32_t r[4] = {ent[st][h].x, ent[st][h].y, ent[st][h].z, ent[st][h].w};
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (idx + (uint32_t)e < hi[st] && r[e] >= (uint32_t)r0 && r[e] < (uint32_t)r1)
                                    atomicSub(&s_cnt[r[e] - (uint32_t)r0], 1u);
                        }
                        // runs longer than 2 x 4 x PL_LANES entries (rare): the rest, one entry per lane and step
                        for (uint32_t idx = lo[st] + 8u * PL_LANES + gl; idx < hi[st]; idx += PL_LANES) {   // (8 = 2 pieces x 4 entries)
                            const uint32_t r = a.post_rows[idx];
                            if (r >= (uint32_t)r0 && r < (uint32_t)r1) atomicSub(&s_cnt[r - (uint32_t)r0], 1u);
                        }
                    }
#pragma unroll
                    for (int st = 0; st < PL_STEPS; ++st) { lo[st] = lo_n[st]; hi[st] = hi_n[st]; }
                }
            }
            __syncthreads();
            PL_LAP(3);                                               // run bounds + postings + LDS decrements
        }
        // ---- bookkeeping (pick_kernel's record_pending) ----
        if (wg == 0 && tid == 0) {
            a.out_idx[rounds] = 0xffffffffull & ~top;
            a.out_isect[rounds] = isect;
        }
        qlen -= isect;
        rounds += 1;
        if (rounds >= maxr) break;
    }
    // ---- hand the state back ----
    __syncthreads();
    if (timing) {
        a.dbg[7] = wall_clock64() - a.dbg[6];
        for (int i = 0; i < 6; ++i) a.dbg[i] = t_acc[i];
        for (int i = 6; i < 12; ++i) a.dbg[8 + i - 6] = t_acc[i];
    }
    for (uint32_t i = tid; i < n_own; i += PL_THREADS) a.counters[r0 + i] = s_cnt[i];
    {   // the alive bytes of this workgroup's slice of the query
        const uint64_t per = (a.nq + n_wg - 1) / n_wg;
        const uint64_t p_lo = (uint64_t)wg * per < a.nq ? (uint64_t)wg * per : a.nq, p_hi = p_lo + per < a.nq ? p_lo + per : a.nq;
        for (uint64_t p = p_lo + tid; p < p_hi; p += PL_THREADS) a.alive[p] = (uint8_t)((s_bits[p >> 5] >> (p & 31u)) & 1u);
    }
    if (wg == 0 && tid == 0) {
        a.state[GS_KEY] = last_key;
        a.state[GS_ROUNDS] = rounds;
        a.state[GS_QLEN] = qlen;
        a.state[GS_ACC] = 0;
        a.state[GS_PENDING] = 0;
        a.state[GS_DONE] = failed ? 0 : 1;        // given up inside a round (spin limits 11-13: each workgroup's own decision, so the
    }                                             // written-back state may mix round counts): the host treats the index as void
    if (failed && tid == 0 && a.state[GS_ERR] == 0) { a.state[GS_ERR] = fail_code; a.state[13] = fail_epoch; a.state[14] = wg; }
}

}  // namespace

void gather_destroy(GatherDev& g) {
    void* owned[] = {g.q_padded, g.q_table, g.q_rec, g.alive, g.post_off, g.post_rows, g.qpos, g.counters, g.state, g.partials, g.out_idx, g.out_isect,
                     g.topk_sel, g.topk_partials, g.cmask, g.cand_count, g.cand_key, g.cand_len, g.cand_qpos, g.own_cands, g.block_pre, g.loop_xchg};
    if (g.loop_stream) (void)hipStreamSynchronize(g.loop_stream);   // graph replays ran there; the blocks go back tagged with g.stream
    for (void* p : owned)
        if (p) arena_free(p, g.stream);
    if (g.pinned) arena_pinned_free(g.pinned);
    if (g.ev_build0) (void)hipEventDestroy(g.ev_build0);
    if (g.ev_build1) (void)hipEventDestroy(g.ev_build1);
    if (g.loop_graph) (void)hipGraphExecDestroy(g.loop_graph);
    if (g.loop_stream) (void)hipStreamDestroy(g.loop_stream);
    g = GatherDev();
}

// state block for a new loop: rounds restart at 0; the uncovered set, its size and the counters carry over
__global__ void gather_begin_kernel(unsigned long long* state, unsigned long long thr, unsigned long long maxr) {
    const int i = threadIdx.x;
    if (i >= GS_SLOTS) return;
    unsigned long long v = 0;
    if (i == GS_QLEN) v = state[GS_QLEN];
    else if (i == GS_THR) v = thr;
    else if (i == GS_MAXR) v = maxr;
    state[i] = v;
}

hipError_t gather_begin(GatherDev& g, uint64_t thr_hashes, uint64_t max_rounds, hipStream_t stream) {
    if (max_rounds == 0) max_rounds = 1;
    if (max_rounds > g.out_cap) {
        // a captured loop graph has the old result pointers baked into its nodes: it goes with them
        if (g.loop_graph) {
            if (g.loop_stream) SMG_TRY(hipStreamSynchronize(g.loop_stream));
            (void)hipGraphExecDestroy(g.loop_graph);
            g.loop_graph = nullptr;
        }
        if (g.out_idx) arena_free(g.out_idx, stream);
        if (g.out_isect) arena_free(g.out_isect, stream);
        g.out_idx = g.out_isect = nullptr;
        SMG_TRY(own_alloc(g, &g.out_idx, max_rounds * 8, stream));
        SMG_TRY(own_alloc(g, &g.out_isect, max_rounds * 8, stream));
        g.out_cap = max_rounds;
    }
    hipLaunchKernelGGL(gather_begin_kernel, dim3(1), dim3(64), 0, stream, g.state, (unsigned long long)thr_hashes,
                       (unsigned long long)max_rounds);
    return hipGetLastError();
}

hipError_t gather_pick(GatherDev& g, unsigned long long* d_key_out, int check_stop, hipStream_t stream) {
    const uint64_t want = (g.ndb + 1023) / 1024;
    const unsigned n_part = (unsigned)(want < 1 ? 1 : (want > GATHER_PICK_BLOCKS ? GATHER_PICK_BLOCKS : want));
    hipLaunchKernelGGL(pick_kernel, dim3(n_part), dim3(256), 0, stream, g.counters, g.ndb, g.index_base, g.state, g.partials,
                       g.out_idx, g.out_isect, d_key_out, check_stop);
    return hipGetLastError();
}

static CandView no_cands() { return CandView{nullptr, 0, nullptr, 0, nullptr, nullptr}; }

hipError_t gather_apply(GatherDev& g, hipStream_t stream) {
    hipLaunchKernelGGL(apply_kernel<true>, dim3(1024), dim3(256), 0, stream, qindex_of(g), g.alive, g.post_off,
                       g.post_rows, g.counters, g.state, (const uint64_t*)nullptr, g.hashes, g.offsets, g.index_base, g.qpos,
                       no_cands());
    return hipGetLastError();
}

hipError_t gather_consume_list(GatherDev& g, const uint64_t* d_list, hipStream_t stream) {
    hipLaunchKernelGGL(apply_kernel<false>, dim3(128), dim3(256), 0, stream, qindex_of(g), g.alive, g.post_off,
                       g.post_rows, g.counters, g.state, d_list, g.hashes, g.offsets, g.index_base, g.qpos, no_cands());
    return hipGetLastError();
}

// Geometry of the persistent loop for this index on `n_wg` workgroups; false: not applicable (the two-kernel rounds serve)
static bool loop_geometry(const GatherDev& g, uint32_t n_wg, LoopArgs* a, size_t* lds, size_t* budget_out) {
    static const bool off = [] { const char* e = getenv("SMG_GATHER_LOOP"); return e && strcmp(e, "persistent") != 0; }();
    if (off || !g.block_pre || g.counters_touched || g.ndb == 0 || g.nq == 0 || n_wg == 0) return false;
    if (g.pinned[0] >= 0xffffffffull) return false;                // pinned[0]: the shard's element count (offsets travel as 32 bits)
    const int lds_max = 160 * 1024;                                // gfx950: 160 KiB per CU, one workgroup per CU here
    a->rows_per_wg = (uint32_t)((g.ndb + (uint64_t)n_wg - 1) / (uint64_t)n_wg);
    a->block_rows = g.block_rows;
    a->B = g.block_B;
    a->bitmap_words = (uint32_t)((g.nq + 31) / 32);
    const size_t fixed = ((size_t)a->bitmap_words + 2 * (size_t)a->rows_per_wg + 1) * 4;
    const size_t budget = (size_t)lds_max - 1024;                  // the kernel's static scalars
    a->chunk = (uint32_t)PL_ROW_PER * PL_THREADS;
    while (a->chunk > 512 && fixed + (size_t)a->chunk * 4 > budget) a->chunk >>= 1;
    if (fixed + (size_t)a->chunk * 4 > budget) return false;       // query or rows too large for LDS
    *lds = fixed + (size_t)a->chunk * 4;
    *budget_out = budget;
    return true;
}

bool gather_loop_eligible(const GatherDev& g, uint32_t n_wg) {
    LoopArgs a;
    size_t lds = 0, budget = 0;
    if (n_wg == 0) {
        int dev = 0, n_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
        n_wg = (uint32_t)(n_cu > 0 ? n_cu : 0);
    }
    return loop_geometry(g, n_wg, &a, &lds, &budget);
}

// the loop's local exchange memory, (re)allocated only when the geometry grows.  A driver that launches several ranks' loops
// from ONE process calls this for all of them first: an allocation may synchronise the device, and a loop kernel that is
// already waiting for its peers would never see them start.
hipError_t gather_loop_reserve(GatherDev& g, hipStream_t stream, uint32_t n_wg, uint64_t rowcap) {
    if (n_wg == 0) {
        int dev = 0, n_cu = 0;
        SMG_TRY(hipGetDevice(&dev));
        SMG_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        n_wg = (uint32_t)n_cu;
    }
    const size_t words = (size_t)2 * n_wg * 4 + 16 + 8 + 8 + (size_t)2 * rowcap;
    if (!g.loop_xchg || g.loop_wgs != n_wg || g.loop_words < words) {
        if (g.loop_xchg) arena_free(g.loop_xchg, stream);
        g.loop_xchg = nullptr;
        SMG_TRY(own_alloc(g, &g.loop_xchg, words * 8, stream));
        g.loop_wgs = n_wg;
        g.loop_words = words;
    }
    return hipSuccess;
}

hipError_t gather_launch_loop(GatherDev& g, hipStream_t stream, uint32_t n_wg, const GatherShared* sh, bool* ran) {
    *ran = false;
    if (!g.out_idx) return hipSuccess;
    int dev = 0, n_cu = 0;
    SMG_TRY(hipGetDevice(&dev));
    SMG_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    if (n_cu <= 0) return hipSuccess;
    if (n_wg == 0 || n_wg > (uint32_t)n_cu) n_wg = (uint32_t)n_cu;
    // Workgroup b runs on XCD b % 8 and a workgroup fills a CU (16 waves of 111 VGPRs): a grid that is a multiple of 8 puts the
    // same number on every XCD.  (Three loops of 85 workgroups on one GPU put 33 on XCD 0, which has 32 CUs: one workgroup
    // never became resident and its peers waited for it until they gave up.)
    if (n_wg >= 8) n_wg -= n_wg % 8;
    LoopArgs a;
    size_t lds = 0, budget = 0;
    if (!loop_geometry(g, n_wg, &a, &lds, &budget)) return hipSuccess;
    a.offsets = g.offsets; a.qpos = g.qpos; a.post_rows = g.post_rows; a.block_pre = g.block_pre; a.post_off = g.post_off;
    a.counters = g.counters; a.alive = g.alive; a.state = g.state; a.out_idx = g.out_idx; a.out_isect = g.out_isect;
    a.ndb = g.ndb; a.nq = g.nq; a.index_base = g.index_base;
    // 1: set, -1: the runtime refused (no persistent loop); once, thread-safe (function-local static initialiser)
    static const int attr_state = [&] {
        const hipError_t e = hipFuncSetAttribute((const void*)gather_loop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget);
        if (e == hipSuccess) return 1;
        (void)hipGetLastError();
        return -1;
    }();
    if (attr_state < 0) return hipSuccess;
    // local exchange memory: [2][n_wg][4] workgroup records, 16 trace words, [2][4] global winner, 8 gate words, [2][rowcap] staged row
    const uint64_t rowcap = sh ? sh->rowcap : 0;
    if (sh && (sh->W == 0 || sh->W > (uint32_t)PL_THREADS || sh->rank >= sh->W || rowcap == 0)) return hipErrorInvalidValue;
    SMG_TRY(gather_loop_reserve(g, stream, n_wg, rowcap));
    a.xchg = g.loop_xchg;
    const uint32_t pf = 1u;            // the waves that do not sweep touch the own best row while the records travel (0: no touches, 13.3 us per round against 11.8)
    a.prefetch = pf;
    static const bool trace = getenv("SMG_GATHER_TRACE") != nullptr;
    a.dbg = trace ? g.loop_xchg + (size_t)2 * n_wg * 4 : nullptr;           // 16 words behind the granules
    a.gwin = g.loop_xchg + (size_t)2 * n_wg * 4 + 16;
    a.gate = a.gwin + 8;
    a.stage = a.gate + 8;
    // a grid on an idle device is resident within microseconds; 20 ms covers a launch queued behind somebody's kernel without
    // making the fallback wait long (SMG_GATHER_GATE_US: tests)
    static const unsigned long long gate_us = [] { const char* e = getenv("SMG_GATHER_GATE_US"); const long long v = e ? atoll(e) : 0; return v > 0 ? (unsigned long long)v : 20000ull; }();
    a.gate_ticks = gate_us * 100ull;
    a.W = sh ? sh->W : 0; a.rank = sh ? sh->rank : 0; a.rowcap = (uint32_t)rowcap;
    a.epoch_base = sh ? (sh->run_id & 0xfffu) << 20 : 0;                    // tags of the shared slots: unique per run, no zeroing between runs
    a.x_rec = sh ? sh->rec : nullptr; a.x_rows = sh ? sh->rows : nullptr;
    a.peers = 0;
    for (uint32_t r = 0; r < GATHER_PEERS_MAX; ++r) a.peer[r] = nullptr;
    if (sh && sh->peers) {
        if (sh->W > GATHER_PEERS_MAX) return hipErrorInvalidValue;
        a.peers = 1;
        for (uint32_t r = 0; r < sh->W; ++r) a.peer[r] = sh->peers[r];
    }
    SMG_TRY(hipMemsetAsync(g.loop_xchg, 0, ((size_t)2 * n_wg * 4 + 16 + 8 + 8) * 8, stream));   // local epochs count from 1 within a launch
    if (rowcap) SMG_TRY(hipMemsetAsync(a.stage, 0, (size_t)2 * rowcap * 8, stream));
    // One workgroup per CU: all of them must be resident at once (the sweeps wait for every workgroup).  A plain launch has
    // the same residency as a cooperative one (MI355X_MICROARCH.md) without its 15-19 us and without the cooperative
    // interception that crashes rocprofv3 here; the occupancy query is the check the cooperative launch would make, and a
    // workgroup that waits too long for a peer gives up (GS_ERR) instead of hanging.
    {
        static int blocks_per_cu = -1;
        if (blocks_per_cu < 0) {
            int nb = 0;
            const hipError_t eo = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)gather_loop_kernel, PL_THREADS, budget);
            blocks_per_cu = eo == hipSuccess ? nb : 0;
            if (eo != hipSuccess) (void)hipGetLastError();
        }
        if (blocks_per_cu < 1) return hipSuccess;
    }
    hipLaunchKernelGGL(gather_loop_kernel, dim3(n_wg), dim3(PL_THREADS), lds, stream, a);
    SMG_TRY(hipGetLastError());
    *ran = true;
    if (trace && !sh) {
        unsigned long long d[16];
        SMG_TRY(hipMemcpyAsync(d, a.dbg, sizeof(d), hipMemcpyDeviceToHost, stream));
        SMG_TRY(hipStreamSynchronize(stream));
        fprintf(stderr, "[gather] persistent loop, workgroup 0 (us): argmax+publish %.1f, sweep %.1f (%llu empty sweeps), row scan %.1f, "
                        "postings %.1f, whole loop %.1f; lds %zu B, chunk %u, rows/wg %u, row blocks %u\n",
                d[0] * 0.01, d[1] * 0.01, d[5], d[2] * 0.01, d[3] * 0.01, d[7] * 0.01, lds, a.chunk, a.rows_per_wg, a.B);
        fprintf(stderr, "[gather]   of which waiting for: row positions %.1f, first run bounds %.1f, entries %.1f\n", d[8] * 0.01, d[9] * 0.01,
                d[10] * 0.01);
    }
    return hipSuccess;
}

// ---- test support: hold CUs ----------------------------------------------------------------------------------------------
// n_wg workgroups that each keep `lds_bytes` of LDS and spin for `micros` microseconds: a stand-in for "somebody else's kernel is
// running on this device" (tests/test_gpu_gather.py: the resident loop must step aside, not fail).
__global__ __launch_bounds__(256) void hold_cus_kernel(unsigned long long ticks, uint32_t* sink) {
    extern __shared__ uint32_t hold_lds[];
    hold_lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    uint32_t x = hold_lds[(threadIdx.x * 7u) & 255u];
    while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(32); x += 1; }
    if (x == 0xdeadbeefu) *sink = x;                     // (keeps the loop and the LDS alive)
}
hipError_t debug_hold_cus(uint32_t n_wg, uint32_t lds_bytes, uint64_t micros, hipStream_t stream) {
    static uint32_t* sink = nullptr;
    if (!sink) SMG_TRY(hipMalloc(&sink, 256));
    if (lds_bytes < 1024) lds_bytes = 1024;
    SMG_TRY(hipFuncSetAttribute((const void*)hold_cus_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(hold_cus_kernel, dim3(n_wg), dim3(256), lds_bytes, stream, (unsigned long long)micros * 100ull, sink);
    return hipGetLastError();
}

hipError_t gather_run_persistent(GatherDev& g, hipStream_t stream, bool* ran) { return gather_launch_loop(g, stream, 0, nullptr, ran); }

hipError_t gather_enqueue_rounds(GatherDev& g, unsigned rounds, hipStream_t stream) {
    for (unsigned r = 0; r < rounds; ++r) {
        SMG_TRY(gather_pick(g, nullptr, 1, stream));
        SMG_TRY(gather_apply(g, stream));
    }
    return hipSuccess;
}

hipError_t gather_enqueue_rounds_graph(GatherDev& g, unsigned rounds, hipStream_t* used) {
    // Runs on the index's own stream: a caller's stream may be the legacy default stream, which cannot capture.  The
    // caller has synchronised its stream before the first call and synchronises *used after every batch.
    if (!g.loop_stream) SMG_TRY(hipStreamCreateWithFlags(&g.loop_stream, hipStreamNonBlocking));
    if (!g.loop_graph) {
        hipGraph_t graph = nullptr;
        SMG_TRY(hipStreamBeginCapture(g.loop_stream, hipStreamCaptureModeThreadLocal));
        const hipError_t e = gather_enqueue_rounds(g, GATHER_GRAPH_ROUNDS, g.loop_stream);
        const hipError_t e2 = hipStreamEndCapture(g.loop_stream, &graph);
        if (e != hipSuccess) { if (graph) (void)hipGraphDestroy(graph); return e; }
        SMG_TRY(e2);
        const hipError_t e3 = hipGraphInstantiate(&g.loop_graph, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        SMG_TRY(e3);
    }
    for (unsigned r = 0; r < rounds; r += GATHER_GRAPH_ROUNDS) SMG_TRY(hipGraphLaunch(g.loop_graph, g.loop_stream));
    *used = g.loop_stream;
    return hipSuccess;
}

// ---- candidate replay --------------------------------------------------------------------------------------------
static hipError_t replay_buffers(GatherDev& g, hipStream_t stream) {
    if (g.cmask) return hipSuccess;
    SMG_TRY(own_alloc(g, &g.topk_sel, (GATHER_TOPK_MAX + 1) * 8, stream));
    SMG_TRY(own_alloc(g, &g.topk_partials, (size_t)GATHER_PICK_BLOCKS * (GATHER_TOPK_MAX + 1) * 8, stream));
    SMG_TRY(own_alloc(g, &g.cmask, (g.nq + 1) * 8, stream));
    SMG_TRY(own_alloc(g, &g.cand_count, GATHER_CAND_MAX * 8, stream));
    SMG_TRY(own_alloc(g, &g.cand_key, GATHER_CAND_MAX * 8, stream));
    SMG_TRY(own_alloc(g, &g.cand_len, GATHER_CAND_MAX * 4, stream));
    SMG_TRY(hipMemsetAsync(g.cmask, 0, (g.nq + 1) * 8, stream));
    SMG_TRY(hipMemsetAsync(g.cand_count, 0, GATHER_CAND_MAX * 8, stream));
    SMG_TRY(hipMemsetAsync(g.cand_key, 0, GATHER_CAND_MAX * 8, stream));
    SMG_TRY(hipMemsetAsync(g.cand_len, 0, GATHER_CAND_MAX * 4, stream));
    return hipSuccess;
}

hipError_t gather_topk_export(GatherDev& g, uint64_t* d_out, uint32_t K, uint64_t stride, hipStream_t stream) {
    if (K == 0 || K > GATHER_TOPK_MAX || stride < GATHER_CAND_HEAD + 1) return hipErrorInvalidValue;
    SMG_TRY(replay_buffers(g, stream));
    const uint64_t want = (g.ndb + 1023) / 1024;
    const unsigned n_part = (unsigned)(want < 1 ? 1 : (want > GATHER_PICK_BLOCKS ? GATHER_PICK_BLOCKS : want));
    hipLaunchKernelGGL(topk_kernel, dim3(n_part), dim3(256), 0, stream, g.counters, g.ndb, g.index_base, g.state,
                       g.topk_partials, g.topk_sel, (int)K + 1);
    const uint64_t per_row = (stride + 255) / 256;
    hipLaunchKernelGGL(export_cands_kernel, dim3((unsigned)(per_row > 64 ? 64 : per_row), K), dim3(256), 0, stream, g.state,
                       g.topk_sel, K, g.hashes, g.offsets, g.ndb, g.index_base, d_out, stride);
    return hipGetLastError();
}

hipError_t gather_cands_load(GatherDev& g, const uint64_t* d_cands, uint32_t n_cand, uint64_t stride, hipStream_t stream) {
    if (n_cand == 0 || n_cand > GATHER_CAND_MAX || stride < GATHER_CAND_HEAD + 1) return hipErrorInvalidValue;
    SMG_TRY(replay_buffers(g, stream));
    if (g.cand_qstride < stride) {                               // first load, or longer records than before
        if (g.cand_qpos) {
            SMG_TRY(hipStreamSynchronize(stream));
            // the old positions are about to go: clear their bits now
            hipLaunchKernelGGL(cands_clear_kernel, dim3(GATHER_CAND_MAX), dim3(256), 0, stream, g.cmask, g.cand_len, g.cand_qpos,
                               g.cand_qstride);
            SMG_TRY(hipMemsetAsync(g.cand_len, 0, GATHER_CAND_MAX * 4, stream));
            arena_free(g.cand_qpos, stream);
            g.cand_qpos = nullptr;
        }
        SMG_TRY(own_alloc(g, &g.cand_qpos, (size_t)GATHER_CAND_MAX * stride * 4, stream));
        g.cand_qstride = stride;
    }
    hipLaunchKernelGGL(cands_clear_kernel, dim3(GATHER_CAND_MAX), dim3(256), 0, stream, g.cmask, g.cand_len, g.cand_qpos,
                       g.cand_qstride);
    if (n_cand < GATHER_CAND_MAX)                                // slots beyond this exchange hold nothing
        SMG_TRY(hipMemsetAsync(g.cand_len + n_cand, 0, (GATHER_CAND_MAX - n_cand) * 4, stream));
    hipLaunchKernelGGL(cands_load_kernel, dim3(n_cand), dim3(256), 0, stream, qindex_of(g), g.alive, d_cands, n_cand, stride,
                       g.cmask, g.cand_count, g.cand_key, g.cand_len, g.cand_qpos, g.cand_qstride, g.state);
    g.cands = d_cands;
    g.cand_stride = stride;
    g.n_cand = n_cand;
    return hipGetLastError();
}

hipError_t gather_replay_rounds(GatherDev& g, unsigned rounds, hipStream_t stream) {
    if (!g.cands) return hipErrorInvalidValue;
    const CandView cv{g.cands, g.cand_stride, g.cand_qpos, g.cand_qstride, g.cmask, g.cand_count};
    for (unsigned r = 0; r < rounds; ++r) {
        hipLaunchKernelGGL(replay_pick_kernel, dim3(1), dim3(64), 0, stream, g.cand_count, g.cand_key, g.n_cand, g.state,
                           g.out_idx, g.out_isect);
        hipLaunchKernelGGL(apply_kernel<true>, dim3(1024), dim3(256), 0, stream, qindex_of(g), g.alive, g.post_off,
                           g.post_rows, g.counters, g.state, (const uint64_t*)nullptr, g.hashes, g.offsets, g.index_base,
                           g.qpos, cv);
    }
    return hipGetLastError();
}

hipError_t gather_enqueue_replay(GatherDev& g, unsigned exchanges, hipStream_t stream) {
    // single shard: the K + 1 best rows by key are exactly the next candidates, so K rounds per exchange can succeed
    const uint32_t K = GATHER_TOPK_MAX;
    if (!g.own_cands) {
        g.own_cands_words = (GATHER_CAND_HEAD + (g.longest_row ? g.longest_row : 1)) * K;
        SMG_TRY(own_alloc(g, &g.own_cands, g.own_cands_words * 8, stream));
    }
    const uint64_t stride = g.own_cands_words / K;
    for (unsigned x = 0; x < exchanges; ++x) {
        SMG_TRY(gather_topk_export(g, g.own_cands, K, stride, stream));
        SMG_TRY(gather_cands_load(g, g.own_cands, K, stride, stream));
        SMG_TRY(gather_replay_rounds(g, K, stream));
    }
    return hipSuccess;
}

}  // namespace smg
