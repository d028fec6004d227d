// gather_build.hip -- the gather index: the database inverted against the query once (round 6: split from gather.hip).
//
// What the reference does: CounterGather.add walks the query against every dataset as it is inserted and keeps the
// intersections (src/sourmash/index/__init__.py:783-789); every later round walks them again (:897-909).  Here the shard is
// inverted ONCE: post_rows[post_off[j] .. post_off[j+1]) = the rows holding query hash j, counters[d] = |Q ∩ D_d|, and qpos[e] =
// the query position of every database hash (or NONE32), which the rounds of gather.hip read instead of looking hashes up again.
//   small databases        build_count_kernel / build_fill_kernel: a lookup per element, atomic cursors
//   range builder          build_bounds_kernel, build_range_kernel (pass 1 by lookups in L2), build_partition_kernel (pass 2a),
//                          build_scatter_kernel (pass 2b: a window's postings sorted by (list, row block) in LDS)
//   staged range builder   pass 1 + 2a by the lean streaming kernel of overlap.hip (build_stage_launch), then
//                          build_count_runs_kernel, build_merge_counts_kernel, build_scatter_kernel, build_bounds_table_kernel
// Every block comes from the arena: a warm build makes no driver call.  DESIGN.md 4.5; measurements in profiles/HISTORY.md 4.5.
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

__global__ __launch_bounds__(256) void qtable_kernel(const uint64_t* __restrict__ Q, uint64_t nq, uint32_t shift,
                                                     uint32_t n_buckets, uint32_t* __restrict__ T) {
    qindex_fill_bucket(Q, nq, shift, n_buckets, T, blockIdx.x * blockDim.x + threadIdx.x);
}

__global__ __launch_bounds__(256) void qrec_kernel(const uint64_t* __restrict__ Q, const uint32_t* __restrict__ T,
                                                   uint32_t n_buckets, QRec* __restrict__ rec) {
    qindex_fill_record(Q, T, n_buckets, rec, blockIdx.x * blockDim.x + threadIdx.x);
}

// pass 1 over the database: qpos of every element, postings histogram, initial counters (CounterGather.add)
__global__ __launch_bounds__(256) void build_count_kernel(QIndex qi, const uint64_t* __restrict__ hashes,
                                                          const uint64_t* __restrict__ offsets, uint64_t ndb,
                                                          uint32_t* __restrict__ qpos, unsigned long long* post_cnt,
                                                          unsigned long long* __restrict__ counters) {
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    for (uint64_t d = wave; d < ndb; d += n_waves) {
        const uint64_t lo = offsets[d], hi = offsets[d + 1];
        unsigned long long cnt = 0;
        for (uint64_t i = lo + lane; i < hi; i += 64) {
            const uint32_t j = q_find(qi, hashes[i]);
            qpos[i] = j;
            if (j != NONE32) {
                atomicAdd(&post_cnt[j], 1ull);
                ++cnt;
            }
        }
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
        if (lane == 0) counters[d] = cnt;
    }
}

// pass 2: scatter the row ids into the postings
__global__ __launch_bounds__(256) void build_fill_kernel(const uint32_t* __restrict__ qpos,
                                                         const uint64_t* __restrict__ offsets, uint64_t ndb,
                                                         unsigned long long* cursor, uint32_t* __restrict__ post_rows) {
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    for (uint64_t d = wave; d < ndb; d += n_waves) {
        const uint64_t lo = offsets[d], hi = offsets[d + 1];
        for (uint64_t i = lo + lane; i < hi; i += 64) {
            const uint32_t j = qpos[i];
            if (j != NONE32) post_rows[atomicAdd(&cursor[j], 1ull)] = (uint32_t)d;
        }
    }
}

// ---- range-partitioned build (large databases) ------------------------------------------------------------------
// The two kernels above spend their time in device-scope atomics (one per database element that hits the query, twice).
// Here the query positions are cut into ranges of BR_RANGE and the rows into B blocks; workgroup (range r, block b)
// walks the slices of its rows that fall into the range -- contiguous, because rows are sorted -- and keeps the
// histogram / the cursors of those BR_RANGE postings in LDS.  Global atomics: one per (row, range) for the counters.
//   bounds   [R + 1][ndb]  first position of row d whose hash is >= Q[r * BR_RANGE]  (row length for r = R)
//   partial  [B][nq]       pass 1: postings of query hash j contributed by block b; then its exclusive prefix over b
// so that in pass 2 the slot of an element is post_off[j] + partial[b][j] + (LDS cursor of j in this workgroup).
constexpr int BR_RANGE = 32768;      // query positions per range: row slices long enough (~1 KB) to read DRAM efficiently
constexpr int BR_EPW = 16;           // row slices a wave flattens per step (see apply_kernel)
constexpr int BR_AHEAD = 4;   // steps of 64 lookups a wave keeps in flight in pass 1 (see build_range_kernel)
constexpr int BR_THREADS = 512;      // 64 KB of LDS per workgroup (u16 slots, two per word): 2 workgroups = 16 waves per CU
// Two-level fill (the default): a 4-byte store per posting straight into its list leaves 32,768 lists x 8 lines open per
// range -- far more than one L2 -- and partially filled lines were evicted and fetched back (8.1 GB written and 13.6 GB
// read for 1 GB of postings, profiles/r01_gather_pmc.txt).  Instead:
//   pass 2a  workgroup (range, row block) re-partitions its postings by sub-range of BR_SUB lists into an intermediate
//            buffer laid out [range][sub-range][row block] (exact sizes from pass 1).  A workgroup appends to BR_NSUB
//            streams, i.e. it has that many lines open: the L2 write-combines them (64 resident workgroups x 16 KB).
//   pass 2b  a window of BR_SUB lists goes to BR_GROUPS workgroups on ONE XCD; each counting-sorts the entries of its
//            row blocks by list in LDS and writes every list's run (about a line) with consecutive lanes.
// (BR_SUB = 256 lists per window of the final scatter, BR_SUB_BITS: gather_parts.hpp -- the staging kernel of overlap.hip packs by them)
constexpr int BR_NSUB = BR_RANGE / BR_SUB;        // sub-ranges per range
constexpr int BR_ROWBITS = 32 - BR_SUB_BITS;      // entry = (row << 8) | list within the window
constexpr int BR_GROUPS = 8;                      // workgroups per window in pass 2b (each takes B / 8 row blocks)
constexpr int BR_SORT_CAP = 12288;                // entries a pass-2b workgroup sorts at a time (48 KB of LDS)
static_assert((1 << BR_SUB_BITS) == BR_SUB, "");
constexpr int MS_BMAX = 64;                      // row blocks the staged builder's directory and scatter are laid out for

__global__ __launch_bounds__(256) void build_bounds_kernel(const uint64_t* __restrict__ Q, uint32_t R,
                                                           const uint64_t* __restrict__ hashes,
                                                           const uint64_t* __restrict__ offsets, uint64_t ndb,
                                                           uint32_t* __restrict__ bounds) {
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    for (uint64_t d = wave; d < ndb; d += n_waves) {
        const uint64_t base = offsets[d], len = offsets[d + 1] - base;
        for (uint32_t r = lane; r <= R; r += 64) {
            uint64_t lo = 0, hi = len;
            if (r < R) {
                const uint64_t x = Q[(uint64_t)r * BR_RANGE];
                while (lo < hi) {
                    const uint64_t mid = (lo + hi) >> 1;
                    if (hashes[base + mid] < x) lo = mid + 1; else hi = mid;
                }
            } else {
                lo = len;
            }
            bounds[(uint64_t)r * ndb + d] = (uint32_t)lo;
        }
    }
}

// MODE 0: pass 1 (lookups, query positions, histogram, counters)   MODE 1: direct fill (one store per posting into its list)
// (Rounds 1-4 also had a MODE 2 -- pass 2a by plain 4-byte stores into 128 open streams -- and a MODE 3 -- the overlap pass by
//  lookups in L2; build_partition_kernel and the streaming kernels of overlap.hip replaced them and they are gone.)
template <int MODE>
__global__ __launch_bounds__(BR_THREADS) void build_range_kernel(QIndex qi, const uint64_t* __restrict__ hashes,
                                                          const uint64_t* __restrict__ offsets, uint64_t ndb,
                                                          const uint32_t* __restrict__ bounds, uint32_t R, uint32_t B,
                                                          uint64_t rows_per_block, uint32_t* __restrict__ partial,
                                                          const uint64_t* __restrict__ post_off,
                                                          uint32_t* __restrict__ post_rows, unsigned long long* counters,
                                                          uint32_t* __restrict__ qpos, uint32_t* __restrict__ subcnt) {
    static_assert(MODE == 0 || MODE == 1, "");
    constexpr bool FILL = MODE == 1;
    constexpr bool COUNT = MODE == 0;
    // pass 1: histogram; direct fill: cursors.  A block holds < 65536 rows and a row adds at most 1 to a slot, so 16 bits do.
    __shared__ uint32_t s_slot[BR_RANGE / 2];
    // Launch order: ranges in groups of 8, range (8g + x) entirely on workgroup ids = x mod 8, i.e. on one XCD (workgroups
    // are dealt to the 8 XCDs round-robin), blocks in ascending order.  The 4-byte stores of pass 2 that fill one posting
    // list then meet in a single L2, whose working set is one open cache line per list of the range.
    const uint32_t local = blockIdx.x % (8u * B);
    const uint32_t r = (blockIdx.x / (8u * B)) * 8u + (local & 7u), b = local >> 3;
    if (r >= R) return;
    const uint64_t j0 = (uint64_t)r * BR_RANGE;
    const uint32_t nj = (uint32_t)(qi.nq - j0 < (uint64_t)BR_RANGE ? qi.nq - j0 : (uint64_t)BR_RANGE);
    for (int k = threadIdx.x; k < BR_RANGE / 2; k += BR_THREADS) s_slot[k] = 0;
    __syncthreads();
    const uint64_t d_lo = (uint64_t)b * rows_per_block;
    const uint64_t d_hi = d_lo + rows_per_block < ndb ? d_lo + rows_per_block : ndb;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t* part_b = partial + (uint64_t)b * qi.nq;
    for (uint64_t dbase = d_lo + (uint64_t)wave * BR_EPW; dbase < d_hi; dbase += (BR_THREADS / 64) * BR_EPW) {
        // lane l < BR_EPW: the slice of row dbase + l inside this range
        uint64_t lo = 0;
        uint32_t n = 0;
        const uint64_t d = dbase + lane;
        if (lane < BR_EPW && d < d_hi) {
            // pass 1 starts its first range at the row's first element: what lies below Q[0] is looked up (and missed) like
            // everything else, so that every element's query position is written and nobody has to pre-fill 2 GB of them
            const uint32_t a = (MODE == 0 && r == 0) ? 0u : bounds[(uint64_t)r * ndb + d], e = bounds[(uint64_t)(r + 1) * ndb + d];
            lo = offsets[d] + a;
            n = e - a;
        }
        uint32_t incl = n;
#pragma unroll
        for (int s = 1; s < BR_EPW; s <<= 1) {
            const uint32_t v = __shfl_up(incl, s);
            if (lane >= s) incl += v;
        }
        const uint32_t total = __shfl(incl, BR_EPW - 1);
        if (total == 0) continue;
        uint32_t bound[BR_EPW - 1];                                 // wave-uniform: end of slices 0 .. 14
#pragma unroll
        for (int k = 0; k < BR_EPW - 1; ++k) bound[k] = __shfl(incl, k);
        const uint32_t excl = incl - n;
        const uint32_t lo_lo = (uint32_t)lo, lo_hi = (uint32_t)(lo >> 32);
        uint32_t row_hits = 0;                                      // lane l < BR_EPW: hits of slice l
        if (COUNT && qi.rec) {
            // Lookups, BR_AHEAD steps of 64 elements at a time: a step is a chain of two loads (the hash from HBM, its
            // record from L2, about 3 us together) and a wave that waits for each step in turn keeps 64 loads in flight --
            // 16 waves per CU then bound the pass at ~100 G lookups/s whatever the bandwidth (6.0 ms at C5, measured).
            // All hash loads of the steps are issued first, then all record loads, then the compares and outputs.
            for (uint32_t t0 = 0; t0 < total; t0 += 64 * BR_AHEAD) {
                uint64_t x[BR_AHEAD], at[BR_AHEAD];
                bool ok[BR_AHEAD];
#pragma unroll
                for (int u = 0; u < BR_AHEAD; ++u) {
                    const uint32_t t = t0 + 64u * (uint32_t)u + (uint32_t)lane;
                    ok[u] = t < total;
                    const uint32_t tt = ok[u] ? t : total - 1;      // lanes past the end re-read the last element
                    int h = 0;
#pragma unroll
                    for (int k = 0; k < BR_EPW - 1; ++k) h += tt >= bound[k];
                    const uint64_t start = ((uint64_t)(uint32_t)__shfl((int)lo_hi, h) << 32) | (uint32_t)__shfl((int)lo_lo, h);
                    at[u] = start + (tt - (uint32_t)__shfl((int)excl, h));
                    x[u] = hashes[at[u]];
                }
                __builtin_amdgcn_sched_barrier(0);                  // every hash load is out before the first one is waited for
                QRecVal rv[BR_AHEAD];
#pragma unroll
                for (int u = 0; u < BR_AHEAD; ++u) rv[u] = q_rec_load(qi, x[u] <= qi.qmax ? x[u] : qi.qmax);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < BR_AHEAD; ++u) {
                    uint32_t j = q_rec_match(qi, x[u], rv[u]);
                    if (!ok[u] || x[u] > qi.qmax) j = NONE32;
                    if (MODE == 0 && ok[u]) qpos[at[u]] = j;
                    const bool hit = j != NONE32;
                    if (MODE == 0 && hit) {
                        const uint32_t k = j - (uint32_t)j0;        // < nj: the slice lies inside the range
                        atomicAdd(&s_slot[k >> 1], 1u << (16u * (k & 1u)));
                    }
                    const uint32_t tb = t0 + 64u * (uint32_t)u;
                    const unsigned long long hits = __ballot(hit);
                    const uint32_t a = excl > tb ? (excl - tb < 64u ? excl - tb : 64u) : 0u;
                    const uint32_t e = incl > tb ? (incl - tb < 64u ? incl - tb : 64u) : 0u;
                    const unsigned long long upto_e = e >= 64u ? ~0ull : ((1ull << e) - 1ull);
                    const unsigned long long upto_a = a >= 64u ? ~0ull : ((1ull << a) - 1ull);
                    row_hits += (uint32_t)__popcll(hits & upto_e & ~upto_a);
                }
            }
        } else
        for (uint32_t t0 = 0; t0 < total; t0 += 64) {               // wave-uniform trip count: the shuffles read lanes 0 .. 15
            const uint32_t t = t0 + (uint32_t)lane;
            int h = 0;
#pragma unroll
            for (int k = 0; k < BR_EPW - 1; ++k) h += t >= bound[k];
            const uint64_t start = ((uint64_t)(uint32_t)__shfl((int)lo_hi, h) << 32) | (uint32_t)__shfl((int)lo_lo, h);
            const uint32_t first = (uint32_t)__shfl((int)excl, h);
            uint32_t j = NONE32;
            if (t < total) {
                if (FILL) {
                    j = qpos[start + (t - first)];                  // pass 1 left it there
                } else {
                    j = q_find(qi, hashes[start + (t - first)]);
                    if (MODE == 0) qpos[start + (t - first)] = j;
                }
            }
            const bool hit = j != NONE32;
            if (hit) {
                const uint32_t k = j - (uint32_t)j0;                // < nj: the slice lies inside the range
                const uint32_t sh = 16u * (k & 1u);
                if (FILL) {
                    const uint32_t mine = (atomicAdd(&s_slot[k >> 1], 1u << sh) >> sh) & 0xffffu;
                    const uint64_t at = post_off ? post_off[j] + part_b[j] : (uint64_t)part_b[j];   // null: partial already holds absolute slots
                    post_rows[at + mine] = (uint32_t)(dbase + (uint64_t)h);
                } else if (MODE == 0) {
                    atomicAdd(&s_slot[k >> 1], 1u << sh);
                }
            }
            if (COUNT) {
                // slice l occupies the flattened positions [excl, incl): its lanes in this step are a contiguous run
                const unsigned long long hits = __ballot(hit);
                const uint32_t a = excl > t0 ? (excl - t0 < 64u ? excl - t0 : 64u) : 0u;
                const uint32_t e = incl > t0 ? (incl - t0 < 64u ? incl - t0 : 64u) : 0u;
                const unsigned long long upto_e = e >= 64u ? ~0ull : ((1ull << e) - 1ull);
                const unsigned long long upto_a = a >= 64u ? ~0ull : ((1ull << a) - 1ull);
                row_hits += (uint32_t)__popcll(hits & upto_e & ~upto_a);
            }
        }
        if (COUNT && lane < BR_EPW && row_hits) atomicAdd(&counters[d], (unsigned long long)row_hits);
    }
    if (FILL) return;
    __syncthreads();
    uint32_t* out = partial + (uint64_t)b * qi.nq + j0;
    for (uint32_t k = threadIdx.x; k < nj; k += BR_THREADS) out[k] = (s_slot[k >> 1] >> (16u * (k & 1u))) & 0xffffu;
    if (subcnt) {
        // postings this workgroup holds per sub-range: PARTS threads per sub-range, each sums its share of the words
        constexpr int PARTS = BR_THREADS / BR_NSUB, WORDS = BR_SUB / 2 / PARTS;
        static_assert(BR_THREADS % BR_NSUB == 0 && (BR_SUB / 2) % PARTS == 0 && PARTS <= 64 && (PARTS & (PARTS - 1)) == 0, "");
        const int sub = threadIdx.x / PARTS, part = threadIdx.x % PARTS;
        uint32_t sum = 0;
        for (int w = 0; w < WORDS; ++w) {
            const uint32_t v = s_slot[sub * (BR_SUB / 2) + part * WORDS + w];
            sum += (v & 0xffffu) + (v >> 16);
        }
        for (int off = PARTS / 2; off > 0; off >>= 1) sum += __shfl_down(sum, off, PARTS);
        if (part == 0) subcnt[((uint64_t)r * BR_NSUB + sub) * B + b] = sum;
    }
}

// pass 2a with staging: the walk of build_range_kernel over the positions pass 1 left, but the workgroup moves through its rows in
// chunks of 8 waves x 16 rows with all waves in step; an entry's final slot in its stream is reserved at once (an LDS
// counter per stream), the entry itself waits in LDS -- stream k's entries of the chunk at s_stage[k][slot - chunk start]
// -- and after the chunk every stream's part goes out with consecutive lanes (~80 entries = 330 bytes at C5).  An entry
// that does not fit its stream's LDS row is stored directly at its slot.  What reaches the L2 are runs, not 4-byte
// stores scattered over 128 open lines.
constexpr int PA_AHEAD = 4;                      // steps of 64 position loads a wave keeps in flight
constexpr int PA_CAPS = 128;                      // staged entries per stream and chunk: 128 x 128 x 4 B = 64 KB of LDS

__global__ __launch_bounds__(BR_THREADS) void build_partition_kernel(uint64_t nq, const uint64_t* __restrict__ offsets,
                                                                     uint64_t ndb, const uint32_t* __restrict__ bounds,
                                                                     uint32_t R, uint32_t B, uint64_t rows_per_block,
                                                                     const uint32_t* __restrict__ qpos,
                                                                     const uint32_t* __restrict__ inter_off,
                                                                     uint32_t* __restrict__ inter) {
    constexpr int WAVES = BR_THREADS / 64;
    __shared__ uint32_t s_stage[BR_NSUB][PA_CAPS];
    __shared__ uint32_t s_gbase[BR_NSUB], s_gcur[BR_NSUB], s_cbase[BR_NSUB];
    const uint32_t local = blockIdx.x % (8u * B);
    const uint32_t r = (blockIdx.x / (8u * B)) * 8u + (local & 7u), b = local >> 3;
    if (r >= R) return;
    const uint64_t j0 = (uint64_t)r * BR_RANGE;
    (void)nq;
    for (int k = threadIdx.x; k < BR_NSUB; k += BR_THREADS) {
        s_gbase[k] = inter_off[((uint64_t)r * BR_NSUB + k) * B + b];
        s_gcur[k] = 0;
    }
    const uint64_t d_lo = (uint64_t)b * rows_per_block;
    const uint64_t d_hi = d_lo + rows_per_block < ndb ? d_lo + rows_per_block : ndb;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t n_chunks = d_hi > d_lo ? (d_hi - d_lo + (uint64_t)(WAVES * BR_EPW) - 1) / (WAVES * BR_EPW) : 0;
    for (uint64_t chunk = 0; chunk < n_chunks; ++chunk) {
        __syncthreads();                                            // the previous chunk is flushed (and, first time, s_gbase is set)
        for (int k = threadIdx.x; k < BR_NSUB; k += BR_THREADS) s_cbase[k] = s_gcur[k];
        __syncthreads();
        const uint64_t dbase = d_lo + (chunk * WAVES + (uint64_t)wave) * BR_EPW;
        if (dbase < d_hi) {
            uint64_t lo = 0;
            uint32_t n = 0;
            const uint64_t d = dbase + lane;
            if (lane < BR_EPW && d < d_hi) {
                const uint32_t a = bounds[(uint64_t)r * ndb + d], e = bounds[(uint64_t)(r + 1) * ndb + d];
                lo = offsets[d] + a;
                n = e - a;
            }
            uint32_t incl = n;
#pragma unroll
            for (int sft = 1; sft < BR_EPW; sft <<= 1) {
                const uint32_t v = __shfl_up(incl, sft);
                if (lane >= sft) incl += v;
            }
            const uint32_t total = __shfl(incl, BR_EPW - 1);
            uint32_t bound[BR_EPW - 1];
#pragma unroll
            for (int k = 0; k < BR_EPW - 1; ++k) bound[k] = __shfl(incl, k);
            const uint32_t excl = incl - n;
            const uint32_t lo_lo = (uint32_t)lo, lo_hi = (uint32_t)(lo >> 32);
            for (uint32_t t0 = 0; t0 < total; t0 += 64 * PA_AHEAD) {   // PA_AHEAD steps of loads in flight (see pass 1)
                uint32_t jv[PA_AHEAD];
                int hv[PA_AHEAD];
#pragma unroll
                for (int u = 0; u < PA_AHEAD; ++u) {
                    const uint32_t t = t0 + 64u * (uint32_t)u + (uint32_t)lane;
                    const bool ok = t < total;
                    const uint32_t tt = ok ? t : total - 1;
                    int h = 0;
#pragma unroll
                    for (int k = 0; k < BR_EPW - 1; ++k) h += tt >= bound[k];
                    const uint64_t start = ((uint64_t)(uint32_t)__shfl((int)lo_hi, h) << 32) | (uint32_t)__shfl((int)lo_lo, h);
                    const uint32_t first = (uint32_t)__shfl((int)excl, h);
                    hv[u] = ok ? h : -1;
                    jv[u] = __builtin_nontemporal_load(&qpos[start + (tt - first)]);   // read once: streaming load
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < PA_AHEAD; ++u) {
                    const uint32_t j = hv[u] >= 0 ? jv[u] : NONE32;
                    if (j != NONE32) {
                        const uint32_t k = j - (uint32_t)j0;
                        const uint32_t sub = k >> BR_SUB_BITS;
                        const uint32_t entry = ((uint32_t)(dbase + (uint64_t)hv[u]) << BR_SUB_BITS) | (k & (BR_SUB - 1));
                        const uint32_t at = atomicAdd(&s_gcur[sub], 1u);                   // final slot in this workgroup's stream
                        const uint32_t in_chunk = at - s_cbase[sub];
                        if (in_chunk < (uint32_t)PA_CAPS) s_stage[sub][in_chunk] = entry;
                        else inter[(uint64_t)s_gbase[sub] + at] = entry;
                    }
                }
            }
        }
        __syncthreads();
        for (int k = wave; k < BR_NSUB; k += WAVES) {               // stream k's part of the chunk, consecutive lanes
            const uint32_t from = s_cbase[k];
            uint32_t n = s_gcur[k] - from;
            n = n < (uint32_t)PA_CAPS ? n : (uint32_t)PA_CAPS;
            uint32_t* dst = inter + (uint64_t)s_gbase[k] + from;
            for (uint32_t i = lane; i < n; i += 64) dst[i] = s_stage[k][i];
        }
    }
}

// pass 2b: workgroup (window w, group g) counting-sorts the entries of its row blocks by list in LDS (at most
// BR_SORT_CAP at a time) and writes every list's run with consecutive lanes at post_off[j] + (postings of j in earlier row
// blocks) + (what earlier batches of this workgroup put there).  Windows w = x (mod 8) run on workgroup ids = x (mod 8),
// i.e. on one XCD, the groups of a window next to each other: the runs that neighbouring groups write into a list are
// adjacent, and the lines they share meet in that XCD's L2.
// ORDERED (row blocks per group <= BR_ORD_NB): the sort key is (list, row block), so that inside a list the entries of a
// row block are contiguous and blocks ascend -- the run of block b in list j is then exactly
// [post_off[j] + partial[b][j], post_off[j] + partial[b + 1][j]), which is what lets a workgroup of the persistent gather
// loop read only ITS rows' part of a list.
constexpr int BR_ORD_NB = 8;
// A window's postings arrive as `n_sub` runs, run i holding those of row block i / m (m = 1: one run per block, laid out by
// pass 2a; m > 1: one run per workgroup of the lean pass 1, m consecutive workgroups to a block), (start, length) in
// inter_off / subcnt [window][n_sub].
template <bool ORDERED>
__global__ __launch_bounds__(512) void build_scatter_kernel(uint64_t nq, uint32_t n_windows, uint32_t B,
                                                            const uint32_t* __restrict__ partial,
                                                            const uint64_t* __restrict__ post_off,
                                                            const uint32_t* __restrict__ subcnt,
                                                            const uint32_t* __restrict__ inter_off,
                                                            const uint32_t* __restrict__ inter, uint32_t* __restrict__ post_rows,
                                                            uint32_t n_sub, uint32_t m) {
    constexpr int NK = ORDERED ? BR_SUB * BR_ORD_NB : BR_SUB;        // sort keys
    __shared__ uint32_t s_sorted[BR_SORT_CAP];
    __shared__ uint32_t s_cnt[NK], s_start[NK + 1], s_fill[NK], s_cur[BR_SUB];
    __shared__ uint32_t s_pre[BR_GROUPS * 8 + 1], s_src[BR_GROUPS * 8];     // flat entry index -> region (at most 64 row blocks per group)
    __shared__ uint32_t s_wsum[8];
    const uint32_t q = blockIdx.x >> 3, x = blockIdx.x & 7u;
    const uint32_t w = (q / BR_GROUPS) * 8u + x, g = q % BR_GROUPS;
    if (w >= n_windows) return;
    const uint64_t j0 = (uint64_t)w * BR_SUB;
    if (j0 >= nq) return;
    const uint32_t nl = (uint32_t)(nq - j0 < (uint64_t)BR_SUB ? nq - j0 : (uint64_t)BR_SUB);
    const uint32_t per = (B + BR_GROUPS - 1) / BR_GROUPS;
    const uint32_t b_lo = g * per, b_hi = b_lo + per < B ? b_lo + per : B;
    if (b_lo >= B) return;
    const uint32_t i_lo = b_lo * m, i_hi = b_hi * m < n_sub ? b_hi * m : n_sub;
    const uint32_t nb = i_hi > i_lo ? i_hi - i_lo : 0u;               // runs of this group: <= 64 (enforced by the host)
    const uint64_t wbase = post_off[j0];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t k = tid; k < (uint32_t)BR_SUB; k += blockDim.x)
        s_cur[k] = k < nl ? (uint32_t)(post_off[j0 + k] - wbase) + partial[(uint64_t)b_lo * nq + j0 + k] : 0u;
    if (tid < 64) {                                                  // the runs' lengths, all loads at once, and their prefix
        const bool ok = (uint32_t)tid < nb;
        const uint32_t len = ok ? subcnt[(uint64_t)w * n_sub + i_lo + tid] : 0u;
        uint32_t incl = len;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (ok) {
            s_pre[tid] = incl - len;
            s_src[tid] = inter_off[(uint64_t)w * n_sub + i_lo + tid];
        }
        if (tid == 63) s_pre[nb] = incl;
    }
    __syncthreads();
    const uint32_t total = s_pre[nb];
    for (uint32_t f0 = 0; f0 < total; f0 += BR_SORT_CAP) {
        const uint32_t f1 = f0 + BR_SORT_CAP < total ? f0 + BR_SORT_CAP : total;
        for (uint32_t k = tid; k < (uint32_t)NK; k += blockDim.x) s_cnt[k] = 0;
        __syncthreads();
        // The batch is read ONCE, all of a thread's loads issued before the first is used (one load per step used to wait
        // for the one before: 24 dependent trips to L2 / HBM per thread and phase, and the batch was read twice).
        constexpr int PER = BR_SORT_CAP / 512;
        static_assert(BR_SORT_CAP % 512 == 0, "");
        uint32_t ent[PER];
        uint32_t key[PER];
        {
            uint32_t reg = 0;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const uint32_t f = f0 + (uint32_t)tid + (uint32_t)i * 512u;
                ent[i] = 0;
                key[i] = 0;
                if (f < f1) {
                    while (f >= s_pre[reg + 1]) ++reg;
                    ent[i] = __builtin_nontemporal_load(&inter[(uint64_t)s_src[reg] + (f - s_pre[reg])]);
                    key[i] = reg;
                }
            }
        }
        // histogram of the batch by sort key
#pragma unroll
        for (int i = 0; i < PER; ++i)
            if (f0 + (uint32_t)tid + (uint32_t)i * 512u < f1) {
                key[i] = ORDERED ? (ent[i] & (BR_SUB - 1)) * BR_ORD_NB + key[i] / m : (ent[i] & (BR_SUB - 1));
                atomicAdd(&s_cnt[key[i]], 1u);
            }
        __syncthreads();
        if (ORDERED) {
            // exclusive scan of NK = 2,048 counts: four consecutive keys per thread, wave scan, 8 wave totals
            static_assert(!ORDERED || NK == 4 * 512, "");
            const uint32_t c0 = s_cnt[4 * tid], c1 = s_cnt[4 * tid + 1], c2 = s_cnt[4 * tid + 2], c3 = s_cnt[4 * tid + 3];
            const uint32_t mine = c0 + c1 + c2 + c3;
            uint32_t incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d);
                if (lane >= d) incl += o;
            }
            if (lane == 63) s_wsum[wave] = incl;
            __syncthreads();
            uint32_t before = 0;
            for (int v = 0; v < wave; ++v) before += s_wsum[v];
            const uint32_t e0 = before + incl - mine;
            s_start[4 * tid] = e0;           s_fill[4 * tid] = e0;
            s_start[4 * tid + 1] = e0 + c0;  s_fill[4 * tid + 1] = e0 + c0;
            s_start[4 * tid + 2] = e0 + c0 + c1;  s_fill[4 * tid + 2] = e0 + c0 + c1;
            s_start[4 * tid + 3] = e0 + c0 + c1 + c2;  s_fill[4 * tid + 3] = e0 + c0 + c1 + c2;
            if (tid == 511) s_start[NK] = e0 + mine;
        } else if (wave == 0) {                                     // exclusive scan of BR_SUB counts by one wave
            uint32_t carry = 0;
            for (int base = 0; base < BR_SUB; base += 64) {
                const uint32_t v = s_cnt[base + lane];
                uint32_t incl = v;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t o = __shfl_up(incl, d);
                    if (lane >= d) incl += o;
                }
                s_start[base + lane] = carry + incl - v;
                s_fill[base + lane] = carry + incl - v;
                carry += __shfl(incl, 63);
            }
            if (lane == 0) s_start[NK] = carry;
        }
        __syncthreads();
        // placement
#pragma unroll
        for (int i = 0; i < PER; ++i)
            if (f0 + (uint32_t)tid + (uint32_t)i * 512u < f1)
                s_sorted[atomicAdd(&s_fill[key[i]], 1u)] = ent[i] >> BR_SUB_BITS;
        __syncthreads();
        // every list's run goes out with consecutive lanes
        constexpr uint32_t KPL = ORDERED ? BR_ORD_NB : 1;            // sort keys per list
        for (uint32_t jl = wave; jl < nl; jl += blockDim.x >> 6) {
            const uint32_t from = s_start[jl * KPL], n = s_start[(jl + 1) * KPL] - from;
            const uint64_t to = wbase + s_cur[jl];
            for (uint32_t i = lane; i < n; i += 64) post_rows[to + i] = s_sorted[from + i];
        }
        __syncthreads();
        for (uint32_t k = tid; k < (uint32_t)BR_SUB; k += blockDim.x)
            if (k < nl) s_cur[k] += s_start[(k + 1) * KPL] - s_start[k * KPL];
        __syncthreads();
    }
}

// What pass 1 leaves undone when it stages the postings itself (lean kernel, MODE 2): partial[b][j], the postings row block b adds
// to list j, counted from the runs.  Workgroup (window, group) as in the scatter kernel: its runs' entries are counted by
// (list, block of the group) in LDS and written out, 8 blocks x 256 lists.
__global__ __launch_bounds__(512) void build_count_runs_kernel(uint64_t nq, uint32_t n_windows, uint32_t B, uint32_t n_sub, uint32_t m,
                                                               const uint32_t* __restrict__ dir_start, const uint32_t* __restrict__ dir_len,
                                                               const uint32_t* __restrict__ inter, uint32_t* __restrict__ partial) {
    __shared__ uint32_t s_cnt[BR_SUB * BR_ORD_NB];
    const uint32_t q = blockIdx.x >> 3, x = blockIdx.x & 7u;
    const uint32_t w = (q / BR_GROUPS) * 8u + x, g = q % BR_GROUPS;
    if (w >= n_windows) return;
    const uint64_t j0 = (uint64_t)w * BR_SUB;
    if (j0 >= nq) return;
    const uint32_t nl = (uint32_t)(nq - j0 < (uint64_t)BR_SUB ? nq - j0 : (uint64_t)BR_SUB);
    const uint32_t per = (B + BR_GROUPS - 1) / BR_GROUPS;              // <= BR_ORD_NB (the host checks)
    const uint32_t b_lo = g * per, b_hi = b_lo + per < B ? b_lo + per : B;
    if (b_lo >= B) return;
    const uint32_t i_lo = b_lo * m, i_hi = b_hi * m < n_sub ? b_hi * m : n_sub;
    const uint32_t nreg = i_hi > i_lo ? i_hi - i_lo : 0u;              // <= 64
    const int tid = threadIdx.x, lane = tid & 63;
    __shared__ uint32_t s_pre[BR_GROUPS * 8 + 1], s_src[BR_GROUPS * 8];
    for (uint32_t k = tid; k < (uint32_t)(BR_SUB * BR_ORD_NB); k += blockDim.x) s_cnt[k] = 0;
    if (tid < 64) {
        const bool ok = (uint32_t)tid < nreg;
        const uint32_t len = ok ? dir_len[(uint64_t)w * n_sub + i_lo + tid] : 0u;
        uint32_t incl = len;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (ok) {
            s_pre[tid] = incl - len;
            s_src[tid] = dir_start[(uint64_t)w * n_sub + i_lo + tid];
        }
        if (tid == 63) s_pre[nreg] = incl;
    }
    __syncthreads();
    const uint32_t total = s_pre[nreg];
    constexpr int PER = 8;                                              // entries a thread has in flight
    for (uint32_t f0 = 0; f0 < total; f0 += 512u * PER) {
        uint32_t ent[PER], blk[PER];
        uint32_t reg = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const uint32_t f = f0 + (uint32_t)tid + (uint32_t)i * 512u;
            ent[i] = 0;
            blk[i] = ~0u;
            if (f < total) {
                while (f >= s_pre[reg + 1]) ++reg;
                ent[i] = __builtin_nontemporal_load(&inter[(uint64_t)s_src[reg] + (f - s_pre[reg])]);
                blk[i] = (i_lo + reg) / m - b_lo;
            }
        }
#pragma unroll
        for (int i = 0; i < PER; ++i)
            if (blk[i] != ~0u) atomicAdd(&s_cnt[(ent[i] & (uint32_t)(BR_SUB - 1)) * BR_ORD_NB + blk[i]], 1u);
    }
    __syncthreads();
    for (uint32_t k = tid; k < (uint32_t)BR_SUB * (b_hi - b_lo); k += blockDim.x) {
        const uint32_t bl = k / (uint32_t)BR_SUB, jl = k % (uint32_t)BR_SUB;
        if (jl < nl) partial[(uint64_t)(b_lo + bl) * nq + j0 + jl] = s_cnt[jl * BR_ORD_NB + bl];
    }
}

// partial[b][j] += post_off[j]: absolute slots, when all of them fit 32 bits (one load less per element in pass 2)
__global__ __launch_bounds__(256) void build_absolute_kernel(uint32_t* __restrict__ partial, uint32_t B, uint64_t nq,
                                                             const uint64_t* __restrict__ post_off) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nq) return;
    const uint32_t base = (uint32_t)post_off[j];
    for (uint32_t b = 0; b < B; ++b) partial[(uint64_t)b * nq + j] += base;
}

// partial[b][j] -> its exclusive prefix over b; post_cnt[j] = the sum (post_cnt[nq] = 0 for the scan)
__global__ __launch_bounds__(256) void build_merge_counts_kernel(uint32_t* __restrict__ partial, uint32_t B, uint64_t nq,
                                                                 unsigned long long* __restrict__ post_cnt) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > nq) return;
    unsigned long long run = 0;
    if (j < nq) {
        for (uint32_t b = 0; b < B; ++b) {
            const uint32_t v = partial[(uint64_t)b * nq + j];
            partial[(uint64_t)b * nq + j] = (uint32_t)run;
            run += v;
        }
    }
    post_cnt[j] = run;
}

// bounds[j][b] = post_off[j] + partial[b][j] for b < B, bounds[j][B] = post_off[j + 1]: the run of row block b inside posting list j
// as two neighbouring words (the resident loop asks for them once per newly covered hash and owned block; as partial[b][j],
// partial[b + 1][j] and post_off[j] they were three loads from three distant arrays).  64 lists x B blocks per workgroup,
// transposed through LDS so that reads run along j and writes along b.
__global__ __launch_bounds__(256) void build_bounds_table_kernel(const uint32_t* __restrict__ partial, uint64_t nq, uint32_t B,
                                                                 const uint64_t* __restrict__ post_off, uint32_t* __restrict__ bounds) {
    __shared__ uint32_t tile[64][65];
    const uint64_t j0 = (uint64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;             // 4 rows of 64 per step
    for (uint32_t bb = 0; bb < B; bb += 64) {
        for (int r = ty; r < 64; r += 4) {                              // r: block within the tile, tx: list
            const uint32_t b = bb + (uint32_t)r;
            tile[r][tx] = (b < B && j0 + tx < nq) ? partial[(uint64_t)b * nq + j0 + tx] : 0u;
        }
        __syncthreads();
        for (int r = ty; r < 64; r += 4) {                              // r: list within the tile, tx: block
            const uint64_t j = j0 + (uint64_t)r;
            const uint32_t b = bb + (uint32_t)tx;
            if (j < nq && b < B) bounds[j * (B + 1) + b] = (uint32_t)post_off[j] + tile[tx][r];
        }
        __syncthreads();
    }
    if (threadIdx.x < 64 && j0 + threadIdx.x < nq) bounds[(j0 + threadIdx.x) * (B + 1) + B] = (uint32_t)post_off[j0 + threadIdx.x + 1];
}

__global__ __launch_bounds__(256) void longest_row_kernel(const uint64_t* __restrict__ offsets, uint64_t ndb,
                                                         unsigned long long* out) {
    unsigned long long m = 0;
    for (uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; d < ndb; d += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long l = offsets[d + 1] - offsets[d];
        m = l > m ? l : m;
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

unsigned blocks_for_rows(uint64_t ndb) {
    const uint64_t b = (ndb + 3) / 4;
    return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace

hipError_t gather_build_kernel_ms(GatherDev& g, float* ms) {
    *ms = 0.f;
    if (!g.ev_build0 || !g.ev_build1) return hipSuccess;
    SMG_TRY(hipEventSynchronize(g.ev_build1));
    return hipEventElapsedTime(ms, g.ev_build0, g.ev_build1);
}

static hipError_t gather_build_body(GatherDev& g, hipStream_t stream);
hipError_t qtable_launch(const uint64_t* Q, uint64_t nq, uint32_t shift, uint32_t buckets, uint32_t* table, hipStream_t stream) {
    hipLaunchKernelGGL(qtable_kernel, dim3((buckets + 256) / 256), dim3(256), 0, stream, Q, nq, shift, buckets, table);
    return hipGetLastError();
}

hipError_t gather_build(GatherDev& g, hipStream_t stream) {
    g.stream = stream;
    const ArenaStats a0 = arena_stats();
    const uint64_t t0 = host_ns();
    g.build_syncs = g.build_sync_wait_ns = 0;
    const hipError_t e = gather_build_body(g, stream);
    if (e == hipSuccess && g.ev_build1) (void)hipEventRecord(g.ev_build1, stream);
    const ArenaStats a1 = arena_stats();
    g.build_host_ns = host_ns() - t0;
    g.build_driver_ns = a1.driver_ns - a0.driver_ns;
    g.build_driver_allocs = a1.driver_allocs - a0.driver_allocs;
    return e;
}

static hipError_t gather_build_body(GatherDev& g, hipStream_t stream) {
    if (g.nq >= NONE32) return hipErrorInvalidValue;             // query positions are u32
    if (g.ndb >= NONE32) return hipErrorInvalidValue;            // row ids are u32
    const uint64_t nq1 = g.nq + 1;
    SMG_TRY(arena_pinned_alloc((void**)&g.pinned, 32 * 8));
    SMG_TRY(hipEventCreate(&g.ev_build0));
    SMG_TRY(hipEventCreate(&g.ev_build1));
    SMG_TRY(own_alloc(g, &g.state, GS_SLOTS * 8));
    SMG_TRY(own_alloc(g, &g.partials, GATHER_PICK_BLOCKS * 8));
    SMG_TRY(own_alloc(g, &g.counters, (g.ndb + 1) * 8));
    SMG_TRY(own_alloc(g, &g.alive, g.nq + 16));
    SMG_TRY(own_alloc(g, &g.post_off, nq1 * 8));
    SMG_TRY(hipEventRecord(g.ev_build0, stream));
    SMG_TRY(hipMemsetAsync(g.state, 0, GS_SLOTS * 8, stream));
    SMG_TRY(hipMemsetAsync(g.counters, 0, (g.ndb + 1) * 8, stream));
    SMG_TRY(hipMemsetAsync(g.alive, 1, g.nq + 16, stream));
    g.pinned[3] = g.nq;
    SMG_TRY(hipMemcpyAsync(&g.state[GS_QLEN], &g.pinned[3], 8, hipMemcpyHostToDevice, stream));
    // database size and the largest query hash decide the table geometry: read back into pinned slots 0..2
    g.pinned[0] = g.pinned[1] = g.pinned[2] = 0;
    if (g.ndb) SMG_TRY(hipMemcpyAsync(&g.pinned[0], g.offsets + g.ndb, 8, hipMemcpyDeviceToHost, stream));
    if (g.ndb) {                                                  // state[GS_KEY] as scratch: zeroed above, zeroed again by begin
        hipLaunchKernelGGL(longest_row_kernel, dim3(blocks_for_rows(g.ndb) > 256 ? 256 : blocks_for_rows(g.ndb)), dim3(256), 0,
                           stream, g.offsets, g.ndb, &g.state[GS_KEY]);
        SMG_TRY(hipMemcpyAsync(&g.pinned[1], &g.state[GS_KEY], 8, hipMemcpyDeviceToHost, stream));
        SMG_TRY(hipMemsetAsync(&g.state[GS_KEY], 0, 8, stream));
    }
    if (g.nq) SMG_TRY(hipMemcpyAsync(&g.pinned[2], g.Q + g.nq - 1, 8, hipMemcpyDeviceToHost, stream));
    SMG_TRY(timed_sync(g, stream));                               // synchronisation 1 of 2
    const uint64_t total = g.pinned[0];
    g.longest_row = g.pinned[1];
    g.q_max = g.pinned[2];
    SMG_TRY(own_alloc(g, &g.q_padded, (g.nq + 4) * 8));
    if (g.nq) SMG_TRY(hipMemcpyAsync(g.q_padded, g.Q, g.nq * 8, hipMemcpyDeviceToDevice, stream));
    for (int i = 0; i < 4; ++i) g.pinned[4 + i] = g.q_max;        // pinned: stays valid until the object goes
    SMG_TRY(hipMemcpyAsync(g.q_padded + g.nq, &g.pinned[4], 4 * 8, hipMemcpyHostToDevice, stream));
    qindex_geometry(g.nq, g.q_max, &g.q_shift, &g.q_buckets);
    SMG_TRY(own_alloc(g, &g.q_table, ((uint64_t)g.q_buckets + 1) * 4));
    hipLaunchKernelGGL(qtable_kernel, dim3((g.q_buckets + 256) / 256), dim3(256), 0, stream, g.Q, g.nq, g.q_shift,
                       g.q_buckets, g.q_table);
    SMG_TRY(hipGetLastError());
    if (g.nq) {
        SMG_TRY(own_alloc(g, &g.q_rec, (uint64_t)g.q_buckets * sizeof(QRec)));
        hipLaunchKernelGGL(qrec_kernel, dim3((g.q_buckets + 255) / 256), dim3(256), 0, stream, g.Q, g.q_table, g.q_buckets, g.q_rec);
        SMG_TRY(hipGetLastError());
    }
    if (g.ndb == 0 || total == 0 || g.nq == 0) {
        SMG_TRY(hipMemsetAsync(g.post_off, 0, nq1 * 8, stream));
        g.npairs = 0;
        return hipSuccess;
    }
    const QIndex qi = qindex_of(g);
    ArenaBuf post_cnt_b, scan_tmp_b, bounds_b, partial_b, subcnt_b, inter_off_b, lay_tmp_b, inter_b, desc_b, misc_b, lean_table_b;
    SMG_TRY(post_cnt_b.get(nq1 * 8, stream));
    unsigned long long* post_cnt = post_cnt_b.as<unsigned long long>();
    size_t scan_bytes = 0;
    SMG_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, (uint64_t*)post_cnt, g.post_off, (uint64_t)0, (size_t)nq1,
                                    rocprim::plus<uint64_t>(), stream));
    SMG_TRY(scan_tmp_b.get(scan_bytes + 256, stream));
    void* scan_tmp = scan_tmp_b.p;
    SMG_TRY(own_alloc(g, &g.qpos, (total + 4) * 4));                // kept: apply reads it instead of looking hashes up again
    // Small problems: one atomic per element is cheapest.  Large ones: range-partitioned, histogram and cursors in LDS.
    const char* force = getenv("SMG_GATHER_BUILD");
    // (round 3: from 1 M elements up instead of 8 M -- the range builder also leaves the block-ordered lists the persistent loop
    //  needs: 5,000 x 1,000 with a 2e5-hash query 0.41 + 11.8 ms -> 0.48 + 9.1 ms, a 12,500-row shard of C5 3.8 + 29 -> 1.9 + 20)
    bool ranges = force ? !strcmp(force, "ranges") : (total >= (1ull << 20) && g.ndb >= 256);
    // scratch of the range-partitioned builder: B x nq per-block prefixes and (R + 1) x ndb slice bounds, 4 bytes each.
    // B can shrink to what the 16-bit LDS slots allow (< 65536 rows per block); past 8 GB the atomic builder is used.
    uint64_t B = 64;
    const uint64_t R64 = (g.nq + BR_RANGE - 1) / BR_RANGE, B_min = (g.ndb + 65534) / 65535;
    while (B > B_min && B > 1 && B * g.nq * 4 > (4ull << 30)) B /= 2;
    if (B < B_min) B = B_min;
    if (ranges && !force && (B * g.nq + (R64 + 1) * g.ndb) * 4 > (8ull << 30)) ranges = false;
    if (!ranges) {
        uint32_t* qpos = g.qpos;
        SMG_TRY(hipMemsetAsync(post_cnt, 0, nq1 * 8, stream));
        hipLaunchKernelGGL(build_count_kernel, dim3(blocks_for_rows(g.ndb)), dim3(256), 0, stream, qi, g.hashes, g.offsets,
                           g.ndb, qpos, post_cnt, g.counters);
        SMG_TRY(hipGetLastError());
        SMG_TRY(rocprim::exclusive_scan(scan_tmp, scan_bytes, (uint64_t*)post_cnt, g.post_off, (uint64_t)0, (size_t)nq1,
                                        rocprim::plus<uint64_t>(), stream));
        SMG_TRY(hipMemcpyAsync(&g.pinned[8], g.post_off + g.nq, 8, hipMemcpyDeviceToHost, stream));
        SMG_TRY(hipMemcpyAsync(post_cnt, g.post_off, nq1 * 8, hipMemcpyDeviceToDevice, stream));   // cursors
        SMG_TRY(timed_sync(g, stream));                           // synchronisation 2 of 2
        g.npairs = g.pinned[8];
        SMG_TRY(own_alloc(g, &g.post_rows, (g.npairs + 4) * 4));
        hipLaunchKernelGGL(build_fill_kernel, dim3(blocks_for_rows(g.ndb)), dim3(256), 0, stream, qpos, g.offsets, g.ndb,
                           post_cnt, g.post_rows);
        SMG_TRY(hipGetLastError());
    } else {
        const uint32_t R = (uint32_t)((g.nq + BR_RANGE - 1) / BR_RANGE);
        if (B > (g.ndb + 127) / 128) B = (g.ndb + 127) / 128;       // at least one full step (8 waves x 16 rows) per block
        if (B < B_min) B = B_min;
        if (B < 1) B = 1;
        uint64_t rows_per_block = (g.ndb + B - 1) / B;
        // two-level fill unless forced off or its packing does not apply (entries hold the row in 24 bits, offsets in 32)
        const char* fill_env = getenv("SMG_GATHER_FILL");
        bool staged = !(fill_env && !strcmp(fill_env, "direct")) && g.ndb < (1ull << BR_ROWBITS);
        // Pass 1 (+ 2a) through the lean streaming kernel's staging form (SMG_GATHER_PASS1=stage|ranges forces / forbids it): the query
        // flows through LDS in ranges and every workgroup walks its own rows once -- no lookups in L2.  It wants a few hundred rows
        // per CU, a query without the hash 2^64 - 1, ranges that fit its LDS (one more synchronisation: the widest range's size
        // comes back first), and it works per WORKGROUP: a row block of the builder becomes `m_sub` consecutive workgroups of `rpw`
        // rows.  Anything it cannot take goes to pass 1 by lookups in L2 (build_range_kernel<0>); a forced form that cannot run is
        // an error.  (Round 4 kept a third form between the two -- the lean kernel counting only, SMG_GATHER_PASS1=lean -- as the
        // fallback for queries whose ranges do not fit the staging form's table slice; the lookups take those now.)
        const char* pass1_s = getenv("SMG_GATHER_PASS1");
        const int pass1_env = !pass1_s ? 0 : !strcmp(pass1_s, "ranges") ? 1 : !strcmp(pass1_s, "stage") ? 3 : 0;
        bool stage1 = false;
        LeanPlan lp{};
        uint64_t m_sub = 1, rpw = 0, n_sub = 0;
        uint32_t stage_W = 0, stage_ranges = 0;
        uint32_t lean_shift = g.q_shift, lean_buckets = g.q_buckets;
        uint32_t* lean_T = g.q_table;
        const uint32_t n_windows = R * BR_NSUB;
        {
            int n_cu = 256;
            { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); }
            if (pass1_env != 1 && staged && B <= (uint64_t)MS_BMAX && g.q_max != ~0ull && (pass1_env == 3 || g.ndb >= (uint64_t)n_cu * 64)) {
                // (the streaming kernels take a coarser table of their own when the shared one has close to two buckets per hash)
                const double mean_row = (double)total / (double)g.ndb;
                lean_table_geometry(g.nq, g.q_max, mean_row, &lean_shift, &lean_buckets);
                if (lean_buckets != g.q_buckets) {
                    SMG_TRY(lean_table_b.get(((uint64_t)lean_buckets + 1) * 4 + 64, stream));
                    lean_T = lean_table_b.as<uint32_t>();
                    hipLaunchKernelGGL(qtable_kernel, dim3((lean_buckets + 256) / 256), dim3(256), 0, stream, g.Q, g.nq, lean_shift, lean_buckets, lean_T);
                    SMG_TRY(hipGetLastError());
                }
                lp = build_lean_plan(g.nq, lean_buckets, mean_row);
                stage_W = build_stage_positions(g.nq, lean_buckets, mean_row);
                stage_ranges = (uint32_t)((g.nq + stage_W - 1) / stage_W);
                SMG_TRY(desc_b.get(build_stage_desc_bytes(stage_ranges) + 64, stream));
                unsigned int* d_widest = (unsigned int*)&g.state[GS_KEY];       // scratch again: zero since the first synchronisation
                SMG_TRY(build_stage_plan(g.Q, g.nq, lean_shift, lean_buckets, stage_W, stage_ranges, desc_b.p, d_widest + 1, stream));
                g.pinned[9] = 0;
                SMG_TRY(hipMemcpyAsync(&g.pinned[9], d_widest, 8, hipMemcpyDeviceToHost, stream));
                SMG_TRY(hipMemsetAsync(&g.state[GS_KEY], 0, 8, stream));
                SMG_TRY(timed_sync(g, stream));                           // synchronisation 2 of 3
                const unsigned int most_buckets = (unsigned int)(g.pinned[9] >> 32);
                const uint64_t rows_cap = lp.rows_cap < build_stage_rows_max() ? lp.rows_cap : build_stage_rows_max();
                const uint64_t m_min = (rows_per_block + rows_cap - 1) / rows_cap;
                const uint64_t rounds = (B * m_min + (uint64_t)n_cu - 1) / (uint64_t)n_cu;
                m_sub = rounds * (uint64_t)n_cu / B;                        // full rounds of resident workgroups, no tail of a few
                if (m_sub < m_min) m_sub = m_min;
                rpw = (rows_per_block + m_sub - 1) / m_sub;
                n_sub = (g.ndb + rpw - 1) / rpw;
                const uint64_t per = (B + BR_GROUPS - 1) / BR_GROUPS;
                stage1 = most_buckets > 0 && most_buckets <= build_stage_buckets_max() && total + 4 < 0xffffffffull &&
                         per <= (uint64_t)BR_ORD_NB && per * m_sub <= 64 && n_sub * (uint64_t)((g.nq + BR_SUB - 1) / BR_SUB) * 8 <= (1ull << 30);
                if (!stage1 && pass1_env == 3) return hipErrorInvalidValue;
                if (stage1) {
                    rows_per_block = rpw * m_sub;
                    B = (g.ndb + rows_per_block - 1) / rows_per_block;
                }
            }
        }
        if (stage1) {
            // pass 1 + 2a in one kernel: query positions, postings staged by window and written as runs; then the runs are counted
            const uint64_t win_used = (g.nq + BR_SUB - 1) / BR_SUB;
            SMG_TRY(partial_b.get(B * g.nq * 4, stream));
            SMG_TRY(inter_b.get((total + 4) * 4, stream));
            SMG_TRY(subcnt_b.get(win_used * n_sub * 4 + 64, stream));         // the directory: a run's length ...
            SMG_TRY(inter_off_b.get(win_used * n_sub * 4 + 64, stream));      // ... and start
            SMG_TRY(misc_b.get(64, stream));
            SMG_TRY(hipMemsetAsync(misc_b.p, 0, 64, stream));
            uint32_t* partial = partial_b.as<uint32_t>();
            SMG_TRY(build_stage_launch(g.Q, g.nq, lean_T, lean_buckets, lean_shift, g.hashes, g.offsets, g.ndb, (uint32_t)rpw, stage_ranges,
                                       desc_b.p, g.counters, g.qpos, inter_b.as<uint32_t>(), inter_off_b.as<uint32_t>(), subcnt_b.as<uint32_t>(),
                                       misc_b.as<unsigned int>(), stream));
            const unsigned win_grid = (unsigned)((n_windows + 7) / 8 * 8 * BR_GROUPS);
            hipLaunchKernelGGL(build_count_runs_kernel, dim3(win_grid), dim3(512), 0, stream, g.nq, n_windows, (uint32_t)B, (uint32_t)n_sub,
                               (uint32_t)m_sub, (const uint32_t*)inter_off_b.as<uint32_t>(), (const uint32_t*)subcnt_b.as<uint32_t>(),
                               (const uint32_t*)inter_b.as<uint32_t>(), partial);
            SMG_TRY(hipGetLastError());
            hipLaunchKernelGGL(build_merge_counts_kernel, dim3((unsigned)((nq1 + 255) / 256)), dim3(256), 0, stream, partial, (uint32_t)B, g.nq, post_cnt);
            SMG_TRY(hipGetLastError());
            SMG_TRY(rocprim::exclusive_scan(scan_tmp, scan_bytes, (uint64_t*)post_cnt, g.post_off, (uint64_t)0, (size_t)nq1,
                                            rocprim::plus<uint64_t>(), stream));
            SMG_TRY(hipMemcpyAsync(&g.pinned[8], g.post_off + g.nq, 8, hipMemcpyDeviceToHost, stream));
            g.pinned[10] = 0;
            SMG_TRY(hipMemcpyAsync(&g.pinned[10], misc_b.p, 8, hipMemcpyDeviceToHost, stream));
            SMG_TRY(timed_sync(g, stream));                               // synchronisation 3 of 3
            if ((g.pinned[10] >> 32) != 0) {
                // a window's part of the staging area overflowed (many rows meeting in a few lists): build by the lookup form instead
                if (pass1_env == 3) return hipErrorInvalidValue;
                stage1 = false;
                SMG_TRY(hipMemsetAsync(g.counters, 0, (g.ndb + 1) * 8, stream));
            } else {
                g.npairs = g.pinned[8];
                SMG_TRY(own_alloc(g, &g.post_rows, (g.npairs + 4) * 4));
                hipLaunchKernelGGL(build_scatter_kernel<true>, dim3(win_grid), dim3(512), 0, stream, g.nq, n_windows, (uint32_t)B,
                                   (const uint32_t*)partial, (const uint64_t*)g.post_off, (const uint32_t*)subcnt_b.as<uint32_t>(),
                                   (const uint32_t*)inter_off_b.as<uint32_t>(), (const uint32_t*)inter_b.as<uint32_t>(), g.post_rows, (uint32_t)n_sub,
                                   (uint32_t)m_sub);
                SMG_TRY(hipGetLastError());
                SMG_TRY(own_alloc(g, &g.block_pre, (g.nq * (B + 1) + 4) * 4));
                hipLaunchKernelGGL(build_bounds_table_kernel, dim3((unsigned)((g.nq + 63) / 64)), dim3(256), 0, stream, (const uint32_t*)partial,
                                   g.nq, (uint32_t)B, (const uint64_t*)g.post_off, g.block_pre);
                SMG_TRY(hipGetLastError());
                g.block_B = (uint32_t)B;
                g.block_rows = (uint32_t)rows_per_block;
                return hipSuccess;
            }
        }
        SMG_TRY(bounds_b.get(((uint64_t)R + 1) * g.ndb * 4, stream));
        SMG_TRY(partial_b.get(B * g.nq * 4, stream));
        uint32_t *bounds = bounds_b.as<uint32_t>(), *partial = partial_b.as<uint32_t>();
        hipLaunchKernelGGL(build_bounds_kernel, dim3(blocks_for_rows(g.ndb)), dim3(256), 0, stream, g.Q, R, g.hashes,
                           g.offsets, g.ndb, bounds);
        SMG_TRY(hipGetLastError());
        uint32_t *subcnt = nullptr, *inter_off = nullptr, *inter = nullptr;
        if (staged) {
            SMG_TRY(subcnt_b.get((uint64_t)n_windows * B * 4, stream));
            SMG_TRY(inter_off_b.get((uint64_t)n_windows * B * 4, stream));
            subcnt = subcnt_b.as<uint32_t>();
            inter_off = inter_off_b.as<uint32_t>();
            SMG_TRY(hipMemsetAsync(subcnt, 0, (uint64_t)n_windows * B * 4, stream));   // ranges past R launch nothing
        }
        const unsigned range_grid = (unsigned)(((R + 7) / 8) * 8 * B);
        hipLaunchKernelGGL(build_range_kernel<0>, dim3(range_grid), dim3(BR_THREADS), 0, stream, qi, g.hashes, g.offsets,
                           g.ndb, bounds, R, (uint32_t)B, rows_per_block, partial, (const uint64_t*)nullptr, (uint32_t*)nullptr,
                           g.counters, g.qpos, subcnt);
        SMG_TRY(hipGetLastError());
        hipLaunchKernelGGL(build_merge_counts_kernel, dim3((unsigned)((nq1 + 255) / 256)), dim3(256), 0, stream, partial,
                           (uint32_t)B, g.nq, post_cnt);
        SMG_TRY(hipGetLastError());
        SMG_TRY(rocprim::exclusive_scan(scan_tmp, scan_bytes, (uint64_t*)post_cnt, g.post_off, (uint64_t)0, (size_t)nq1,
                                        rocprim::plus<uint64_t>(), stream));
        SMG_TRY(hipMemcpyAsync(&g.pinned[8], g.post_off + g.nq, 8, hipMemcpyDeviceToHost, stream));
        if (staged) {
            // region of (window, row block) in the intermediate buffer: exclusive scan of the exact counts, window-major
            size_t lay_bytes = 0;
            SMG_TRY(rocprim::exclusive_scan(nullptr, lay_bytes, subcnt, inter_off, 0u, (size_t)n_windows * B, rocprim::plus<uint32_t>(), stream));
            SMG_TRY(lay_tmp_b.get(lay_bytes + 256, stream));
            SMG_TRY(rocprim::exclusive_scan(lay_tmp_b.p, lay_bytes, subcnt, inter_off, 0u, (size_t)n_windows * B, rocprim::plus<uint32_t>(), stream));
        }
        SMG_TRY(timed_sync(g, stream));                           // the last synchronisation (2 of 2, or 3 of 3 with the lean pass 1)
        g.npairs = g.pinned[8];
        SMG_TRY(own_alloc(g, &g.post_rows, (g.npairs + 4) * 4));
        const uint64_t inter_words = g.npairs + 4;
        if (staged && (inter_words >= 0xffffffffull || B > 512)) staged = false;
        if (staged) {
            SMG_TRY(inter_b.get(inter_words * 4, stream));
            inter = inter_b.as<uint32_t>();
            hipLaunchKernelGGL(build_partition_kernel, dim3(range_grid), dim3(BR_THREADS), 0, stream, g.nq, g.offsets, g.ndb,
                               (const uint32_t*)bounds, R, (uint32_t)B, rows_per_block, (const uint32_t*)g.qpos,
                               (const uint32_t*)inter_off, inter);
            SMG_TRY(hipGetLastError());
            const uint32_t per = (uint32_t)((B + BR_GROUPS - 1) / BR_GROUPS);
            const bool ordered = per <= (uint32_t)BR_ORD_NB;            // row-block runs inside every list (persistent loop)
            if (ordered)
                hipLaunchKernelGGL(build_scatter_kernel<true>, dim3((unsigned)((n_windows + 7) / 8 * 8 * BR_GROUPS)), dim3(512), 0, stream, g.nq,
                                   n_windows, (uint32_t)B, (const uint32_t*)partial, (const uint64_t*)g.post_off, (const uint32_t*)subcnt,
                                   (const uint32_t*)inter_off, (const uint32_t*)inter, g.post_rows, (uint32_t)B, 1u);
            else
                hipLaunchKernelGGL(build_scatter_kernel<false>, dim3((unsigned)((n_windows + 7) / 8 * 8 * BR_GROUPS)), dim3(512), 0, stream, g.nq,
                                   n_windows, (uint32_t)B, (const uint32_t*)partial, (const uint64_t*)g.post_off, (const uint32_t*)subcnt,
                                   (const uint32_t*)inter_off, (const uint32_t*)inter, g.post_rows, (uint32_t)B, 1u);
            SMG_TRY(hipGetLastError());
            if (ordered && g.npairs < 0xffffffffull) {
                // where every row block's run begins and ends inside every list, for the resident loop: [nq][B + 1]
                SMG_TRY(own_alloc(g, &g.block_pre, (g.nq * (B + 1) + 4) * 4));
                hipLaunchKernelGGL(build_bounds_table_kernel, dim3((unsigned)((g.nq + 63) / 64)), dim3(256), 0, stream, (const uint32_t*)partial,
                                   g.nq, (uint32_t)B, (const uint64_t*)g.post_off, g.block_pre);
                SMG_TRY(hipGetLastError());
                g.block_B = (uint32_t)B;
                g.block_rows = (uint32_t)rows_per_block;
            }
        } else {
            const bool absolute = g.npairs < 0xffffffffull;
            if (absolute) {
                hipLaunchKernelGGL(build_absolute_kernel, dim3((unsigned)((g.nq + 255) / 256)), dim3(256), 0, stream, partial,
                                   (uint32_t)B, g.nq, (const uint64_t*)g.post_off);
                SMG_TRY(hipGetLastError());
            }
            hipLaunchKernelGGL(build_range_kernel<1>, dim3(range_grid), dim3(BR_THREADS), 0, stream, qi, g.hashes, g.offsets,
                               g.ndb, bounds, R, (uint32_t)B, rows_per_block, partial,
                               absolute ? (const uint64_t*)nullptr : (const uint64_t*)g.post_off, g.post_rows, g.counters, g.qpos,
                               (uint32_t*)nullptr);
            SMG_TRY(hipGetLastError());
        }
    }
    return hipSuccess;       // scratch blocks go back to the arena tagged with `stream` (ArenaBuf destructors): no wait needed
}

}  // namespace smg
