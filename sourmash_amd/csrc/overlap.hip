// overlap.hip -- streaming walks over a resident collection against a large query: the query flows through LDS in ranges, the
// database is read once.
//
//   overlap_ranges_launch   |Q ∩ row| for every row: the overlap pass of search / prefetch over a collection
//                           (src/core/src/index/linear.rs:52-113, src/sourmash/index/__init__.py:115-170,241-256 walk the rows one by
//                           one) -- overlap_lean_kernel<0> for collections that give every CU a few dozen rows, stream_lookup_kernel
//                           below that, the one-wave-per-row kernel of pair_ops.hip for what neither can take
//   build_stage_launch      the same walk as pass 1 + 2a of the gather index build (gather_build.hip): overlap_lean_kernel<2> also
//                           leaves every database hash's query position and the postings, staged by window
//
// Split from gather.hip in round 5 (Weak 10), with the superseded forms removed: round 3's overlap_wide_kernel and its
// two-workgroups-per-CU geometry, the lean kernel's counting form (MODE 1), the range-partitioned overlap pass by lookups in L2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <stdlib.h>
#include "gather_api.hpp"
#include "arena.hpp"
#include "wavemask.hpp"
#include "qindex.hpp"
#include "gather_parts.hpp"

namespace smg {

// largest number of query hashes in any range of `bpr` buckets (the caller checks it against the LDS room of the kernel that
// walks the ranges)
__global__ __launch_bounds__(256) void stream_range_max_kernel(const uint32_t* __restrict__ T, uint32_t n_buckets, uint32_t n_ranges,
                                                              uint32_t bpr, unsigned int* out) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_ranges) return;
    const uint32_t b0 = r * bpr, b1 = b0 + bpr < n_buckets ? b0 + bpr : n_buckets;
    atomicMax(out, T[b1] - T[b0]);
}
static hipError_t stream_range_max_launch(const uint32_t* T, uint32_t n_buckets, uint32_t n_ranges, uint32_t bpr, unsigned int* out,
                                          hipStream_t stream) {
    hipLaunchKernelGGL(stream_range_max_kernel, dim3((n_ranges + 255) / 256), dim3(256), 0, stream, T, n_buckets, n_ranges, bpr, out);
    return hipGetLastError();
}

// ---- streaming lookups: the query flows through LDS, the database is read once -----------------------------------
// The range kernels above look every database hash up in structures that live in L2 (two dependent 16-byte loads per
// element from lines nobody else in the wave touches): 5e8 lookups cost 5-6 ms at C5 whatever else the pass does.
// Here a workgroup owns a block of rows and walks the QUERY in order: the hash space is cut at multiples of
// SL_BUCKETS buckets of the first-level table (about SL_BUCKETS query hashes each, because there is about one hash per
// bucket); for each such range the workgroup loads that slice of the table and of the query into LDS (32 KB), then
// visits each of its rows once: a group of 16 lanes reads the row's next 16 hashes at the row's cursor, the ones below
// the range's upper bound are looked up in LDS (table entry -> at most a few query hashes, compared in registers) and
// consumed, the cursor moves on.  Rows are sorted, so a row's hashes inside a range are contiguous and every database
// hash is read from HBM once, by consecutive lanes.  No per-(row, range) bounds are precomputed: the cursors carry over.
// Workgroup (block b, group g) covers ranges [g * per, (g + 1) * per); its first cursors come from a binary search.
constexpr int SL_BUCKETS = 2048;          // table buckets per range
constexpr int SL_QCAP = 2432;             // query hashes a range may hold (at most ~2,048 by construction, +- 45); else the caller falls back.
                                          // 39.9 KB of LDS in all: four workgroups (32 waves) per CU
constexpr int SL_ROWS = 1024;             // rows per block (row starts, cursors and per-row hit counts live in LDS)
constexpr int SL_THREADS = 512;
constexpr int SL_GROUP = 16;              // lanes per row visit
constexpr int SL_GROUPS = SL_THREADS / SL_GROUP;
constexpr int SL_AHEAD = 4;               // row visits a group has in flight: a visit is one ~1 us load from HBM, and 51 million of
                                          // them (rows x ranges at C5) must overlap

// counters[d] += |Q ∩ row d|
__global__ __launch_bounds__(SL_THREADS) void stream_lookup_kernel(const uint64_t* __restrict__ Q, const uint32_t* __restrict__ T,
                                                                   uint32_t n_buckets, uint32_t shift, uint64_t qmax,
                                                                   const uint64_t* __restrict__ hashes,
                                                                   const uint64_t* __restrict__ offsets, uint64_t ndb,
                                                                   uint32_t n_blocks, uint32_t n_ranges, uint32_t ranges_per_group,
                                                                   uint32_t bpr, unsigned long long* __restrict__ counters) {
    // bpr: buckets per range (<= SL_BUCKETS): the table holds between one and two query hashes per bucket, the caller picks
    // the power of two that puts about 2,000 of them into a range
    __shared__ __attribute__((aligned(16))) uint64_t s_q[SL_QCAP];
    __shared__ __attribute__((aligned(16))) uint32_t s_t[SL_BUCKETS + 4];
    __shared__ uint32_t s_base[SL_ROWS + 1];                             // row starts relative to the block's first hash
    __shared__ uint32_t s_cur[SL_ROWS], s_hits[SL_ROWS];
    const uint32_t b = blockIdx.x % n_blocks, g = blockIdx.x / n_blocks;
    const uint64_t d_lo = (uint64_t)b * SL_ROWS;
    const uint32_t n_rows = (uint32_t)(ndb - d_lo < (uint64_t)SL_ROWS ? ndb - d_lo : (uint64_t)SL_ROWS);
    const uint32_t r_lo = g * ranges_per_group;
    const uint32_t r_hi = r_lo + ranges_per_group < n_ranges ? r_lo + ranges_per_group : n_ranges;
    if (r_lo >= r_hi) return;
    const int tid = threadIdx.x;
    const uint64_t block_base = offsets[d_lo];
    const uint64_t* rows = hashes + block_base;
    // first cursors: where the group's first range starts in every row
    const uint64_t first_hash = ((uint64_t)r_lo * bpr) << shift;
    for (uint32_t i = tid; i <= n_rows; i += SL_THREADS) s_base[i] = (uint32_t)(offsets[d_lo + i] - block_base);
    __syncthreads();
    for (uint32_t i = tid; i < n_rows; i += SL_THREADS) {
        uint32_t lo = 0, hi = s_base[i + 1] - s_base[i];
        const uint64_t* row = rows + s_base[i];
        if (r_lo != 0) {
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (row[mid] < first_hash) lo = mid + 1; else hi = mid;
            }
        } else {
            lo = 0;
        }
        s_cur[i] = lo;
        s_hits[i] = 0;
    }
    const int grp = tid / SL_GROUP, gl = tid % SL_GROUP;
    const int sh16 = (tid & 63) / SL_GROUP * SL_GROUP;                   // this group's bit offset inside the wave's ballot
    for (uint32_t r = r_lo; r < r_hi; ++r) {
        const uint32_t b0 = r * bpr, b1 = b0 + bpr < n_buckets ? b0 + bpr : n_buckets;
        const uint32_t p0 = T[b0], p1 = T[b1];
        const bool last = r + 1 == n_ranges;
        // hashes below `upper` belong to this range or an earlier one (earlier ones are consumed already)
        const uint64_t upper = last ? ~0ull : ((uint64_t)b1 << shift);
        __syncthreads();                                                  // the previous range's readers are done
        for (uint32_t i = tid; i < b1 - b0 + 1; i += SL_THREADS) s_t[i] = T[b0 + i] - p0;
        for (uint32_t i = tid; i < p1 - p0; i += SL_THREADS) s_q[i] = Q[p0 + i];
        __syncthreads();
        // one group of 16 lanes per row; SL_AHEAD rows' loads are issued before the first of them is looked at
        for (uint32_t i0 = grp; i0 < n_rows; i0 += SL_GROUPS * SL_AHEAD) {
            uint64_t e[SL_AHEAD];
            uint32_t c[SL_AHEAD], len[SL_AHEAD], rb[SL_AHEAD];
#pragma unroll
            for (int u = 0; u < SL_AHEAD; ++u) {
                const uint32_t i = i0 + u * SL_GROUPS;
                const bool row_ok = i < n_rows;
                rb[u] = row_ok ? s_base[i] : 0u;
                len[u] = row_ok ? s_base[i + 1] - rb[u] : 0u;
                c[u] = row_ok ? s_cur[i] : 0u;
                e[u] = c[u] + gl < len[u] ? rows[(uint64_t)rb[u] + c[u] + gl] : ~0ull;
            }
#pragma unroll
            for (int u = 0; u < SL_AHEAD; ++u) {
                const uint32_t i = i0 + u * SL_GROUPS;
                uint32_t cur = c[u], hits = 0;
                uint64_t ev = e[u];
                for (;;) {
                    const bool have = cur + gl < len[u];
                    const bool in = have && (last || ev < upper);
                    const uint32_t taken = (uint32_t)__popc((uint32_t)((__ballot(in) >> sh16) & 0xffffu));
                    bool hit = false;
                    if (in && ev <= qmax) {
                        const uint32_t k = (uint32_t)(ev >> shift) - b0;         // < SL_BUCKETS: the hash lies in this range
                        const uint32_t t0 = s_t[k], t1 = s_t[k + 1];
                        for (uint32_t t = t0; t < t1; ++t) {
                            const uint64_t qv = s_q[t];
                            if (qv == ev) { hit = true; break; }
                            if (qv > ev) break;
                        }
                    }
                    hits += (uint32_t)__popc((uint32_t)((__ballot(hit) >> sh16) & 0xffffu));
                    cur += taken;
                    if (taken < (uint32_t)SL_GROUP) break;                // the row's part of this range is through
                    ev = cur + gl < len[u] ? rows[(uint64_t)rb[u] + cur + gl] : ~0ull;   // a longer slice: keep reading
                }
                if (gl == 0 && i < n_rows) { s_cur[i] = cur; if (hits) s_hits[i] += hits; }
            }
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < n_rows; i += SL_THREADS)
        if (s_hits[i]) atomicAdd(&counters[d_lo + i], (unsigned long long)s_hits[i]);   // one add per (row, group of ranges)
}

// ---- the lean walk: one workgroup per CU, a wave per row visit ------------------------------------------------------
// stream_lookup_kernel above spends 118 lane-instructions per database hash (profiles/r02_gather_sq.txt: VALU issue 55 % busy
// for 5e8 lookups): with ~2,000 query hashes per range a row's part of a range is ~10 hashes, so 51 million visits each pay
// cursor / ballot / bounds bookkeeping for ten useful lanes of sixteen, and every visit's 128-byte read overlaps the next
// one's.  Here a workgroup takes the whole LDS of a CU: ranges of ~10,000 query hashes (88 KB) and ~10,000 table buckets
// (40 KB), so that a row's part of a range is ~50 hashes and ONE wave reads it as one 512-byte load: six times fewer visits,
// most lanes busy.  A workgroup owns a contiguous block of rows and walks every range over them, so every row is counted by
// exactly one workgroup: plain stores, no atomics.  (Round 3's form of this kernel -- row cursors in LDS, per-lane flags;
// overlap_wide_kernel and its two-workgroups-per-CU geometry -- was superseded by the lean visits below in round 4 and removed in
// round 5: profiles/HISTORY.md 4.4 keeps its measurements.)
constexpr int OW_THREADS = 1024;
constexpr int OW_WAVES = OW_THREADS / 64;
template <int SLOTS_, int BUCKETS_, int QCAP_, typename TT_, int WAVES_PER_EU_, int BATCH_>
struct OwGeom {
    static constexpr int BATCH = BATCH_;          // visits looked up side by side
    static constexpr int SLOTS = SLOTS_;          // rows a wave owns, each with its next 64 hashes in (or on the way to) registers
    static constexpr int BUCKETS = BUCKETS_;      // table buckets per range (at most)
    static constexpr int QCAP = QCAP_;            // query hashes a range may hold; the caller checks the widest range
    using TT = TT_;                               // a table entry in LDS: offset of the bucket's first query hash within the slice
    static constexpr int ROWS = OW_WAVES * SLOTS_; // rows per workgroup (at most)
    static constexpr size_t T_BYTES = (((size_t)BUCKETS_ + 4) * sizeof(TT_) + 7) & ~(size_t)7;
    static constexpr size_t LDS = (size_t)QCAP_ * 8 + T_BYTES + ((size_t)3 * ROWS + 8) * 4;
    static constexpr int WAVES_PER_EU = WAVES_PER_EU_;
};
using OwLean = OwGeom<25, 10240, 11264, uint32_t, 4, 4>;  // the lean kernel's geometry: ranges cut by query hashes held, not at a power of two of buckets

// Query hashes per range of the streaming kernels.  A visit loads the next 64 hashes of its row and uses the ones below the
// range's upper bound: a range should hold so many query hashes that a row's part of it is ~48 hashes (more, and one visit in
// sixteen has to fetch a second block on the spot; fewer, and the visits multiply) -- 48 x nq / (mean row length), at most what
// the LDS slice holds.  C5: 48 x 1e6 / 5,000 = 9,600.
double lean_hashes_per_range(uint64_t nq, double mean_row) {
    double q = 48.0 * (double)nq / (mean_row < 1.0 ? 1.0 : mean_row);
    if (q > 9600.0) q = 9600.0;
    if (q < 512.0) q = 512.0;
    return q;
}

// The streaming kernels hold a range's slice of the table in LDS as well, so the BUCKETS per query hash decide how many hashes
// a range can hold.  qindex_geometry gives between one and two (its callers look single hashes up in L2 and want short buckets):
// at 1.95 -- a 1.1e6-hash query -- a 10,240-bucket slice held 5,250 hashes instead of the 9,600 wanted and the overlap pass took
// 2.44 ms where the 1.0e6-hash query (1.07) takes 1.54.  When the wanted range does not fit the slice, the streaming kernels use a
// table of their own with buckets twice as wide.
void lean_table_geometry(uint64_t nq, uint64_t q_max, double mean_row, uint32_t* shift, uint32_t* buckets) {
    const double want = lean_hashes_per_range(nq, mean_row) * (double)*buckets / (double)nq;     // buckets a range would need
    if (want > 1.05 * (double)OwLean::BUCKETS && *shift < 63 && *buckets > 2) {
        ++*shift;
        *buckets = (uint32_t)(q_max >> *shift) + 1;
    }
}

LeanPlan build_lean_plan(uint64_t nq, uint32_t buckets, double mean_row) {
    uint32_t bpr = (uint32_t)(lean_hashes_per_range(nq, mean_row) * (double)buckets / (double)nq);
    if (bpr > (uint32_t)OwLean::BUCKETS) bpr = OwLean::BUCKETS;
    if (bpr < 64) bpr = 64;
    return LeanPlan{bpr, (buckets + bpr - 1) / bpr, (uint32_t)OwLean::QCAP, (uint32_t)OwLean::ROWS};
}

// what the lean kernel's staging form (MODE 2) needs besides the walk's arguments
struct RangeDesc { uint32_t b0, nb, p0, cnt; uint64_t upper; };   // first bucket, buckets, first query position, positions, first hash of the next range
struct StageArgs {
    const RangeDesc* desc = nullptr;      // [n_ranges]
    uint32_t* inter = nullptr;            // the postings, window by window and workgroup by workgroup, wherever their runs were placed
    uint32_t* dir_start = nullptr;        // [windows][n_sub] a run's start in inter ...
    uint32_t* dir_len = nullptr;          //                  ... and length
    uint32_t n_sub = 0;
    unsigned int* misc = nullptr;         // [0] words of inter given out, [1] != 0: a window's part of the staging area overflowed
};
constexpr int LEAN_NW = 32;               // windows per range at most (W <= 8,192 query positions)
constexpr int LEAN_CAPW = 352;            // postings a window's part of the staging area holds

// ---- the wide form, lean visits (round 4) -----------------------------------------------------------------------------
// Counters of the kernel above at C5 (profiles/r03_overlap_pmc.txt): 0.65e9 vector + 0.72e9 scalar + 0.13e9 LDS wave-instructions
// per pass for 13 million visits -- ~115 instructions a visit, of which the lookup proper needs ~35 -- and its waves spend 57 % of
// their cycles in s_waitcnt.  Twice the waves per SIMD (OwTwo) bought 3 %: it is the length of a visit's own dependent chain.
// Where the instructions and the waits went: a row's start, end and cursor lived in LDS and were read back through
// v_readfirstlane three to five times per visit (each a trip to LDS in FRONT of the load or the compare that needs it), hits
// went to LDS counters, the last range's special cases (2^64 - 1 as a hash, hashes above the query) were tested in every range.
// Here
//   * a slot's row state is two scalars that never leave registers: pos[k] / end[k], element offsets of the row's next
//     unconsumed hash and of its end -- the next load's address is scalar arithmetic on them;
//   * hits are a scalar per slot as well (population count of the found mask); the compiler parks scalars it has no
//     register for in lanes of a vector register (one v_readlane / v_writelane), which is far cheaper than an LDS counter;
//   * "which lanes hold a hash of this range" is ONE compare (e < upper; lanes past the row's end hold 2^64 - 1), "found" is
//     two compares: a bucket's first two query hashes are read whatever its size -- for a bucket of fewer the words behind it
//     belong to later buckets and cannot equal a hash that falls into this one;
//   * no clamp on the way to the query slice (two spare words behind it), no test against the largest query hash (hashes above
//     it fall into padding buckets); a query that holds 2^64 - 1 itself takes the kernel above.
// MODE 2 (STAGE): pass 1 AND pass 2a of the gather index build.  Every database hash's position in the query (or NONE32) goes to
// `qpos` -- a visit stores the positions of the hashes it consumes, consecutive lanes; counts[] is the builder's 64-bit overlap.  The ranges are cut by query POSITION (sa.desc: `W` positions each, a
// multiple of BR_SUB, so that the windows of BR_SUB lists the final scatter works by nest in them), the table slice of a range is
// clipped to the range's positions, and every posting found -- (row << BR_SUB_BITS) | list within its window -- waits in LDS in
// its window's part of a staging area until the range is through; then the workgroup reserves room for all of them in `inter`
// with ONE atomic, writes every window's run with consecutive lanes and leaves (start, length) in the directory
// dir[window][workgroup] the counting and scatter kernels read the runs by.  A window's part of the staging area holds
// LEAN_CAPW postings (six standard deviations above the ~250 a window gets from 400 rows of a C5-like database); more than that
// in any window raises sa.misc[1] and the host builds by the slower path instead.
template <class G, int MODE>
__global__ __launch_bounds__(OW_THREADS) __attribute__((amdgpu_waves_per_eu(G::WAVES_PER_EU, G::WAVES_PER_EU)))
void overlap_lean_kernel(const uint64_t* __restrict__ Q, const uint32_t* __restrict__ T, uint32_t n_buckets, uint32_t shift,
                         const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ offsets, uint64_t ndb,
                         uint32_t rows_per_wg, uint32_t n_ranges, uint32_t bpr, unsigned long long* __restrict__ counts,
                         uint32_t* __restrict__ qpos, uint64_t nq, StageArgs sa) {
    constexpr int SLOTS = G::SLOTS, BUCKETS = G::BUCKETS, QCAP = G::QCAP, BATCH = G::BATCH;
    static_assert(MODE == 0 || MODE == 2, "0: overlaps, 2: the builder's staging form");
    constexpr bool STAGE = MODE == 2, QPOS = MODE != 0;
    using TT = typename G::TT;
    extern __shared__ __attribute__((aligned(16))) uint64_t ow_lds[];
    uint64_t* s_q = ow_lds;                                              // [QCAP + 2]
    TT* s_t = reinterpret_cast<TT*>(s_q + QCAP + 2);                     // [BUCKETS + 4]
    uint32_t* s_stage = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(s_t) + G::T_BYTES);   // STAGE: [LEAN_NW][LEAN_CAPW] postings waiting, by window
    uint32_t* s_wcur = s_stage + LEAN_NW * LEAN_CAPW;                    //        [LEAN_NW] postings staged per window
    uint32_t* s_wbase = s_wcur + LEAN_NW;                                //        [LEAN_NW] where the window's run starts in `inter`
    uint32_t* s_wn = s_wbase + LEAN_NW;                                  //        [LEAN_NW] its length
    const uint64_t d_lo = (uint64_t)blockIdx.x * rows_per_wg;
    if (d_lo >= ndb) return;
    const uint32_t n_rows = (uint32_t)(ndb - d_lo < (uint64_t)rows_per_wg ? ndb - d_lo : (uint64_t)rows_per_wg);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t block_base = offsets[d_lo];
    const uint64_t* rows = hashes + block_base;
    uint32_t* const qrows = QPOS ? qpos + block_base : nullptr;
    if (STAGE && tid < LEAN_NW) s_wcur[tid] = 0;
    uint32_t pos[SLOTS], end[SLOTS], hv[SLOTS];
    uint64_t e[SLOTS];
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        const uint32_t i = (uint32_t)wave + (uint32_t)k * OW_WAVES;          // wave-uniform: the loads below are scalar
        pos[k] = end[k] = 0;
        hv[k] = 0;
        if (i < n_rows) {
            pos[k] = uniform32((uint32_t)(offsets[d_lo + i] - block_base));
            end[k] = uniform32((uint32_t)(offsets[d_lo + i + 1] - block_base));
        }
    }
    auto ask = [&](int k) {                                                // the next 64 hashes of slot k's row
        const uint32_t left = end[k] - pos[k];
        e[k] = ~0ull;
        if ((uint32_t)lane < left) e[k] = (rows + pos[k])[lane];
    };
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) ask(k);
    constexpr int QPER = (QCAP + OW_THREADS - 1) / OW_THREADS, TPER = (BUCKETS + 1 + OW_THREADS - 1) / OW_THREADS;
    uint32_t n_p0 = 0, n_p1 = 0;
    RangeDesc n_rd{};
    if (STAGE) n_rd = sa.desc[0];
    else { n_p0 = T[0]; n_p1 = T[bpr < n_buckets ? bpr : n_buckets]; }
    uint32_t done_p0 = 0, done_cnt = 0;                                   // STAGE: the slice whose postings are still in LDS
    // STAGE, after a barrier: every window's run gets its place in `inter` (one atomic for the workgroup) and its directory entry
    auto stage_place = [&]() {
        if (tid < 64) {
            const uint32_t nw = (done_cnt + (uint32_t)BR_SUB - 1) >> BR_SUB_BITS;        // windows of the range (<= LEAN_NW)
            uint32_t have = (uint32_t)lane < nw ? s_wcur[lane] : 0u;
            if (have > (uint32_t)LEAN_CAPW) { atomicOr(&sa.misc[1], 1u); have = LEAN_CAPW; }
            uint32_t incl = have;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d);
                if (lane >= d) incl += o;
            }
            const uint32_t total = __shfl(incl, 63);
            uint32_t base = 0;
            if (lane == 0 && total) base = atomicAdd(&sa.misc[0], total);
            base = __shfl(base, 0);
            if (lane < LEAN_NW) {
                s_wbase[lane] = base + incl - have;
                s_wn[lane] = have;
                s_wcur[lane] = 0;
            }
            if ((uint32_t)lane < nw) {
                const uint64_t di = (uint64_t)((done_p0 >> BR_SUB_BITS) + (uint32_t)lane) * sa.n_sub + blockIdx.x;
                sa.dir_start[di] = base + incl - have;
                sa.dir_len[di] = have;
            }
        }
    };
    auto stage_write = [&]() {                                             // after another barrier: the runs go out, consecutive lanes
        for (uint32_t i = tid; i < (uint32_t)(LEAN_NW * LEAN_CAPW); i += OW_THREADS) {
            const uint32_t wdw = i / (uint32_t)LEAN_CAPW, x = i - wdw * (uint32_t)LEAN_CAPW;
            if (x < s_wn[wdw]) sa.inter[s_wbase[wdw] + x] = s_stage[i];
        }
    };
    // a posting found: position jr of the range's slice, row `rowid`
    auto stage_put = [&](uint32_t jr, uint32_t rowid) {
        const uint32_t wdw = jr >> BR_SUB_BITS;
        const uint32_t slot = atomicAdd(&s_wcur[wdw], 1u);
        if (slot < (uint32_t)LEAN_CAPW) s_stage[wdw * (uint32_t)LEAN_CAPW + slot] = (rowid << BR_SUB_BITS) | (jr & (uint32_t)(BR_SUB - 1));
    };
    for (uint32_t r = 0; r < n_ranges; ++r) {
        const bool last = r + 1 == n_ranges;
        // b0: the slice's first bucket; kcap: buckets in the slice (a lane's bucket index is clamped to it: the padding behind)
        uint32_t b0, b1 = 0, kcap, p0, cnt_q, cnt_t;
        uint64_t upper;
        if (STAGE) {
            b0 = uniform32(n_rd.b0); kcap = uniform32(n_rd.nb); p0 = uniform32(n_rd.p0); cnt_q = uniform32(n_rd.cnt);
            upper = uniform64(n_rd.upper);
            cnt_t = kcap + 1;
        } else {
            b0 = r * bpr;
            b1 = b0 + bpr < n_buckets ? b0 + bpr : n_buckets;
            upper = last ? ~0ull : ((uint64_t)b1 << shift);
            p0 = uniform32(n_p0); cnt_q = uniform32(n_p1) - p0; cnt_t = b1 - b0 + 1;
            kcap = bpr;
        }
        __syncthreads();                                                  // the previous range's readers are done
        if (STAGE && r > 0) {
            stage_place();
            __syncthreads();
            stage_write();
        }
        if (QPOS) {
            done_p0 = p0;
            done_cnt = cnt_q;
        }
        {
            constexpr int FILL_STEP = 4;                                  // (registers: 25 slots x (2 + 1) stay live across the fill)
#pragma unroll
            for (int u0 = 0; u0 < TPER; u0 += FILL_STEP) {
                uint32_t tv[FILL_STEP];
#pragma unroll
                for (int u = 0; u < FILL_STEP; ++u) {
                    const uint32_t i = (uint32_t)tid + (uint32_t)(u0 + u) * OW_THREADS;
                    tv[u] = (u0 + u < TPER && i < cnt_t) ? T[b0 + i] : 0u;
                }
#pragma unroll
                for (int u = 0; u < FILL_STEP; ++u) {
                    const uint32_t i = (uint32_t)tid + (uint32_t)(u0 + u) * OW_THREADS;
                    // every slot of the slice is written, the ones behind the range's buckets with the slice's size
                    uint32_t v = tv[u] - p0;
                    if (STAGE) v = tv[u] < p0 ? 0u : (v < cnt_q ? v : cnt_q);  // clipped to the range's positions: buckets at its ends reach past them
                    if (u0 + u < TPER && i < (uint32_t)BUCKETS + 4u) s_t[i] = i < cnt_t ? (TT)v : (TT)cnt_q;
                }
            }
#pragma unroll
            for (int u0 = 0; u0 < QPER; u0 += FILL_STEP) {
                uint64_t qv[FILL_STEP];
#pragma unroll
                for (int u = 0; u < FILL_STEP; ++u) {
                    const uint32_t i = (uint32_t)tid + (uint32_t)(u0 + u) * OW_THREADS;
                    qv[u] = (u0 + u < QPER && i < cnt_q) ? Q[p0 + i] : 0ull;
                }
#pragma unroll
                for (int u = 0; u < FILL_STEP; ++u) {
                    const uint32_t i = (uint32_t)tid + (uint32_t)(u0 + u) * OW_THREADS;
                    if (u0 + u < QPER && i < cnt_q) s_q[i] = qv[u];
                }
            }
        }
        // padding buckets behind the table slice; two words of 2^64 - 1 behind the query slice: a lookup reads the two query
        // hashes at its bucket's start whatever the bucket holds, and scans on while they are below its hash -- the slice is
        // sorted, so what follows a bucket is larger than any hash that falls into it, and the padding ends every scan
        if (tid < 2) s_q[cnt_q + tid] = ~0ull;                            // (the table slice's padding is written by its fill)
        __syncthreads();
        if (!last) {                                                     // the next range's bounds
            if (STAGE) {
                n_rd = sa.desc[r + 1];
            } else {
                const uint32_t nb0 = b1, nb1 = nb0 + bpr < n_buckets ? nb0 + bpr : n_buckets;
                n_p0 = T[nb0];
                n_p1 = T[nb1];
            }
        }
#pragma unroll
        for (int v0 = 0; v0 < SLOTS; v0 += BATCH) {
            if ((uint32_t)wave + (uint32_t)v0 * OW_WAVES >= n_rows) break;  // no rows in this batch or behind it (wave-uniform)
            uint64_t in[BATCH];
            uint32_t t0[BATCH], t1[BATCH];
#pragma unroll
            for (int w = 0; w < BATCH; ++w) {
                const int k = v0 + w;
                if (k >= SLOTS) continue;
                in[w] = mask_of(e[k] < upper);                               // lanes past the row's end hold 2^64 - 1
                uint32_t kk = (uint32_t)(e[k] >> shift) - b0;                // < kcap for the lanes of `in`
                kk = kk < kcap ? kk : kcap;                                  // the others: the padding buckets behind the slice
                t0[w] = (uint32_t)s_t[kk];
                t1[w] = (uint32_t)s_t[kk + 1];
            }
            uint64_t qa[BATCH], qb[BATCH];
#pragma unroll
            for (int w = 0; w < BATCH; ++w) {
                if (v0 + w >= SLOTS) continue;
                qa[w] = s_q[t0[w]];
                qb[w] = s_q[t0[w] + 1];
            }
#pragma unroll
            for (int w = 0; w < BATCH; ++w) {
                const int k = v0 + w;
                if (k >= SLOTS) continue;
                const uint64_t second = mask_of(qb[w] == e[k]);
                uint64_t found = in[w] & (mask_of(qa[w] == e[k]) | second);
                uint32_t jr = 0;                                             // STAGE: position within the slice of the hash found
                if (QPOS) jr = t0[w] + (lanes_of(second) ? 1u : 0u);
                // a bucket of three or more whose second hash is still below the lane's hash (rare: ~1 visit in 3 has such a
                // lane, and a scan is a divergent loop over LDS): scan on.  (Without the size test every hash ABOVE both hashes of
                // a bucket of two came here too -- 3 % of the lookups, three visits in four: 2.46 -> 3.44 ms.)
                const uint64_t deep = mask_of(t1[w] > t0[w] + 2u) & mask_of(qb[w] < e[k]);
                if (__builtin_expect(deep != 0ull, 0)) {
                    bool hit = false;
                    if (lanes_of(deep))
                        for (uint32_t t = t0[w] + 2;; ++t) {
                            const uint64_t qv = s_q[t];
                            if (qv >= e[k]) { hit = qv == e[k]; if (QPOS && hit) jr = t; break; }
                        }
                    found |= in[w] & mask_of(hit);
                }
                if (QPOS) {
                    if (lanes_of(in[w])) (qrows + pos[k])[lane] = lanes_of(found) ? p0 + jr : NONE32;
                    if (STAGE && lanes_of(found)) stage_put(jr, (uint32_t)d_lo + (uint32_t)wave + (uint32_t)k * OW_WAVES);
                }
                hv[k] += (uint32_t)__popcll(found);
                uint32_t taken = (uint32_t)__popcll(in[w]);
                pos[k] += taken;
                while (__builtin_expect(taken == 64u, 0)) {                  // the row's part of this range goes on (rare): block by block
                    const uint32_t left = end[k] - pos[k];
                    const uint64_t ev = (uint32_t)lane < left ? (rows + pos[k])[lane] : ~0ull;
                    const bool more = ev < upper;
                    bool h2 = false;
                    if (more) {
                        uint32_t k2 = (uint32_t)(ev >> shift) - b0;
                        k2 = k2 < kcap ? k2 : kcap;
                        uint32_t t = (uint32_t)s_t[k2];
                        for (;; ++t) {
                            const uint64_t qv = s_q[t];
                            if (qv >= ev) { h2 = qv == ev; break; }
                        }
                        if (QPOS) {
                            (qrows + pos[k])[lane] = h2 ? p0 + t : NONE32;
                            if (STAGE && h2) stage_put(t, (uint32_t)d_lo + (uint32_t)wave + (uint32_t)k * OW_WAVES);
                        }
                    }
                    hv[k] += (uint32_t)__popcll(mask_of(h2));
                    taken = (uint32_t)__popcll(mask_of(more));
                    pos[k] += taken;
                }
                if (!last) ask(k);                                           // the row's part of the next range, a range ahead
            }
        }
    }
    if (QPOS) {
        __syncthreads();                                                  // the last range's adds are in
        if (STAGE) {
            stage_place();
            __syncthreads();
            stage_write();
        }
        // what the walk never consumed (a row's hash 2^64 - 1, which reads like the filler of lanes past a row's end): not in the query
#pragma unroll
        for (int k = 0; k < SLOTS; ++k)
            for (uint32_t i = pos[k] + (uint32_t)lane; i < end[k]; i += 64) qrows[i] = NONE32;
    }
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        const uint32_t i = (uint32_t)wave + (uint32_t)k * OW_WAVES;
        if (lane == 0 && i < n_rows) counts[d_lo + i] = hv[k];
    }
}

// The builder's pass 1 + 2a through the kernel above (MODE 2): ranges of `W` query positions described by range_plan_kernel
using OwStage = OwGeom<25, 9216, 8192, uint32_t, 4, 4>;
constexpr size_t LEAN_STAGE_LDS = ((size_t)OwStage::QCAP + 2) * 8 + OwStage::T_BYTES + ((size_t)LEAN_NW * LEAN_CAPW + 3 * LEAN_NW) * 4;
static_assert(LEAN_STAGE_LDS <= 160 * 1024, "");
static_assert(LEAN_NW * BR_SUB >= OwStage::QCAP, "a range's windows all have a part of the staging area");

__global__ __launch_bounds__(256) void range_plan_kernel(const uint64_t* __restrict__ Q, uint64_t nq, uint32_t shift, uint32_t n_buckets,
                                                         uint32_t W, uint32_t n_ranges, RangeDesc* __restrict__ desc, unsigned int* max_nb) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_ranges) return;
    const uint64_t p0 = (uint64_t)r * W;
    const bool last = r + 1 == n_ranges;
    RangeDesc d;
    d.p0 = (uint32_t)p0;
    d.cnt = (uint32_t)(nq - p0 < (uint64_t)W ? nq - p0 : (uint64_t)W);
    d.upper = last ? ~0ull : Q[p0 + W];                                  // hashes below it belong to this range or an earlier one
    d.b0 = r == 0 ? 0u : (uint32_t)(Q[p0] >> shift);                      // (what lies below the query's first hash is range 0's, and misses)
    const uint32_t b1 = last ? n_buckets - 1 : (uint32_t)(d.upper >> shift);
    d.nb = b1 - d.b0 + 1;
    desc[r] = d;
    atomicMax(max_nb, d.nb);
}

uint32_t build_stage_positions(uint64_t nq, uint32_t buckets, double mean_row) {   // W: query positions per range
    double w = 8900.0 * (double)nq / (double)buckets;                     // ~8,900 buckets per range, the slice has room for 9,216
    const double q = lean_hashes_per_range(nq, mean_row);                 // ... and a row's part of a range ~48 hashes
    if (w > q) w = q;
    if (w > (double)OwStage::QCAP) w = (double)OwStage::QCAP;
    const uint32_t W = ((uint32_t)w / (uint32_t)BR_SUB) * (uint32_t)BR_SUB;
    return W < (uint32_t)BR_SUB ? (uint32_t)BR_SUB : W;
}
uint32_t build_stage_buckets_max() { return (uint32_t)OwStage::BUCKETS; }
uint32_t build_stage_rows_max() { return (uint32_t)OwStage::ROWS; }
size_t build_stage_desc_bytes(uint32_t n_ranges) { return (size_t)n_ranges * sizeof(RangeDesc); }

hipError_t build_stage_plan(const uint64_t* Q, uint64_t nq, uint32_t shift, uint32_t n_buckets, uint32_t W, uint32_t n_ranges, void* desc,
                            unsigned int* max_nb, hipStream_t stream) {
    hipLaunchKernelGGL(range_plan_kernel, dim3((n_ranges + 255) / 256), dim3(256), 0, stream, Q, nq, shift, n_buckets, W, n_ranges,
                       (RangeDesc*)desc, max_nb);
    return hipGetLastError();
}

hipError_t build_stage_launch(const uint64_t* Q, uint64_t nq, const uint32_t* T, uint32_t n_buckets, uint32_t shift, const uint64_t* hashes,
                              const uint64_t* offsets, uint64_t ndb, uint32_t rows_per_wg, uint32_t n_ranges, const void* desc,
                              unsigned long long* counters, uint32_t* qpos, uint32_t* inter, uint32_t* dir_start, uint32_t* dir_len,
                              unsigned int* misc, hipStream_t stream) {
    static const int attr = [] {                                   // once, thread-safe (function-local static initialiser)
        const hipError_t ea = hipFuncSetAttribute((const void*)overlap_lean_kernel<OwStage, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LEAN_STAGE_LDS);
        if (ea == hipSuccess) return 1;
        (void)hipGetLastError();
        return -1;
    }();
    if (attr < 0) return hipErrorInvalidValue;
    const uint64_t n_sub = (ndb + rows_per_wg - 1) / rows_per_wg;
    StageArgs sa;
    sa.desc = (const RangeDesc*)desc;
    sa.inter = inter;
    sa.dir_start = dir_start;
    sa.dir_len = dir_len;
    sa.n_sub = (uint32_t)n_sub;
    sa.misc = misc;
    hipLaunchKernelGGL((overlap_lean_kernel<OwStage, 2>), dim3((unsigned)n_sub), dim3(OW_THREADS), LEAN_STAGE_LDS, stream, Q, T, n_buckets, shift,
                       hashes, offsets, ndb, rows_per_wg, n_ranges, 0u, counters, qpos, nq, sa);
    return hipGetLastError();
}

// op 0: overlap[d] = cnt[d]; op 1: overlap[d] -= cnt[d], saturating (rows at 0 stay dropped, index/__init__.py:908-909)
__global__ __launch_bounds__(256) void overlap_finish_kernel(const unsigned long long* __restrict__ cnt, uint64_t ndb,
                                                             unsigned long long* __restrict__ overlap, int op) {
    const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= ndb) return;
    const unsigned long long c = cnt[d];
    if (op == 0) overlap[d] = c;
    else overlap[d] = c >= overlap[d] ? 0 : overlap[d] - c;
}

// |Q ∩ row| for every row of a large collection against a large query: the streaming forms above.  The lean walk when the
// collection gives every CU a few dozen rows and every range of the query fits its LDS slice; the 16-lane streaming kernel
// otherwise; and when a range of the table holds more query hashes than either has room for (a query crowded into a sliver of
// the hash space), hipErrorNotSupported -- the caller (pair_ops.hip: overlap_vector_launch) then runs the one-wave-per-row kernel,
// which takes anything.  SMG_OVERLAP=wide|stream pins a form (a form that cannot run is then an error: tests), SMG_OVERLAP=rows
// never comes here.  Two stream synchronisations (the table geometry needs the largest query hash; the ranges' widths).
hipError_t overlap_ranges_launch(const uint64_t* Q, uint64_t nq, const uint64_t* hashes, const uint64_t* offsets, uint64_t ndb,
                                 unsigned long long* overlap, int op, hipStream_t stream) {
    if (nq == 0 || ndb == 0 || nq >= NONE32 || ndb >= NONE32) return hipErrorInvalidValue;
    struct Pinned {
        unsigned long long* p = nullptr;
        ~Pinned() { if (p) arena_pinned_free(p); }
    } pin;
    SMG_TRY(arena_pinned_alloc((void**)&pin.p, 64));
    SMG_TRY(hipMemcpyAsync(&pin.p[0], Q + nq - 1, 8, hipMemcpyDeviceToHost, stream));
    SMG_TRY(hipMemcpyAsync(&pin.p[3], offsets + ndb, 8, hipMemcpyDeviceToHost, stream));
    SMG_TRY(hipStreamSynchronize(stream));
    const uint64_t q_max = pin.p[0];
    const double mean_row = (double)pin.p[3] / (double)ndb;
    // The streaming kernels keep row cursors as 32-bit element offsets from their workgroup's first row: a collection of 2^32
    // hashes or more (32 GB) could put more than that under one workgroup (the index build makes the same check on pinned[0]).
    if (pin.p[3] >= 0xffffffffull) return hipErrorNotSupported;
    static const char* const form = getenv("SMG_OVERLAP");
    const bool only_wide = form && !strcmp(form, "wide"), only_stream = form && !strcmp(form, "stream");
    uint32_t shift = 0, buckets = 1;
    qindex_geometry(nq, q_max, &shift, &buckets);
    lean_table_geometry(nq, q_max, mean_row, &shift, &buckets);
    ArenaBuf table_b, cnt_b;
    SMG_TRY(table_b.get(((uint64_t)buckets + 1) * 4 + 64, stream));
    SMG_TRY(cnt_b.get(ndb * 8 + 64, stream));
    uint32_t* table = table_b.as<uint32_t>();
    unsigned long long* cnt = cnt_b.as<unsigned long long>();
    SMG_TRY(hipMemsetAsync(cnt, 0, ndb * 8 + 64, stream));
    SMG_TRY(qtable_launch(Q, nq, shift, buckets, table, stream));
    int n_cu = 256;
    { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); }
    const bool try_lean = !only_stream && (only_wide || ndb >= (uint64_t)n_cu * 64) && q_max != ~0ull;   // (2^64 - 1 in the query: the lean
    uint32_t bpr = SL_BUCKETS;                                             //  kernel's filler value would be a hit)
    while (bpr > 64 && (double)bpr * (double)nq / (double)buckets > 2200.0) bpr >>= 1;      // the 16-lane form: about 2,000 query hashes per range
    const uint32_t n_ranges = (buckets + bpr - 1) / bpr;
    const LeanPlan lean = build_lean_plan(nq, buckets, mean_row);                           // the lean form: ranges cut by query hashes held
    // the widest range of either partition decides whether its LDS has room: both maxima come back with one synchronisation
    unsigned int* d_widest = (unsigned int*)(cnt + ndb);                   // the 64 spare bytes, zeroed above
    if (!only_wide) SMG_TRY(stream_range_max_launch(table, buckets, n_ranges, bpr, d_widest, stream));
    if (try_lean) SMG_TRY(stream_range_max_launch(table, buckets, lean.n_ranges, lean.bpr, d_widest + 1, stream));
    SMG_TRY(hipMemcpyAsync(&pin.p[1], d_widest, 8, hipMemcpyDeviceToHost, stream));
    SMG_TRY(hipStreamSynchronize(stream));
    const unsigned int widest = (unsigned int)(pin.p[1] & 0xffffffffull), l_widest = (unsigned int)(pin.p[1] >> 32);
    constexpr size_t LEAN_LDS = ((size_t)OwLean::QCAP + 2) * 8 + OwLean::T_BYTES;
    static const int lean_attr = [] {                              // once, thread-safe (function-local static initialiser)
        const hipError_t ea = hipFuncSetAttribute((const void*)overlap_lean_kernel<OwLean, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LEAN_LDS);
        if (ea == hipSuccess) return 1;
        (void)hipGetLastError();
        return -1;
    }();
    if (try_lean && lean_attr > 0 && l_widest > 0 && l_widest <= (unsigned)OwLean::QCAP) {
        // rows a workgroup owns: every workgroup resident at once, in full rounds (a last round of a few workgroups would cost a
        // whole pass over the query for a fraction of the rows); SMG_OVERLAP_ROWS overrides (tuning / tests)
        static const uint64_t rpw_env = [] { const char* e = getenv("SMG_OVERLAP_ROWS"); return e ? (uint64_t)atoll(e) : 0ull; }();
        const uint64_t resident = (uint64_t)n_cu, cap = OwLean::ROWS;
        uint64_t rpw = (ndb + resident - 1) / resident;
        if (rpw > cap) {
            const uint64_t per_round = cap * resident;
            const uint64_t slots = ((ndb + per_round - 1) / per_round) * resident;
            rpw = (ndb + slots - 1) / slots;
        }
        if (rpw_env) rpw = rpw_env;
        if (rpw > cap) rpw = cap;
        if (rpw < 1) rpw = 1;
        const uint64_t n_wg = (ndb + rpw - 1) / rpw;
        hipLaunchKernelGGL((overlap_lean_kernel<OwLean, 0>), dim3((unsigned)n_wg), dim3(OW_THREADS), LEAN_LDS, stream, Q, (const uint32_t*)table, buckets,
                           shift, hashes, offsets, ndb, (uint32_t)rpw, lean.n_ranges, lean.bpr, cnt, (uint32_t*)nullptr, (uint64_t)0, StageArgs{});
    } else if (only_wide) {
        return hipErrorInvalidValue;
    } else if (widest <= (unsigned)SL_QCAP) {
        const uint32_t n_blocks = (uint32_t)((ndb + SL_ROWS - 1) / SL_ROWS);
        // every workgroup resident at once (4 per CU by LDS and waves): with even a few more than fit, the kernel takes two
        // rounds -- 784 workgroups on 768 slots ran 4.2 ms with the CUs idle 42 % of the wave-time (profiles/r02_gather_sq.txt)
        const uint32_t slots = (uint32_t)n_cu * 4u;
        uint32_t n_groups = slots / n_blocks;                                // floor: never one workgroup more than fits
        if (n_groups > n_ranges) n_groups = n_ranges;
        if (n_groups < 1) n_groups = 1;
        const uint32_t per = (n_ranges + n_groups - 1) / n_groups;
        n_groups = (n_ranges + per - 1) / per;
        hipLaunchKernelGGL(stream_lookup_kernel, dim3(n_blocks * n_groups), dim3(SL_THREADS), 0, stream, Q, (const uint32_t*)table, buckets,
                           shift, q_max, hashes, offsets, ndb, n_blocks, n_ranges, per, bpr, cnt);
    } else {
        return only_stream ? hipErrorInvalidValue : hipErrorNotSupported;
    }
    hipLaunchKernelGGL(overlap_finish_kernel, dim3((unsigned)((ndb + 255) / 256)), dim3(256), 0, stream, cnt, ndb, overlap, op);
    return hipGetLastError();
}

}  // namespace smg
