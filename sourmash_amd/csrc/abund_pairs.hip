// abund_pairs.hip -- all pairs of abundance-tracking sketches through lists sorted by hash (round 5).
//
// What is computed: prod[i][j] = sum over the hashes common to sketches i and j of abund_i * abund_j (u64, wrapping like the
// reference's release build), common[i][j] = the number of those hashes -- the integer part of angular_similarity,
// src/core/src/sketch/minhash.rs:635-680, for every pair of a collection (src/sourmash/compare.py:14-64 walks the pairs).
//
// Rounds 3-4 ran the reference's two-pointer walk for every pair (compare_ext.hip, one lane per pair over LDS-staged segments):
// 5 x 10^5 pairs x 10^4 steps at config C3's shape, 5.5 ms, a dependent LDS chain per step.  The work that has to be done is the
// MATCHES: 2.5 x 10^8 products at C3 (10 % of every pair's hashes are shared), a twentieth of the walk's steps.  Here the
// collection is turned once into per-block lists -- the elements of 64 consecutive sketches sorted by hash, each with its row within
// the block and its abundance (round 6: the sorted rows merged slice by slice in LDS, below) -- and an output tile (block bi x block bj)
// is the merge-join of two such lists: equal hashes meet, and every (row of bi, row of bj) pair of a met hash adds one product to
// the tile's 64 x 64 accumulators in LDS.  A tile's join is cut into Z hash slices when there are few tiles (C3: 136 tiles on
// 256 CUs), the slices' sums meeting in the zeroed matrices through atomics.
//
// The join, per workgroup: the next <= 4,096 entries of list B go to LDS (whole runs of equal hashes only); every entry of list A up
// to B's last staged hash is taken by one thread, which finds its hash in the staged B entries by a fixed-step search (12 LDS reads,
// independent of its neighbours': 16 waves per CU overlap them) and the end of the run of equal hashes there (6 more); short runs
// it adds itself, longer ones are shared out over the workgroup through a worklist in LDS (see the kernel).  The next chunk of B
// and the next batch of A are loaded into registers while the present ones are matched.  Work grows with |A| + |B| + matches per
// tile, not with pairs x lengths.  What bounds it is instruction issue and the dependent LDS reads of the searches, not the LDS
// atomics (measured: profiles/r05_abund_join_experiments.txt, tools/ubench/lds_atomic.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "arena.hpp"
#include "device_api.hpp"

namespace smg {

namespace {

constexpr int AP_T = 64;                 // sketches per block = tile edge
constexpr int AP_TS = AP_T + 1;          // row stride of the tile's accumulators in LDS: a hash shared by a whole block puts the 64
                                         // lanes of a wave on 64 different rows and ONE column -- at a stride of 64 that is one bank
constexpr int AP_THREADS = 1024;
constexpr int AP_CHUNK = 4096;           // entries of list B staged per round (64 KB of hashes + payloads)
constexpr int AP_ZMAX = 16;              // hash slices per tile at most
#ifndef AP_INLINE_N
#define AP_INLINE_N 16
#endif
constexpr int AP_INLINE = AP_INLINE_N;            // an entry of list A that meets at most this many entries of B adds its products itself
constexpr int AP_G = 16;                 // lanes that share a longer run

#define AP_TRY(expr)                       \
    do {                                   \
        hipError_t e_ = (expr);            \
        if (e_ != hipSuccess) return e_;   \
    } while (0)

// ---- the lists: a block's 64 sorted rows merged, hash slice by hash slice, in LDS (round 6) ---------------------------------
// Round 5 built the lists with two device radix sorts (by hash, then by block: nine passes over every element, 0.7 ms of the call's
// 1.9 at C3's shape) although every row arrives sorted.  Now the hash space is cut into `n_slices` equal slices (a shift of the
// hash: about a thousand entries of a block fall into one), `cuts[row][s]` = where slice s starts in the row (one binary search per
// row and slice), and one workgroup merges a (block, slice): its place in the block's list is known from the cuts alone (the
// entries of the block's rows below the slice), the <= SM_CAP entries go to LDS grouped by SM_BUCKETS order-preserving sub-buckets
// of the slice (count, prefix, place), and an entry's final position is its bucket's start + the number of the bucket's entries with
// a smaller (hash, row) -- a handful of compares, no sort.  A slice with more entries than the LDS room (hashes far from uniform)
// is ranked against the rows' parts in memory instead: slow, same result.
constexpr int SM_CAP = 2048;             // entries of a (block, slice) merged in LDS
constexpr int SM_BUCKETS = 2048;         // sub-buckets of a slice
constexpr int SM_BUCKET_BITS = 11;
constexpr int SM_THREADS = 256;
constexpr int SM_TARGET = 640;           // entries per (block, slice) aimed at; the shift lands between this and twice this

struct ApPlan { uint32_t shift, n_slices; };

__global__ __launch_bounds__(1024) void ap_plan_kernel(const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ offsets, uint32_t n,
                                                       uint32_t s_max, ApPlan* __restrict__ plan) {
    __shared__ unsigned long long s_m[16];
    unsigned long long m = 0;
    for (uint32_t r = threadIdx.x; r < n; r += 1024) {
        const uint64_t o0 = offsets[r], o1 = offsets[r + 1];
        if (o1 > o0) { const unsigned long long last = hashes[o1 - 1]; m = last > m ? last : m; }      // rows are sorted
    }
#pragma unroll
    for (int d = 32; d; d >>= 1) { const unsigned long long o = __shfl_xor(m, d); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) m = s_m[w] > m ? s_m[w] : m;
        uint32_t shift = 0;
        while ((m >> shift) >= (unsigned long long)s_max) ++shift;             // s_max >= 2: ends at 63 at the latest
        plan->shift = shift;
        plan->n_slices = (uint32_t)(m >> shift) + 1u;                           // <= s_max
    }
}

// cuts[r][s], s = 0 .. s_max: the first entry of row r that belongs to slice s or a later one (relative to the row's start)
__global__ __launch_bounds__(256) void ap_cuts_kernel(const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ offsets, uint32_t n,
                                                      uint32_t s_max, const ApPlan* __restrict__ plan, uint32_t* __restrict__ cuts) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t r = (uint32_t)(idx / (s_max + 1u)), s = (uint32_t)(idx % (s_max + 1u));
    if (r >= n) return;
    const uint32_t shift = plan->shift, ns = plan->n_slices;
    const uint64_t o0 = offsets[r];
    const uint32_t len = (uint32_t)(offsets[r + 1] - o0);
    uint32_t lo = 0, hi = len;
    if (s >= ns) lo = len;
    else if (s > 0) {
        const uint64_t x = (uint64_t)s << shift;                               // <= the largest hash: no overflow
        const uint64_t* row = hashes + o0;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (row[mid] < x) lo = mid + 1; else hi = mid;
        }
    }
    cuts[idx] = lo;
}

template <bool NARROW>
__global__ __launch_bounds__(SM_THREADS) void ap_slice_kernel(const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ abunds,
                                                              const uint64_t* __restrict__ offsets, uint32_t n,
                                                              const uint32_t* __restrict__ cuts, uint32_t s_max,
                                                              const ApPlan* __restrict__ plan, uint64_t* __restrict__ l_hash,
                                                              uint64_t* __restrict__ l_pay, uint8_t* __restrict__ l_row) {
    __shared__ uint64_t t_h[SM_CAP], t_v[SM_CAP];
    __shared__ uint32_t s_cnt[SM_BUCKETS], s_start[SM_BUCKETS + 1];
    __shared__ uint8_t t_r[SM_CAP];
    __shared__ uint32_t s_c0[AP_T], s_len[AP_T], s_wsum[SM_THREADS / 64];
    __shared__ uint32_t s_total;
    __shared__ uint64_t s_out0;
    const uint32_t ns = plan->n_slices, shift = plan->shift;
    const uint32_t s = blockIdx.x % s_max, b = blockIdx.x / s_max;
    if (s >= ns) return;
    const int tid = threadIdx.x;
    const uint32_t r0 = b * AP_T, nr = n - r0 < (uint32_t)AP_T ? n - r0 : (uint32_t)AP_T;
    if (tid < 64) {
        uint32_t c0 = 0, len = 0;
        if ((uint32_t)tid < nr) {
            const uint64_t at = (uint64_t)(r0 + tid) * (s_max + 1u) + s;
            c0 = cuts[at];
            len = cuts[at + 1] - c0;
        }
        s_c0[tid] = c0;
        s_len[tid] = len;
        unsigned long long below = c0;                                       // the block's entries below the slice: where it starts in the list
        uint32_t tot = len;
#pragma unroll
        for (int d = 32; d; d >>= 1) { below += __shfl_xor(below, d); tot += __shfl_xor(tot, d); }
        if (tid == 0) { s_total = tot; s_out0 = offsets[r0] + below; }
    }
    for (int i = tid; i < SM_BUCKETS; i += SM_THREADS) s_cnt[i] = 0;
    __syncthreads();
    const uint32_t total = s_total;
    if (total == 0) return;
    const uint64_t out0 = s_out0;
    const uint32_t rr = (uint32_t)tid >> 2, q = (uint32_t)tid & 3u;          // four lanes walk a row's part of the slice
    const uint32_t my_len = s_len[rr];
    const uint64_t my_at = rr < nr ? offsets[r0 + rr] + s_c0[rr] : 0;
    const uint64_t* row = hashes + my_at;
    const uint64_t* arow = abunds + my_at;
    const uint64_t base = (uint64_t)s << shift;
    auto bucket = [&](uint64_t h) { const uint64_t rel = h - base; return (uint32_t)(shift > (uint32_t)SM_BUCKET_BITS ? rel >> (shift - SM_BUCKET_BITS) : rel); };
    auto put = [&](uint64_t o, uint64_t h, uint64_t a, uint32_t r) {
        l_hash[o] = h;
        if (NARROW) l_pay[o] = ((uint64_t)r << 32) | (uint32_t)a;            // row and 32-bit abundance in one word
        else { l_pay[o] = a; l_row[o] = (uint8_t)r; }
    };
    if (total <= (uint32_t)SM_CAP) {
        for (uint32_t i = q; i < my_len; i += 4) atomicAdd(&s_cnt[bucket(row[i])], 1u);
        __syncthreads();
        {                                                                    // exclusive prefix of the bucket counts; the counts become cursors
            constexpr int PER = SM_BUCKETS / SM_THREADS;
            uint32_t c[PER], sum = 0;
#pragma unroll
            for (int u = 0; u < PER; ++u) { c[u] = s_cnt[tid * PER + u]; sum += c[u]; }
            uint32_t incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if ((tid & 63) >= d) incl += o; }
            if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
            __syncthreads();
            uint32_t run = incl - sum;
            for (int w = 0; w < (tid >> 6); ++w) run += s_wsum[w];
#pragma unroll
            for (int u = 0; u < PER; ++u) { s_start[tid * PER + u] = run; s_cnt[tid * PER + u] = 0; run += c[u]; }
            if (tid == SM_THREADS - 1) s_start[SM_BUCKETS] = run;
        }
        __syncthreads();
        for (uint32_t i = q; i < my_len; i += 4) {
            const uint64_t h = row[i];
            const uint32_t k = bucket(h);
            const uint32_t p = s_start[k] + atomicAdd(&s_cnt[k], 1u);
            t_h[p] = h; t_v[p] = arow[i]; t_r[p] = (uint8_t)rr;
        }
        __syncthreads();
        for (uint32_t p = tid; p < total; p += SM_THREADS) {
            const uint64_t h = t_h[p];
            const uint32_t r = t_r[p], k = bucket(h);
            const uint32_t b0 = s_start[k], b1 = s_start[k + 1];
            uint32_t rank = 0;
            for (uint32_t j = b0; j < b1; ++j) {
                const uint64_t hj = t_h[j];
                rank += (hj < h || (hj == h && (uint32_t)t_r[j] < r)) ? 1u : 0u;
            }
            put(out0 + b0 + rank, h, t_v[p], r);
        }
    } else {
        // more entries than the LDS holds: an entry's place = the entries of the other rows' parts in front of it (equal hashes: the
        // earlier rows') + its own index
        for (uint32_t i = q; i < my_len; i += 4) {
            const uint64_t h = row[i];
            uint64_t rank = i;
            for (uint32_t o = 0; o < nr; ++o) {
                if (o == rr) continue;
                const uint64_t* other = hashes + offsets[r0 + o] + s_c0[o];
                uint32_t lo = 0, hi = s_len[o];
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    const uint64_t v = other[mid];
                    if (v < h || (v == h && o < rr)) lo = mid + 1; else hi = mid;
                }
                rank += lo;
            }
            put(out0 + rank, h, arow[i], rr);
        }
    }
}

__device__ __forceinline__ uint64_t ap_lower_bound(const uint64_t* __restrict__ a, uint64_t lo, uint64_t hi, uint64_t x) {
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// one workgroup per (tile, slice): tile t = (bi, bj), bi <= bj, enumerated row by row of the upper triangle
template <bool NARROW>
__global__ __launch_bounds__(AP_THREADS) void ap_join_kernel(const uint64_t* __restrict__ l_hash, const uint64_t* __restrict__ l_pay,
                                                             const uint8_t* __restrict__ l_row, const uint64_t* __restrict__ offsets,
                                                             uint32_t n, uint32_t nb, uint32_t Z, uint32_t* __restrict__ common,
                                                             unsigned long long* __restrict__ prod) {
    extern __shared__ __attribute__((aligned(16))) uint64_t ap_lds[];      // AP_LDS bytes (more than the 64 KB a static array may take)
    uint64_t* s_bh = ap_lds;                                         // [AP_CHUNK] staged hashes of list B
    uint64_t* s_bp = s_bh + AP_CHUNK;                                // [AP_CHUNK] their payloads
    unsigned long long* s_prod = reinterpret_cast<unsigned long long*>(s_bp + AP_CHUNK);   // [AP_T * AP_TS]
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_prod + AP_T * AP_TS);                  // [AP_T * AP_TS]
    unsigned long long* s_wpay = reinterpret_cast<unsigned long long*>(s_cnt + AP_T * AP_TS);   // [2][AP_THREADS] worklists: abundance (+ row) of A
    uint32_t* s_work = reinterpret_cast<uint32_t*>(s_wpay + 2 * AP_THREADS);                // [2][AP_THREADS] run start | length << 12 | row << 19
    uint8_t* s_br = reinterpret_cast<uint8_t*>(s_work + 2 * AP_THREADS);                    // [AP_CHUNK] rows of the staged entries (wide abundances)
    __shared__ uint32_t s_wn[2];                                     // entries ever appended to the two worklists (never reset)
    __shared__ uint64_t s_bound[4];                                  // this slice's ranges in the two lists
    const int tid = threadIdx.x;
    const uint32_t z = blockIdx.x % Z;
    uint32_t t = blockIdx.x / Z, bi = 0;
    while (t >= nb - bi) { t -= nb - bi; ++bi; }                     // (nb is small: at most a few hundred steps)
    const uint32_t bj = bi + t;
    const bool diag = bi == bj;
    const uint32_t row0 = bi * AP_T, col0 = bj * AP_T;
    const uint32_t ra_end = row0 + AP_T < n ? row0 + AP_T : n, rb_end = col0 + AP_T < n ? col0 + AP_T : n;
    const uint64_t A0 = offsets[row0], A1 = offsets[ra_end], B0 = offsets[col0], B1 = offsets[rb_end];
    for (int i = tid; i < AP_T * AP_TS; i += AP_THREADS) { s_prod[i] = 0; s_cnt[i] = 0; }
    if (tid < 2) s_wn[tid] = 0;
    if (tid == 0) {
        // slice z of the tile: an equal share of list A's entries, moved to the start of a run of equal hashes; list B between the
        // same two hash values
        uint64_t a0 = A0 + (A1 - A0) * z / Z, a1 = z + 1 == Z ? A1 : A0 + (A1 - A0) * (z + 1) / Z;
        if (a0 > A0 && a0 < A1) a0 = ap_lower_bound(l_hash, A0, a0, l_hash[a0]);
        if (a1 > A0 && a1 < A1) a1 = ap_lower_bound(l_hash, A0, a1, l_hash[a1]);
        uint64_t b0 = B0, b1 = B1;
        if (a0 < a1) {
            if (a0 > A0) b0 = ap_lower_bound(l_hash, B0, B1, l_hash[a0]);
            if (a1 < A1) b1 = ap_lower_bound(l_hash, b0, B1, l_hash[a1]);
        } else b1 = b0;
        s_bound[0] = a0; s_bound[1] = a1; s_bound[2] = b0; s_bound[3] = b1;
    }
    __syncthreads();
    uint64_t a_cur = s_bound[0];
    const uint64_t a_end = s_bound[1];
    uint64_t b_cur = s_bound[2];
    const uint64_t b_end = s_bound[3];
    // Round structure: chunk r of list B is in LDS; chunk r + 1 is already on its way into registers (its start is known as soon as
    // chunk r has been trimmed to whole runs); the batch of list A being matched was loaded while the previous batch was searched.
    // What a round waits for is then one LDS write + barrier and the first A batch -- not a trip to memory per step.
    //
    // Matching is in two steps.  A hash held by most sketches of both blocks is a run of up to 64 consecutive entries in A -- one
    // wave's lanes -- each meeting a run of up to 64 in B: walked by the owning lanes, that is 64 dependent steps in one wave while
    // the other fifteen wait at the batch's barrier.  So a lane only FINDS its run (lower bound over the chunk, upper bound over the
    // next 64 entries); runs of up to AP_INLINE it adds itself, longer ones go on a worklist in LDS, and after the batch's barrier
    // the whole workgroup takes the list, 16 lanes to a run.  The lists alternate between two buffers and their counters only grow,
    // so no barrier is needed between one batch's list walk and the next batch's appends.  (Measured at C3, where a hash sits in 6
    // of a block's 64 sketches and nothing is long: sending every run of more than two through the list costs 1.17 -> 1.6 ms; the
    // list is for collections with a core of shared hashes.)
    constexpr int BPT = AP_CHUNK / AP_THREADS;                       // staged entries per thread
    uint32_t wbase0 = 0, wbase1 = 0, par = 0;                        // list counters at the start of their present use; the batch's parity
    uint64_t rh[BPT], rp[BPT];
    uint8_t rr[BPT];
    auto fetch_b = [&](uint64_t from) {                              // entries [from, from + AP_CHUNK) of list B -> registers
#pragma unroll
        for (int u = 0; u < BPT; ++u) {
            const uint64_t i = from + (uint64_t)tid + (uint64_t)u * AP_THREADS;
            rh[u] = ~0ull; rp[u] = 0; rr[u] = 0;
            if (i < b_end) {
                rh[u] = l_hash[i];
                rp[u] = l_pay[i];
                if (!NARROW) rr[u] = l_row[i];
            }
        }
    };
    if (a_cur < a_end && b_cur < b_end) fetch_b(b_cur);
    while (a_cur < a_end && b_cur < b_end) {
        // ---- stage the fetched entries of B; whole runs of equal hashes only are used (a run holds at most AP_T entries) ----
        uint64_t nb_stage = b_end - b_cur < (uint64_t)AP_CHUNK ? b_end - b_cur : (uint64_t)AP_CHUNK;
#pragma unroll
        for (int u = 0; u < BPT; ++u) {
            const uint32_t i = (uint32_t)tid + (uint32_t)u * AP_THREADS;
            s_bh[i] = rh[u];
            s_bp[i] = rp[u];
            if (!NARROW) s_br[i] = rr[u];
        }
        __syncthreads();
        if (b_cur + nb_stage < b_end) {                              // the run the chunk ends in may go on: leave it for the next round
            const uint64_t last = s_bh[nb_stage - 1];
            uint64_t cut = nb_stage - 1;
            while (cut > 0 && s_bh[cut - 1] == last) --cut;           // (every thread walks the same <= AP_T broadcast reads)
            nb_stage = cut;                                           // > 0: a run is shorter than the chunk
        }
        const uint64_t hi = s_bh[nb_stage - 1];                       // A's entries up to this hash meet everything they can meet here
        if (b_cur + nb_stage < b_end) fetch_b(b_cur + nb_stage);      // the next round's entries: in flight behind this round's work
        // ---- A's entries <= hi, a batch of AP_THREADS at a time; the batch behind the one being matched is loaded meanwhile ----
        uint64_t nh = ~0ull, npay = 0;
        uint8_t nrow = 0;
        auto fetch_a = [&](uint64_t from) {
            const uint64_t idx = from + (uint64_t)tid;
            nh = ~0ull; npay = 0; nrow = 0;
            if (idx < a_end) {
                nh = l_hash[idx];
                npay = l_pay[idx];
                if (!NARROW) nrow = l_row[idx];
            }
        };
        fetch_a(a_cur);
        for (;;) {
            const uint64_t h = nh, pay = npay;
            const uint32_t ra = NARROW ? (uint32_t)(pay >> 32) : (uint32_t)nrow;
            const bool mine = h <= hi && a_cur + (uint64_t)tid < a_end;
            fetch_a(a_cur + AP_THREADS);                              // used only if this whole batch is taken
            const uint32_t wbase = par ? wbase1 : wbase0;
            auto add = [&](uint32_t row_a, unsigned long long aa, uint32_t j) {
                const uint64_t pb = s_bp[j];
                const uint32_t rb = NARROW ? (uint32_t)(pb >> 32) : (uint32_t)s_br[j];
                if (diag && rb <= row_a) return;                     // a diagonal tile joins a list with itself: every pair once
                const unsigned long long ab = NARROW ? (unsigned long long)(uint32_t)pb : (unsigned long long)pb;
                atomicAdd(&s_prod[row_a * AP_TS + rb], aa * ab);
                atomicAdd(&s_cnt[row_a * AP_TS + rb], 1u);
            };
            // Both searches are written without loops over lane-dependent bounds: twelve and six fixed steps, every lane of the wave
            // in every step (a `while (lo < hi)` costs its exec-mask bookkeeping in each of its ~18 steps: the kernel is bound by
            // instruction issue and by these dependent LDS reads, profiles/r05_abund_join_experiments.txt).  The whole staging area is
            // searched, not just its first nb_stage entries: what lies behind them is the cut-off run of the largest staged hash and
            // 2^64 - 1 fillers, and no entry of A taken in this round (h <= hi) can equal either.
            uint32_t lo = 0;
#pragma unroll
            for (uint32_t len = AP_CHUNK / 2; len; len >>= 1) lo += s_bh[lo + len - 1] < h ? len : 0u;
            if (mine) {
                if (lo < (uint32_t)nb_stage && s_bh[lo] == h) {
                    uint32_t cnt = 1;                                 // the run of h: entries lo .. lo + cnt - 1, cnt <= 64
#pragma unroll
                    for (uint32_t len = AP_T / 2; len; len >>= 1) {
                        const uint32_t at = lo + cnt + len - 1;
                        if (at < (uint32_t)nb_stage && s_bh[at] == h) cnt += len;
                    }
                    const uint32_t ulo = lo + cnt;
                    const unsigned long long aa = NARROW ? (unsigned long long)(uint32_t)pay : (unsigned long long)pay;
                    if (cnt <= (uint32_t)AP_INLINE) {
                        for (uint32_t j = lo; j < ulo; ++j) add(ra, aa, j);
                    } else {
                        const uint32_t slot = atomicAdd(&s_wn[par], 1u) - wbase;      // < AP_THREADS: one per thread at most
                        s_work[par * AP_THREADS + slot] = lo | (cnt << 12) | (ra << 19);
                        s_wpay[par * AP_THREADS + slot] = aa;
                    }
                }
            }
            const int took = __syncthreads_count(mine ? 1 : 0);       // sorted: the entries taken are a prefix of the batch
            {
                const uint32_t wend = s_wn[par];                      // (nobody appends to this list again before the next barrier)
                const uint32_t nw = wend - wbase;
                if (par) wbase1 = wend; else wbase0 = wend;
                // (a wave's four groups take entries 16 apart: neighbours on the list are neighbouring rows of one hash, whose
                //  accumulator rows start two banks apart -- four groups on the same 32 banks)
                const uint32_t q = (uint32_t)tid / AP_G;
                for (uint32_t w = (q % 4u) * 16u + q / 4u; w < nw; w += AP_THREADS / AP_G) {
                    const uint32_t e = s_work[par * AP_THREADS + w];
                    const unsigned long long aa = s_wpay[par * AP_THREADS + w];
                    const uint32_t lo = e & 4095u, cnt = (e >> 12) & 127u, row_a = e >> 19;
                    for (uint32_t o = (uint32_t)tid % AP_G; o < cnt; o += AP_G) add(row_a, aa, lo + o);
                }
                par ^= 1u;
            }
            a_cur += (uint64_t)took;
            if (took < AP_THREADS) break;
        }
        b_cur += nb_stage;
        __syncthreads();                                             // the staged entries are about to be overwritten
    }
    __syncthreads();
    // ---- the tile's sums go out; the mirrored entries with them (the matrices are symmetric) ----
    for (int i = tid; i < AP_T * AP_T; i += AP_THREADS) {
        const uint32_t r = row0 + (uint32_t)i / AP_T, c = col0 + (uint32_t)i % AP_T;
        if (r >= n || c >= n || (diag && c <= r)) continue;
        const unsigned long long p = s_prod[((uint32_t)i / AP_T) * AP_TS + (uint32_t)i % AP_T];
        const uint32_t k = s_cnt[((uint32_t)i / AP_T) * AP_TS + (uint32_t)i % AP_T];
        if (Z == 1) {
            prod[(uint64_t)r * n + c] = p; prod[(uint64_t)c * n + r] = p;
            common[(uint64_t)r * n + c] = k; common[(uint64_t)c * n + r] = k;
        } else if (k) {
            atomicAdd(&prod[(uint64_t)r * n + c], p); atomicAdd(&prod[(uint64_t)c * n + r], p);
            atomicAdd(&common[(uint64_t)r * n + c], k); atomicAdd(&common[(uint64_t)c * n + r], k);
        }
    }
}

constexpr size_t AP_LDS = (size_t)AP_CHUNK * 16 + (size_t)AP_T * AP_TS * 12 + (size_t)AP_THREADS * 2 * 12 + AP_CHUNK;
static_assert(AP_CHUNK <= 4096 && AP_T <= 64, "a worklist word holds a 12-bit run start, a 7-bit length and a 6-bit row");
static_assert((AP_CHUNK & (AP_CHUNK - 1)) == 0 && AP_T == 64, "the searches halve a power of two");
static_assert(AP_THREADS == 1024 && AP_G == 16, "the list walk deals 64 groups of 16 lanes, four to a wave");

}  // namespace

// -> hipErrorNotSupported when this formulation does not apply (2^32 elements or more): the caller keeps the walk kernel
hipError_t abund_pairs_launch(const uint64_t* d_hashes, const uint64_t* d_abunds, const uint64_t* d_offsets, uint32_t n, uint64_t total,
                              bool narrow, uint32_t* d_common, unsigned long long* d_prod, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (total >= 0xffffffffull) return hipErrorNotSupported;
    const uint32_t nb = (n + AP_T - 1) / AP_T;
    const uint64_t tiles = (uint64_t)nb * (nb + 1) / 2;
    if (tiles * AP_ZMAX > 0x7fffffffull) return hipErrorNotSupported;
    AP_TRY(hipMemsetAsync(d_common, 0, (size_t)n * n * 4, stream));       // (pairs with nothing in common stay zero; slices add up)
    AP_TRY(hipMemsetAsync(d_prod, 0, (size_t)n * n * 8, stream));
    if (total == 0) return hipSuccess;
    // the per-block lists: plan (shift of the slices from the largest hash), cuts, one workgroup per (block, slice)
    uint64_t per_block = total / nb / (uint64_t)SM_TARGET;
    const uint32_t s_max = (uint32_t)(per_block < 2 ? 2 : (per_block > 0x100000ull ? 0x100000ull : per_block));
    const uint64_t n_cuts = (uint64_t)n * (s_max + 1u);
    if ((uint64_t)s_max * nb > 0x7fffffffull || n_cuts / 256 > 0x7ffffff0ull) return hipErrorNotSupported;
    ArenaBuf plan_b, cuts_b, lh_b, lp_b, lr_b;
    AP_TRY(plan_b.get(64, stream));
    AP_TRY(cuts_b.get(n_cuts * 4 + 64, stream));
    AP_TRY(lh_b.get(total * 8 + 64, stream));
    AP_TRY(lp_b.get(total * 8 + 64, stream));
    if (!narrow) AP_TRY(lr_b.get(total + 64, stream));
    hipLaunchKernelGGL(ap_plan_kernel, dim3(1), dim3(1024), 0, stream, d_hashes, d_offsets, n, s_max, plan_b.as<ApPlan>());
    hipLaunchKernelGGL(ap_cuts_kernel, dim3((unsigned)((n_cuts + 255) / 256)), dim3(256), 0, stream, d_hashes, d_offsets, n, s_max,
                       (const ApPlan*)plan_b.as<ApPlan>(), cuts_b.as<uint32_t>());
    if (narrow)
        hipLaunchKernelGGL((ap_slice_kernel<true>), dim3(s_max * nb), dim3(SM_THREADS), 0, stream, d_hashes, d_abunds, d_offsets, n,
                           (const uint32_t*)cuts_b.as<uint32_t>(), s_max, (const ApPlan*)plan_b.as<ApPlan>(), lh_b.as<uint64_t>(), lp_b.as<uint64_t>(),
                           (uint8_t*)nullptr);
    else
        hipLaunchKernelGGL((ap_slice_kernel<false>), dim3(s_max * nb), dim3(SM_THREADS), 0, stream, d_hashes, d_abunds, d_offsets, n,
                           (const uint32_t*)cuts_b.as<uint32_t>(), s_max, (const ApPlan*)plan_b.as<ApPlan>(), lh_b.as<uint64_t>(), lp_b.as<uint64_t>(),
                           lr_b.as<uint8_t>());
    AP_TRY(hipGetLastError());
    // hash slices per tile: enough work items to fill the chip twice over when the tiles alone do not (SMG_ABUND_SLICES overrides)
    static const uint32_t z_env = [] { const char* e = getenv("SMG_ABUND_SLICES"); return e ? (uint32_t)atoi(e) : 0u; }();
    // ~2.5 workgroups per CU, and an ODD count: C3 (136 tiles, round 6) 1.49 / 1.19 / 1.28 / 1.18 / 1.28 / 1.21 / 1.27 ms for the call at
    // 2 / 3 / 4 / 5 / 6 / 7 / 8 slices -- with an even count the slices of one tile sit on workgroup ids that differ by less than 8 and
    // so on different XCDs at the same moment, all adding into the same output tile through eight L2s
    uint32_t Z = tiles >= 640 ? 1u : (uint32_t)((640 + tiles - 1) / tiles) | 1u;
    if (Z > (uint32_t)AP_ZMAX - 1u) Z = (uint32_t)AP_ZMAX - 1u;
    if (z_env >= 1 && z_env <= (uint32_t)AP_ZMAX) Z = z_env;
    // (initialised once, whichever host thread comes first: a function-local static's initialiser is serialised by the language)
    static const int attr = [] {
        const hipError_t ea = hipFuncSetAttribute((const void*)ap_join_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AP_LDS);
        const hipError_t eb = hipFuncSetAttribute((const void*)ap_join_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AP_LDS);
        if (ea == hipSuccess && eb == hipSuccess) return 1;
        (void)hipGetLastError();
        return -1;
    }();
    if (attr < 0) return hipErrorNotSupported;
    if (narrow)
        hipLaunchKernelGGL((ap_join_kernel<true>), dim3((unsigned)(tiles * Z)), dim3(AP_THREADS), AP_LDS, stream, (const uint64_t*)lh_b.as<uint64_t>(),
                           (const uint64_t*)lp_b.as<uint64_t>(), (const uint8_t*)nullptr, d_offsets, n, nb, Z, d_common, d_prod);
    else
        hipLaunchKernelGGL((ap_join_kernel<false>), dim3((unsigned)(tiles * Z)), dim3(AP_THREADS), AP_LDS, stream, (const uint64_t*)lh_b.as<uint64_t>(),
                           (const uint64_t*)lp_b.as<uint64_t>(), (const uint8_t*)lr_b.as<uint8_t>(), d_offsets, n, nb, Z, d_common, d_prod);
    return hipGetLastError();
}

}  // namespace smg
