// dictindex.hip -- the compare index (bit columns for the frequent hashes + inverted lists for the rare ones) without
// sorting the collection.
//
// What the index needs is a grouping of every (hash, row) of the CSR by hash: the set of distinct hashes with the number of
// sketches holding each, then per hash either a bit column or the list of its rows.  sparse_pairs.hip gets it from one
// device radix sort of all (hash, row) pairs plus one atomic per element for the bits (3.1 + 2.0 ms at C4, more than the
// triangle of the bit matrix takes).  Here the rows' own order does the work:
//   * the hash space is cut into DX_P buckets by the top bits (hashes are MurmurHash3 values below max_hash: uniform); rows are
//     sorted, so a row's hashes in a bucket are a contiguous slice, and one streaming pass writes where every slice starts
//     (dx_bounds_kernel);
//   * pass 1, one workgroup per bucket: every slice of the bucket is walked (16 rows flattened per wave step, as in
//     gather.hip) and its hashes counted in an LDS hash table -- at most DX_MAXD distinct ones; a bucket that holds more
//     raises a flag and the caller falls back to the sort.  The table becomes the bucket's piece of the dictionary: per
//     distinct hash either its index among the bucket's frequent hashes or the offset of its row list among the bucket's
//     rare elements;
//   * one scan over the DX_P buckets gives every bucket its first bit column and its first rare slot; the host reads the
//     totals (one synchronisation, as before) and runs the cost model;
//   * pass 2, one workgroup per bucket again: the bucket's dictionary goes back into LDS, the slices are walked once more,
//     128 rows at a time; frequent hashes set bits in an LDS tile [128 rows][the bucket's words] that is then written to
//     the bit rows (plain stores; atomicOr only for the first and last word of the span, which neighbouring buckets share),
//     rare hashes append their row to their list through an LDS cursor.
// Device-scope atomics: two per (row, bucket) instead of one per element; no sort; the CSR is read three times.
// Bit columns are numbered bucket by bucket in table order, not by hash value: any bijection gives the same popcounts.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "device_api.hpp"

namespace smg {

namespace {

constexpr int DX_P = DICT_BUCKETS;
constexpr int DX_LOG_P = 9;
static_assert((1 << DX_LOG_P) == DX_P, "");
constexpr int DX_CAP = 2048;                      // LDS table slots per bucket
constexpr int DX_LOG_CAP = 11;
constexpr int DX_MAXD = DICT_MAX_DISTINCT;        // distinct hashes per bucket (half the slots)
constexpr int DX_THREADS = 1024;                  // 16 waves per bucket: the passes are chains of dependent loads, waves are what overlaps them
constexpr int DX_WAVES = DX_THREADS / 64;
constexpr int DX_EPW = 16;                        // rows a wave flattens per step
constexpr int DX_BSEG = 8;                        // segments a row is cut into for the bounds pass
constexpr int DX_CHUNK = DX_WAVES * DX_EPW;       // rows per pass-2 chunk
constexpr int DX_TILE_W = DX_MAXD / 32 + 2;       // words a bucket's bit columns can span
constexpr uint64_t DX_EMPTY = ~0ull;
constexpr uint32_t DX_FREQ = 0x80000000u;         // dictionary value: frequent flag | index, or rare offset
static_assert(DX_CAP == 2 * DX_MAXD && (1 << DX_LOG_CAP) == DX_CAP, "");

struct DxParams {
    unsigned long long max_hash;
};

// bucket of a hash: the range [0, max_hash] cut into DX_P equal parts (monotone in x, so a sorted row's hashes of one bucket
// are a contiguous slice).  Plain top bits would leave up to half of the buckets empty (max_hash = 2^64 / 1000 lies just above
// 2^54), i.e. half of the workgroups idle and twice the load on the others.
struct DxBucketing {
    uint32_t sh;                    // x >> sh: the hash reduced to 32 significant bits
    uint64_t mult;                  // (x32 * mult) >> 32 = x32 * DX_P / (max32 + 1)
};
__device__ __forceinline__ DxBucketing dx_bucketing(unsigned long long max_hash) {
    const int bits = max_hash ? 64 - __clzll((long long)max_hash) : 0;
    DxBucketing g;
    g.sh = bits > 32 ? (uint32_t)(bits - 32) : 0u;
    const uint64_t max32 = max_hash >> g.sh;
    g.mult = ((uint64_t)DX_P << 32) / (max32 + 1);
    return g;
}
__device__ __forceinline__ uint32_t dx_bucket(uint64_t x, const DxBucketing& g) {
    const uint64_t b = ((x >> g.sh) * g.mult) >> 32;
    return b < (uint64_t)DX_P ? (uint32_t)b : (uint32_t)(DX_P - 1);
}
__device__ __forceinline__ uint32_t dx_slot(uint64_t x) { return (uint32_t)((x * 0x9e3779b97f4a7c15ull) >> (64 - DX_LOG_CAP)); }

// largest hash of the collection (rows are sorted: the last element of every row)
__global__ __launch_bounds__(256) void dx_max_kernel(const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ offsets,
                                                     uint32_t n, DxParams* prm) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long m = 0;
    if (r < n) {
        const uint64_t lo = offsets[r], hi = offsets[r + 1];
        if (hi > lo) m = hashes[hi - 1];
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_down(m, off);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(&prm->max_hash, m);
}

// bounds[r][b] = first position of row r whose hash lies in bucket >= b (row length for b = DX_P): one wave per row, every
// element looks at its predecessor and writes the bounds of the buckets that begin between the two
__global__ __launch_bounds__(256) void dx_bounds_kernel(const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ offsets,
                                                        uint32_t n, const DxParams* __restrict__ prm, uint32_t* __restrict__ bounds) {
    const DxBucketing shift = dx_bucketing(prm->max_hash);
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    // a wave takes one of DX_BSEG segments of a row: the walk is a chain of dependent rounds, and a thousand rows of 5,000
    // hashes (C3) would otherwise be a thousand waves running 20 rounds each
    for (uint64_t item = wave; item < (uint64_t)n * DX_BSEG; item += n_waves) {
        const uint64_t r = item / DX_BSEG;
        const uint32_t seg = (uint32_t)(item % DX_BSEG);
        const uint64_t lo = offsets[r];
        const uint32_t len = (uint32_t)(offsets[r + 1] - lo);
        uint32_t* out = bounds + r * (uint64_t)(DX_P + 1);
        const uint32_t s_lo = (uint32_t)((uint64_t)len * seg / DX_BSEG), s_hi = (uint32_t)((uint64_t)len * (seg + 1) / DX_BSEG);
        for (uint32_t i0 = s_lo; i0 < s_hi; i0 += 256) {             // four steps of loads in flight
            uint64_t x[4], px[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + 64u * (uint32_t)u + (uint32_t)lane;
                const uint32_t ii = i < s_hi ? i : s_hi - 1;
                x[u] = hashes[lo + ii];
                px[u] = hashes[lo + (ii ? ii - 1 : 0)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + 64u * (uint32_t)u + (uint32_t)lane;
                if (i < s_hi) {
                    const uint32_t b = dx_bucket(x[u], shift);
                    const uint32_t first = i == 0 ? 0u : dx_bucket(px[u], shift) + 1u;
                    for (uint32_t bb = first; bb <= b; ++bb) out[bb] = i;
                }
            }
        }
        if (seg == DX_BSEG - 1) {
            const uint32_t after = len ? dx_bucket(hashes[lo + len - 1], shift) + 1u : 0u;
            for (uint32_t bb = after + (uint32_t)lane; bb <= (uint32_t)DX_P; bb += 64) out[bb] = len;
        }
    }
}

// The slices of DX_EPW rows inside one bucket as ONE list that the 64 lanes of a wave walk together (the idiom of
// gather.hip's range kernels): lane l < DX_EPW loads slice l, prefix sums give every flattened position its slice.
struct DxGroup {
    uint32_t total;                 // elements of the group
    uint32_t incl, excl;            // lane l < DX_EPW: end / start of slice l in the flattened list
    uint32_t lo_lo, lo_hi;          // lane l < DX_EPW: absolute index of the slice's first element
    uint32_t bound[DX_EPW - 1];     // wave-uniform: end of slices 0 .. 14
};
// the loads of a group (issued early: the chain bounds -> hashes -> table is what a wave waits for) ...
struct DxRaw {
    uint64_t lo;                    // lane l < DX_EPW: absolute index of the first element of slice l
    uint32_t cnt;                   // ... and its length
};
__device__ __forceinline__ DxRaw dx_group_fetch(const uint64_t* __restrict__ offsets, const uint32_t* __restrict__ bounds, uint32_t n,
                                                uint32_t b, uint64_t row0, int lane) {
    DxRaw r;
    r.lo = 0;
    r.cnt = 0;
    const uint64_t row = row0 + (uint64_t)lane;
    if (lane < DX_EPW && row < n) {
        const uint32_t* bp = bounds + row * (uint64_t)(DX_P + 1) + b;
        const uint32_t a = bp[0], e = bp[1];
        r.lo = offsets[row] + a;
        r.cnt = e - a;
    }
    return r;
}
// ... and the prefix sums over them
__device__ __forceinline__ void dx_group_build(DxGroup& g, const DxRaw& r, int lane) {
    uint32_t incl = r.cnt;
#pragma unroll
    for (int s = 1; s < DX_EPW; s <<= 1) {
        const uint32_t v = __shfl_up(incl, s);
        if (lane >= s) incl += v;
    }
    g.total = __shfl(incl, DX_EPW - 1);
#pragma unroll
    for (int k = 0; k < DX_EPW - 1; ++k) g.bound[k] = __shfl(incl, k);
    g.incl = incl;
    g.excl = incl - r.cnt;
    g.lo_lo = (uint32_t)r.lo;
    g.lo_hi = (uint32_t)(r.lo >> 32);
}
// flattened position t (< g.total) -> slice h and the element's absolute index
__device__ __forceinline__ uint64_t dx_group_at(const DxGroup& g, uint32_t t, int& h) {
    h = 0;
#pragma unroll
    for (int k = 0; k < DX_EPW - 1; ++k) h += t >= g.bound[k];
    const uint64_t start = ((uint64_t)(uint32_t)__shfl((int)g.lo_hi, h) << 32) | (uint32_t)__shfl((int)g.lo_lo, h);
    return start + (t - (uint32_t)__shfl((int)g.excl, h));
}

// pass 1: count the distinct hashes of bucket blockIdx.x; leave its piece of the dictionary and its statistics
//   stats[b] = {distinct, frequent, rare elements, overflow}
__global__ __launch_bounds__(DX_THREADS) void dx_count_kernel(const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ offsets,
                                                             uint32_t n, const DxParams* __restrict__ prm,
                                                             const uint32_t* __restrict__ bounds, uint32_t threshold,
                                                             uint64_t* __restrict__ dict_key, uint32_t* __restrict__ dict_val,
                                                             uint32_t* __restrict__ dict_cnt, uint32_t* __restrict__ stats,
                                                             unsigned long long* __restrict__ out) {
    __shared__ unsigned long long s_key[DX_CAP];
    __shared__ uint32_t s_cnt[DX_CAP];
    __shared__ uint32_t s_n, s_ones, s_over, s_out, s_nf, s_rare;
    const uint32_t b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < DX_CAP; k += DX_THREADS) { s_key[k] = DX_EMPTY; s_cnt[k] = 0; }
    if (tid == 0) { s_n = 0; s_ones = 0; s_over = 0; s_out = 0; s_nf = 0; s_rare = 0; }
    __syncthreads();
    volatile uint32_t* over = &s_over;
    auto insert = [&](uint64_t x) {
        if (x == DX_EMPTY) { atomicAdd(&s_ones, 1u); return; }         // the table's empty marker is a legal hash: counted apart
        uint32_t slot = dx_slot(x);
        bool placed = false;
        for (int probe = 0; probe < 256; ++probe) {                     // (a table this crowded has overflowed anyway)
            const unsigned long long k = s_key[slot];
            if (k == x) { placed = true; break; }
            if (k == DX_EMPTY) {
                const unsigned long long prev = atomicCAS(&s_key[slot], DX_EMPTY, (unsigned long long)x);
                if (prev == DX_EMPTY) {
                    if (atomicAdd(&s_n, 1u) >= (uint32_t)DX_MAXD) s_over = 1;   // fuller than the dictionary has room for
                    placed = true;
                    break;
                }
                if (prev == x) { placed = true; break; }
            }
            slot = (slot + 1) & (DX_CAP - 1);
        }
        if (placed) atomicAdd(&s_cnt[slot], 1u);
        else s_over = 1;                                                // table full
    };
    // the next group's bounds are asked for before this group is walked, and a group's hashes two steps at a time
    DxRaw next = dx_group_fetch(offsets, bounds, n, b, (uint64_t)wave * DX_EPW, lane);
    for (uint64_t row0 = (uint64_t)wave * DX_EPW; row0 < n; row0 += DX_CHUNK) {
        if (*over) break;                                           // the bucket does not fit: stop at once, the sort takes over
        const DxRaw cur = next;
        next = dx_group_fetch(offsets, bounds, n, b, row0 + DX_CHUNK, lane);      // (rows past the end: nothing is loaded)
        DxGroup g;
        dx_group_build(g, cur, lane);
        for (uint32_t t0 = 0; t0 < g.total; t0 += 128) {            // wave-uniform trip count (the shuffles read lanes 0 .. 15)
            if (*over) break;                                       // (one LDS word read by all lanes: uniform)
            uint64_t x[2];
            bool ok[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t t = t0 + 64u * (uint32_t)u + (uint32_t)lane;
                ok[u] = t < g.total;
                int h;
                x[u] = hashes[dx_group_at(g, ok[u] ? t : g.total - 1, h)];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (ok[u]) insert(x[u]);
        }
    }
    __syncthreads();
    // the table becomes the bucket's dictionary: value = index among the frequent hashes, or offset among the rare elements
    unsigned long long pairs = 0;
    if (!s_over) {
        for (int k = tid; k < DX_CAP + 1; k += DX_THREADS) {
            unsigned long long key;
            uint32_t m;
            if (k < DX_CAP) { key = s_key[k]; m = s_cnt[k]; if (key == DX_EMPTY) continue; }
            else { key = DX_EMPTY; m = s_ones; if (m == 0) continue; }   // the all-ones hash: last entry, found by its key
            const uint32_t i = atomicAdd(&s_out, 1u);
            uint32_t val;
            if (m > threshold) val = DX_FREQ | atomicAdd(&s_nf, 1u);
            else { val = atomicAdd(&s_rare, m); pairs += (unsigned long long)m * (m - 1) / 2; }
            if (i < (uint32_t)DX_MAXD + 1) {
                dict_key[(uint64_t)b * (DX_MAXD + 1) + i] = key;
                dict_val[(uint64_t)b * (DX_MAXD + 1) + i] = val;
                dict_cnt[(uint64_t)b * (DX_MAXD + 1) + i] = m;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) pairs += __shfl_down(pairs, off);
    if (lane == 0 && pairs) atomicAdd(&out[2], pairs);
    __syncthreads();
    if (tid == 0) {
        stats[b * 4 + 0] = s_out;
        stats[b * 4 + 1] = s_nf;
        stats[b * 4 + 2] = s_rare;
        stats[b * 4 + 3] = s_over;
        if (s_over) atomicAdd(&out[4], 1ull);
    }
}

// exclusive prefixes over the buckets: bases[b] = {first bit column, first rare slot}; totals for the host:
//   out[0] distinct hashes, out[1] frequent ones, out[2] rare pair increments (summed by pass 1), out[3] rare elements,
//   out[4] buckets that overflowed
__global__ __launch_bounds__(DX_P) void dx_scan_kernel(const uint32_t* __restrict__ stats, uint32_t* __restrict__ bases,
                                                       unsigned long long* __restrict__ out) {
    __shared__ unsigned long long s_f[DX_P], s_r[DX_P], s_d[DX_P];
    const int b = threadIdx.x;
    s_d[b] = stats[b * 4 + 0];
    s_f[b] = stats[b * 4 + 1];
    s_r[b] = stats[b * 4 + 2];
    __syncthreads();
    for (int d = 1; d < DX_P; d <<= 1) {                            // Hillis-Steele inclusive scans
        const unsigned long long f = b >= d ? s_f[b - d] : 0, r = b >= d ? s_r[b - d] : 0, u = b >= d ? s_d[b - d] : 0;
        __syncthreads();
        s_f[b] += f; s_r[b] += r; s_d[b] += u;
        __syncthreads();
    }
    bases[b * 2 + 0] = (uint32_t)(s_f[b] - stats[b * 4 + 1]);
    bases[b * 2 + 1] = (uint32_t)(s_r[b] - stats[b * 4 + 2]);
    if (b == DX_P - 1) { out[0] = s_d[b]; out[1] = s_f[b]; out[3] = s_r[b]; }
}

// pass 2: bits and row lists.  Workgroup (row chunk c, bucket range g) owns DX_EROWS rows and walks the DX_BR consecutive
// buckets of its range one after the other: the bucket's dictionary goes into an LDS table, the rows' slices are walked,
// frequent hashes set bits in an LDS tile [rows][the bucket's words], rare hashes append their row to their list (one
// returning device atomic per rare element: the list's cursor).  Consecutive buckets share the word their bit columns meet
// in; inside a range that word is carried from one bucket to the next in LDS, so all stores to the bit rows are plain
// except the first and the last word of the RANGE (shared with the neighbouring ranges' workgroups): 2 atomics per
// (row, range) where one workgroup per bucket needed 2 per (row, bucket) -- 10 million at C4, 0.66 ms of a 1.1 ms pass.
constexpr int DX_ETHREADS = 512;
constexpr int DX_EWAVES = DX_ETHREADS / 64;
constexpr int DX_EROWS = DX_EWAVES * DX_EPW;      // 128 rows per workgroup
constexpr int DX_BR = 32;                         // buckets per range, at most (a power of two; fewer for small collections: see dict_emit_launch)
static_assert(DX_P % DX_BR == 0 && DX_ETHREADS == 4 * DX_EROWS, "");

__global__ __launch_bounds__(DX_ETHREADS) void dx_emit_kernel(const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ offsets,
                                                             uint32_t n, const uint32_t* __restrict__ bounds,
                                                             const uint64_t* __restrict__ dict_key, const uint32_t* __restrict__ dict_val,
                                                             const uint32_t* __restrict__ dict_cnt, uint32_t* __restrict__ dict_cur,
                                                             const uint32_t* __restrict__ stats, const uint32_t* __restrict__ bases,
                                                             uint32_t* __restrict__ bits, uint32_t words_per_row,
                                                             uint32_t* __restrict__ rare_rows, uint32_t* __restrict__ rare_end,
                                                             uint32_t br) {
    __shared__ unsigned long long s_key[DX_CAP];
    __shared__ uint32_t s_val[DX_CAP], s_end[DX_CAP], s_ent[DX_CAP];
    __shared__ uint32_t s_tile[DX_EROWS][DX_TILE_W];
    __shared__ uint32_t s_carry[2][DX_EROWS];
    __shared__ uint32_t s_ones_val, s_ones_end, s_ones_ent;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t chunk0 = (uint64_t)blockIdx.x * DX_EROWS;
    const uint32_t b_lo = blockIdx.y * br;
    const uint32_t frr = (uint32_t)tid >> 2, fw = (uint32_t)tid & 3u;     // flush assignment: row of the chunk, word lane
    const uint64_t frow = chunk0 + frr;
    uint32_t carry_idx = 0xffffffffu;                                     // word the carry belongs to (uniform); none yet
    bool first_word_pending = true;                                       // the range's first word has not been stored yet
    int cb = 0;                                                           // carry buffer in use
    for (uint32_t b = b_lo; b < b_lo + br; ++b) {
        const uint32_t nd = stats[b * 4 + 0], nf = stats[b * 4 + 1];
        if (nd == 0) continue;                                            // uniform
        const uint32_t fbase = bases[b * 2 + 0], rbase = bases[b * 2 + 1];
        const uint32_t w0 = fbase >> 5;
        const uint32_t W = nf ? ((fbase + nf - 1) >> 5) - w0 + 1 : 0u;    // <= DX_TILE_W
        // this wave's slices of the bucket and their first 128 hashes are asked for now: nothing below depends on them until
        // the table is built, two barriers later
        const uint64_t row0 = chunk0 + (uint64_t)wave * DX_EPW;
        const DxRaw raw = dx_group_fetch(offsets, bounds, n, b, row0 < n ? row0 : (uint64_t)n, lane);
        __syncthreads();                                                  // the previous bucket's flush is through
        for (int k = tid; k < DX_CAP; k += DX_ETHREADS) s_key[k] = DX_EMPTY;
        for (uint32_t w = fw; w < W; w += 4) s_tile[frr][w] = 0;
        if (tid == 0) { s_ones_val = 0; s_ones_end = 0; s_ones_ent = 0; }
        DxGroup g;
        dx_group_build(g, raw, lane);
        uint64_t x0[2] = {0, 0};
        int h0[2] = {0, 0};
        if (g.total) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t t = 64u * (uint32_t)u + (uint32_t)lane;
                x0[u] = hashes[dx_group_at(g, t < g.total ? t : g.total - 1, h0[u])];
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < nd; i += DX_ETHREADS) {
            const uint64_t e = (uint64_t)b * (DX_MAXD + 1) + i;
            const unsigned long long key = dict_key[e];
            const uint32_t val = dict_val[e], m = dict_cnt[e];
            const bool freq = (val & DX_FREQ) != 0;
            const uint32_t v = freq ? (DX_FREQ | (fbase + (val & ~DX_FREQ))) : val;      // frequent: global bit column
            if (key == DX_EMPTY) { s_ones_val = v; s_ones_end = freq ? 0u : val + m; s_ones_ent = i; continue; }
            uint32_t slot = dx_slot(key);
            for (;;) {                                                    // keys are distinct: plain claims, nobody looks yet
                if (atomicCAS(&s_key[slot], DX_EMPTY, key) == DX_EMPTY) break;
                slot = (slot + 1) & (DX_CAP - 1);
            }
            s_val[slot] = v;
            s_end[slot] = freq ? 0u : val + m;
            s_ent[slot] = i;
        }
        __syncthreads();
        auto emit = [&](uint64_t x, int h) {
            uint32_t v, end, ent;
            if (x == DX_EMPTY) {
                v = s_ones_val; end = s_ones_end; ent = s_ones_ent;
            } else {
                uint32_t slot = dx_slot(x);
                for (int probe = 0; probe < DX_CAP && s_key[slot] != x; ++probe) slot = (slot + 1) & (DX_CAP - 1);   // present: pass 1 saw it
                v = s_val[slot]; end = s_end[slot]; ent = s_ent[slot];
            }
            if (v & DX_FREQ) {
                const uint32_t f = v & ~DX_FREQ;
                atomicOr(&s_tile[wave * DX_EPW + h][(f >> 5) - w0], 1u << (f & 31));
            } else {
                const uint32_t pos = v + atomicAdd(&dict_cur[(uint64_t)b * (DX_MAXD + 1) + ent], 1u);   // the list's cursor
                rare_rows[(uint64_t)rbase + pos] = (uint32_t)(row0 + (uint64_t)h);
                rare_end[(uint64_t)rbase + pos] = rbase + end;
            }
        };
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (64u * (uint32_t)u + (uint32_t)lane < g.total) emit(x0[u], h0[u]);
        for (uint32_t t0 = 128; t0 < g.total; t0 += 64) {                 // longer groups: the rest, a step at a time
            const uint32_t t = t0 + (uint32_t)lane;
            int h;
            const uint64_t at = dx_group_at(g, t < g.total ? t : g.total - 1, h);
            if (t >= g.total) continue;
            emit(hashes[at], h);
        }
        __syncthreads();
        if (W == 0) continue;
        // flush: words 0 .. W-2 go out, word W-1 becomes the carry; word 0 takes the old carry along if it is the same word
        uint32_t* dst_row = frow < n ? bits + frow * (uint64_t)words_per_row : nullptr;
        const bool same = carry_idx == w0;
        if (fw == 0) {
            uint32_t v0 = s_tile[frr][0];
            if (carry_idx != 0xffffffffu) {
                const uint32_t c = s_carry[cb][frr];
                if (same) v0 |= c;
                else if (dst_row && c) {                                  // the old carry's word is complete: store it
                    if (first_word_pending) atomicOr(dst_row + carry_idx, c); else dst_row[carry_idx] = c;
                }
            }
            if (W == 1) s_carry[cb ^ 1][frr] = v0;                        // still the last word: keep carrying
            else if (dst_row && v0) {
                if (first_word_pending && (same || carry_idx == 0xffffffffu)) atomicOr(dst_row + w0, v0); else dst_row[w0] = v0;
            }
        }
        if (W > 1) {
            for (uint32_t w = fw ? fw : 4u; w + 1 < W; w += 4) {          // interior words (word 0 is handled above)
                const uint32_t v = s_tile[frr][w];
                if (dst_row && v) dst_row[w0 + w] = v;
            }
            if (((W - 1) & 3u) == fw) s_carry[cb ^ 1][frr] = s_tile[frr][W - 1];
        }
        // the range's first word is out once any word has been stored: that is the case unless everything so far is one carry
        if (!(W == 1 && (same || carry_idx == 0xffffffffu))) first_word_pending = false;
        carry_idx = w0 + W - 1;
        cb ^= 1;
    }
    __syncthreads();
    if (carry_idx != 0xffffffffu && fw == 0 && frow < n) {                // the range's last word: the next range shares it
        const uint32_t c = s_carry[cb][frr];
        if (c) atomicOr(bits + frow * (uint64_t)words_per_row + carry_idx, c);
    }
}

}  // namespace

size_t dict_scratch_bytes(uint32_t n) {
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    return up(sizeof(DxParams)) + up((size_t)n * (DX_P + 1) * 4) + up((size_t)DX_P * (DX_MAXD + 1) * 8) +
           3 * up((size_t)DX_P * (DX_MAXD + 1) * 4) + up((size_t)DX_P * 16) + up((size_t)DX_P * 8) + 256;
}

namespace {
struct DxLayout {
    DxParams* prm;
    uint32_t* bounds;
    uint64_t* key;
    uint32_t *val, *cnt, *cur, *stats, *bases;
};
DxLayout dx_layout(void* scratch, uint32_t n) {
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    char* p = (char*)scratch;
    DxLayout l;
    l.prm = (DxParams*)p; p += up(sizeof(DxParams));
    l.bounds = (uint32_t*)p; p += up((size_t)n * (DX_P + 1) * 4);
    l.key = (uint64_t*)p; p += up((size_t)DX_P * (DX_MAXD + 1) * 8);
    l.val = (uint32_t*)p; p += up((size_t)DX_P * (DX_MAXD + 1) * 4);
    l.cnt = (uint32_t*)p; p += up((size_t)DX_P * (DX_MAXD + 1) * 4);
    l.cur = (uint32_t*)p; p += up((size_t)DX_P * (DX_MAXD + 1) * 4);
    l.stats = (uint32_t*)p; p += up((size_t)DX_P * 16);
    l.bases = (uint32_t*)p;
    return l;
}
}  // namespace

// geometry, slice bounds, pass 1 and the scan; d_out[0..4] (zeroed by the caller) as described at dx_scan_kernel
hipError_t dict_count_launch(const uint64_t* d_hashes, const uint64_t* d_offsets, uint32_t n, uint32_t threshold, void* d_scratch,
                             unsigned long long* d_out, hipStream_t stream) {
    if (n == 0) return hipErrorInvalidValue;
    const DxLayout l = dx_layout(d_scratch, n);
    hipLaunchKernelGGL(dx_max_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_hashes, d_offsets, n, l.prm);
    const uint64_t seg_blocks = ((uint64_t)n * DX_BSEG + 3) / 4;
    const unsigned rows_grid = (unsigned)(seg_blocks < 32768 ? seg_blocks : 32768);
    hipLaunchKernelGGL(dx_bounds_kernel, dim3(rows_grid), dim3(256), 0, stream, d_hashes, d_offsets, n, (const DxParams*)l.prm, l.bounds);
    hipLaunchKernelGGL(dx_count_kernel, dim3(DX_P), dim3(DX_THREADS), 0, stream, d_hashes, d_offsets, n, (const DxParams*)l.prm,
                       (const uint32_t*)l.bounds, threshold, l.key, l.val, l.cnt, l.stats, d_out);
    hipLaunchKernelGGL(dx_scan_kernel, dim3(1), dim3(DX_P), 0, stream, (const uint32_t*)l.stats, l.bases, d_out);
    return hipGetLastError();
}

// pass 2: d_bits [n][words_per_row] zeroed by the caller (null when nothing is frequent), d_rare_rows / d_rare_end sized for
// the rare elements d_out[3] reported (null when there are none)
hipError_t dict_emit_launch(const uint64_t* d_hashes, const uint64_t* d_offsets, uint32_t n, void* d_scratch, uint32_t* d_bits,
                            uint32_t words_per_row, uint32_t* d_rare_rows, uint32_t* d_rare_end, hipStream_t stream) {
    const DxLayout l = dx_layout(d_scratch, n);
    if (d_rare_rows) {                                                                        // the rare lists' cursors
        const hipError_t e = hipMemsetAsync(l.cur, 0, (size_t)DX_P * (DX_MAXD + 1) * 4, stream);
        if (e != hipSuccess) return e;
    }
    // a workgroup visits the buckets of its range one after the other (~7 us each): few row chunks -> shorter ranges, so that
    // there are at least ~512 workgroups (two per CU) and the chain a workgroup runs stays short
    const uint32_t n_chunks = (n + DX_EROWS - 1) / DX_EROWS;
    uint32_t br = DX_BR;
    while (br > 1 && (uint64_t)n_chunks * (DX_P / br) < 512) br >>= 1;
    hipLaunchKernelGGL(dx_emit_kernel, dim3(n_chunks, DX_P / br), dim3(DX_ETHREADS), 0, stream, d_hashes,
                       d_offsets, n, (const uint32_t*)l.bounds, (const uint64_t*)l.key, (const uint32_t*)l.val, (const uint32_t*)l.cnt,
                       l.cur, (const uint32_t*)l.stats, (const uint32_t*)l.bases, d_bits, words_per_row, d_rare_rows, d_rare_end, br);
    return hipGetLastError();
}

}  // namespace smg
