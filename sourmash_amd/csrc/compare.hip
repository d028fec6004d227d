// compare.hip -- all-vs-all sorted-u64 merge-intersection, tiled in LDS (gfx950).
//
// GPU counterpart of
//   src/sourmash/compare.py:14-64      compare_serial (the N(N-1)/2 Python loop)
//   src/core/src/sketch/minhash.rs:539-558  count_common
//   src/core/src/sketch/minhash.rs:915-953  Intersection (two-pointer walk)
//   src/core/src/sketch/minhash.rs:1765-1807 intersection_size (common, union)
//   src/core/src/sketch/minhash.rs:624-631  jaccard = common / max(1, union)
//
// Layout: CSR -- d_hashes holds every sketch's sorted unique u64 hashes back to
// back, d_offsets[n+1] the row starts.
//
// Kernel (compare_hash_kernel, below): one workgroup owns a tile of 16 row sketches x 32 column sketches.  The tile's
// sketches are streamed through LDS in lock-step "slabs": every round each sketch contributes its next <= 64 hashes
// (coalesced loads, one wave per sketch segment), the slab's upper bound `hi` is the smallest last-loaded hash among
// sketches that still have more to come, the rows' staged hashes <= hi go into an LDS hash table and the columns' are
// looked up in it.  Sketches then advance by exactly what was consumed: the reference's merge walk cut at common hash
// boundaries, so that HBM/L2 sees every hash of a tile once per round.  Works for any length mix (empty, 1-hash,
// 50k-hash rows) because the cut points are data driven.  (Rounds 1-5 also kept the per-pair two-pointer walk over the
// same slabs, compare_tile_kernel, behind SMG_COMPARE_KERNEL=walk, and four table geometries behind SMG_COMPARE_VARIANT:
// no default reached them; removed in round 6 -- profiles/r02_compare_kernels.txt and HISTORY.md keep their numbers.)
//
// union = n_i + n_j - common (scaled sketches; equals the walk's union count),
// so Jaccard needs only the u32 common matrix and the row lengths.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "device_api.hpp"
#include "arena.hpp"
#include "wavemask.hpp"

namespace smg {

constexpr int CT = 16;            // tile edge (sketches)
constexpr int CMP_ZMAX = 16;      // hash-range slices per tile (grid.z); a tile uses ceil(longest / slice_len) of them

__device__ __forceinline__ uint64_t lower_bound_row(const uint64_t* __restrict__ a, uint64_t lo, uint64_t hi, uint64_t x) {
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// One unit of work: slice z (of zt) of the tile (row tile index `ty` in launch order, column tile cb).
struct WorkItem { uint32_t ty, cb, z, zt; };

// Planning pass: one lane per tile decides how many hash-range slices the tile is cut into
// (ceil(longest sketch / slice_len), at most CMP_ZMAX) and appends its work items.  Tiles that get
// sliced (they contain an unusually long sketch) go to the `heavy` list, which the workers drain
// first: longest-processing-time-first scheduling keeps ragged collections from leaving a tail.
__global__ __launch_bounds__(256) void compare_plan_kernel(
    const uint64_t* __restrict__ offsets, uint32_t n, uint32_t row_lo, uint32_t row_hi, int symmetric,
    uint32_t rb_first, uint32_t rb_stride, uint32_t n_row_tiles, uint32_t n_col_tiles, uint32_t slice_len,
    WorkItem* __restrict__ heavy, WorkItem* __restrict__ light, unsigned int* __restrict__ counters, uint32_t ctc) {
    // ctc: columns per tile (HC); rows per tile are CT
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_row_tiles * n_col_tiles) return;
    const uint32_t ty = t / n_col_tiles, cb = t % n_col_tiles;
    const uint32_t rb = rb_first + ty * rb_stride;
    const uint32_t row0 = row_lo + rb * CT, col0 = cb * ctc;
    if (symmetric && col0 + ctc <= row0) return;             // strictly below the diagonal
    uint64_t best = 0;
    for (uint32_t i = 0; i < ctc; ++i) {
        const uint32_t r = row0 + i, c = col0 + i;
        if (i < (uint32_t)CT && r < row_hi) { const uint64_t l = offsets[r + 1] - offsets[r]; best = l > best ? l : best; }
        if (c < n) { const uint64_t l = offsets[c + 1] - offsets[c]; best = l > best ? l : best; }
    }
    if (best == 0) return;                                    // nothing can intersect
    uint32_t zt = (uint32_t)((best + slice_len - 1) / slice_len);
    zt = zt < 1 ? 1 : (zt > (uint32_t)CMP_ZMAX ? (uint32_t)CMP_ZMAX : zt);
    WorkItem* list = zt > 1 ? heavy : light;
    const unsigned int base = atomicAdd(&counters[zt > 1 ? 0 : 1], zt);
    for (uint32_t z = 0; z < zt; ++z) list[base + z] = WorkItem{ty, cb, z, zt};
}

// Workers: persistent workgroups pull work items (heavy list first) with one atomic per item.

// ---- the tile through a hash table ------------------------------------------------------------------------------------
// A two-pointer walk spends one step per element of BOTH lists for EVERY pair (256 pairs x (n_i + n_j) steps per tile,
// 13 VALU instructions a step): at C4 it was bound by instruction issue, not by LDS or memory.  The slab structure
// admits a formulation whose work grows with the ELEMENTS of a tile instead: per round, the <= 64 staged hashes of each
// of the 16 ROW sketches are inserted into an LDS hash table (key = hash, value = 16-bit mask of the rows holding it),
// then every staged hash of the 32 COLUMN sketches is looked up once; bit r of the mask it finds says row r shares it.
// Rounds are cut at the same data-driven bound `hi` as the walk (every sketch's hashes <= hi are among its staged ones),
// so the counts are the walk's counts: |A_r ∩ B_c| summed over disjoint hash ranges.
// Per round: 1,024 inserts + 2,048 lookups for 512 pairs.  The loop is written around its VALU instruction count (it is
// what bounded the first version): the masks are counted in per-lane 4-bit counters (3 instructions per 8
// (row, column) cells, nothing divergent) that go to the tile's LDS counters every 15 rounds through a DPP row sum; a
// probe reads an aligned pair of slots; the sets of lanes still probing live in scalar registers as wave masks.
constexpr int HR = CT;                // row sketches of a tile (inserted)
constexpr int HC = 32;                // column sketches of a tile (looked up)
constexpr int HSEG = 64;              // hashes staged per sketch and round: one per lane
// table slots: 2^LOGT (template parameter); rows put in at most 16 x 64 = 1024 distinct hashes per round
constexpr unsigned long long H_EMPTY = ~0ull;   // never a key: a staged hash equal to 2^64 - 1 is counted out of band
constexpr int HWAVES = 8;             // waves per workgroup: sketch s = i * HWAVES + wave, so every wave holds HR / 8 rows
constexpr int HBLOCK = HWAVES * 64;
constexpr int HPW = (HR + HC) / HWAVES;   // sketches per wave
static_assert(HR % HWAVES == 0 && HC % HWAVES == 0, "");

constexpr int NR = HR / HWAVES;       // row sketches per wave
constexpr int NC = HC / HWAVES;       // column sketches per wave
static_assert(HR == 16 && NC == 4 && HC == 32, "the packed counters below hold 16 rows x 4 columns per wave");

template <int LOGT>
__device__ __forceinline__ uint32_t pair_slot(uint64_t v) {           // the even slot a hash's probe sequence starts on
    // the hashes of a round share their top bits (a narrow value range), so the slot comes from the low word folded with
    // the high one; shifts and xors only -- a 32-bit multiply costs four of them on this VALU
    uint32_t x = (uint32_t)v ^ (uint32_t)(v >> 32);
    x ^= x >> 15;
    return (x & (uint32_t)((1 << (LOGT - 1)) - 1)) << 1;
}

template <int CTRL>
__device__ __forceinline__ uint32_t row_rotated(uint32_t x) {         // lane l of a 16-lane row <- lane (l + n) % 16: DPP row_ror:n
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, false);
}

// The per-lane 4-bit counters of a wave (see the kernel) go to the tile's LDS counters.  acc[p][k] nibble j counts bit
// t = 4j + k of y_p = m[2p] | m[2p+1] << 16, i.e. row t & 15 of column j' = 2p + (t >> 4) of this wave (tile column
// 8j' + wave).  Even and odd nibbles are split into byte lanes (<= 15 each), summed over the 16 lanes of a DPP row
// (<= 240, still a byte), and lane b < 4 of each row adds byte b of every register to its counter.
__device__ __forceinline__ void flush_nibble_counts(uint32_t (&acc)[NC / 2][4], uint32_t* s_cnt, int lane, int wave) {
    const int b = lane & 3;
    const bool pusher = (lane & 15) < 4;
    uint32_t* const mine = s_cnt + (8 * (b & 1)) * HC + (b >> 1) * HWAVES + wave;
    const uint32_t sh = 8u * (uint32_t)b;
#pragma unroll
    for (int p = 0; p < NC / 2; ++p)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t part[2] = {acc[p][k] & 0x0f0f0f0fu, (acc[p][k] >> 4) & 0x0f0f0f0fu};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint32_t x = part[h];
                x += row_rotated<0x128>(x);
                x += row_rotated<0x124>(x);
                x += row_rotated<0x122>(x);
                x += row_rotated<0x121>(x);
                // byte b of (p, k, h): row 8 (b & 1) + 4 h + k, column 8 (2 p + (b >> 1)) + wave
                if (pusher) atomicAdd(&mine[(4 * h + k) * HC + 2 * HWAVES * p], (x >> sh) & 0xffu);
            }
            acc[p][k] = 0;
        }
}

template <int MINW, int LOGT>
__global__ __launch_bounds__(HBLOCK, MINW) void compare_hash_kernel(
    const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ offsets, uint32_t n,
    uint32_t row_lo, uint32_t row_hi, uint32_t* __restrict__ common, int symmetric,
    uint32_t rb_first, uint32_t rb_stride, const WorkItem* __restrict__ heavy,
    const WorkItem* __restrict__ light, unsigned int* __restrict__ counters) {
    // same contract as compare_tile_kernel (symmetric modes, work lists, output rows), tiles of HR rows x HC columns
    constexpr int HT = 1 << LOGT;
    __shared__ __attribute__((aligned(16))) unsigned long long s_key[HT];     // pairs of slots are read as one 16-byte access
    __shared__ uint32_t s_mask[HT / 2];            // 16-bit row masks of the slots, two to a word
    __shared__ uint32_t s_cnt[HR * HC];
    __shared__ uint64_t s_pos[HR + HC], s_end[HR + HC];
    __shared__ unsigned long long s_hi2[2];        // the round's bound, one cell per round parity
    __shared__ uint32_t s_live2[2], s_top[2];      // s_live2: bit 0 rows / bit 1 columns with data left; s_top: rows / columns
    __shared__ uint64_t s_piv[2];                  // whose staged part holds the hash 2^64 - 1
    __shared__ WorkItem s_item;
    __shared__ int s_have;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < HT; k += HBLOCK) { s_key[k] = H_EMPTY; if (k < HT / 2) s_mask[k] = 0; }

    for (;;) {
        __syncthreads();                                   // previous item fully done (LDS reuse)
        if (tid == 0) {
            int have = 0;
            unsigned int i = atomicAdd(&counters[2], 1u);
            if (i < counters[0]) { s_item = heavy[i]; have = 1; }
            else {
                i = atomicAdd(&counters[3], 1u);
                if (i < counters[1]) { s_item = light[i]; have = 1; }
            }
            s_have = have;
        }
        for (int k = tid; k < HR * HC; k += HBLOCK) s_cnt[k] = 0;
        __syncthreads();
        if (!s_have) return;
        const WorkItem it = s_item;
        const uint32_t rb = rb_first + it.ty * rb_stride;
        const uint32_t row0 = row_lo + rb * HR, col0 = it.cb * HC;
        if (tid < HR + HC) {
            const uint32_t s = tid < HR ? row0 + tid : col0 + (tid - HR);
            const bool ok = tid < HR ? (s < row_hi) : (s < n);
            s_pos[tid] = ok ? offsets[s] : 0;
            s_end[tid] = ok ? offsets[s + 1] : 0;
        }
        __syncthreads();
        if (it.zt > 1) {                                   // hash-range slice of a heavy tile
            if (tid == 0) {
                uint64_t best_len = 0, best_pos = 0;
                for (int i = 0; i < HR + HC; ++i) {
                    const uint64_t len = s_end[i] - s_pos[i];
                    if (len > best_len) { best_len = len; best_pos = s_pos[i]; }
                }
                s_piv[0] = it.z == 0 ? 0ull : hashes[best_pos + (uint64_t)it.z * best_len / it.zt];
                s_piv[1] = it.z + 1 == it.zt ? ~0ull : hashes[best_pos + (uint64_t)(it.z + 1) * best_len / it.zt];
            }
            __syncthreads();
            if (tid < HR + HC) {
                const uint64_t lo = s_pos[tid], hi = s_end[tid];
                const uint64_t a = it.z == 0 ? lo : lower_bound_row(hashes, lo, hi, s_piv[0]);
                const uint64_t b = it.z + 1 == it.zt ? hi : lower_bound_row(hashes, a, hi, s_piv[1]);
                s_pos[tid] = a;
                s_end[tid] = b;
            }
            __syncthreads();
        }
        if (tid == 0) { s_top[0] = 0; s_top[1] = 0; s_hi2[0] = ~0ull; s_hi2[1] = ~0ull; s_live2[0] = 0; s_live2[1] = 0; }
        // this wave's sketches: s = i * HWAVES + wave; positions are wave-uniform and live in scalar registers
        const uint64_t* at[HPW];                           // next unread hash of sketch i
        uint32_t left[HPW];                                // hashes still unread (a slice of a sketch is < 2^32 long)
#pragma unroll
        for (int i = 0; i < HPW; ++i) {
            const uint64_t p0 = uniform64(s_pos[i * HWAVES + wave]), p1 = uniform64(s_end[i * HWAVES + wave]);
            at[i] = hashes + p0;
            left[i] = (uint32_t)(p1 - p0);
        }
        __syncthreads();                                   // s_top / s_hi2 / s_live2 are set before anyone's atomics
        uint64_t e[HPW];                                   // lane l: the l-th staged hash of sketch i (2^64 - 1 past its end)
#pragma unroll
        for (int i = 0; i < HPW; ++i) e[i] = (uint32_t)lane < left[i] ? at[i][lane] : ~0ull;
        uint32_t acc[NC / 2][4];                           // this lane's share of the tile's counts, 4 bits per (row, column)
#pragma unroll
        for (int p = 0; p < NC / 2; ++p)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[p][k] = 0;
        uint32_t since = 0;                                // rounds added to acc since it last went to s_cnt (wave-uniform)

        for (uint32_t round = 0;; ++round) {
            const uint32_t par = round & 1u;
            // ---- the round's bound: smallest 64th staged hash among sketches with more to come.  `left` is wave-uniform,
            //      so which sketches take part is decided in scalar code; lane 63 holds their 64th hashes ----
            uint64_t wmin = ~0ull;
            uint32_t live = 0;
#pragma unroll
            for (int i = 0; i < HPW; ++i) {
                if (left[i] > (uint32_t)HSEG) wmin = e[i] < wmin ? e[i] : wmin;
                if (left[i]) live |= i < NR ? 1u : 2u;
            }
            if (lane == HSEG - 1 && wmin != ~0ull) atomicMin(&s_hi2[par], (unsigned long long)wmin);
            if (lane == 0 && live) atomicOr(&s_live2[par], live);
            __syncthreads();
            if (s_live2[par] != 3u) break;                 // every row or every column exhausted
            const uint64_t hi = uniform64(s_hi2[par]);
            if (tid == 0) { s_hi2[par ^ 1u] = ~0ull; s_live2[par ^ 1u] = 0; }     // next round's cells (not touched before its atomics)
            // ---- what this round consumes (a prefix of every staged segment); the next round's hashes start loading now.
            //      hi = 2^64 - 1 only when no sketch has more than it has staged: that round consumes everything left and is
            //      the only one that can meet the hash 2^64 - 1 (the empty-slot key), which is counted out of band.  Lanes
            //      past a segment's end hold 2^64 - 1 and fail `<= cut` by themselves ----
            const bool last = hi == ~0ull;
            const uint64_t cut = last ? ~0ull - 1 : hi;
            bool mine[HPW];
            uint32_t take[HPW];
#pragma unroll
            for (int i = 0; i < HPW; ++i) {
                mine[i] = e[i] <= cut;
                take[i] = last ? left[i] : (uint32_t)__popcll(mask_of(mine[i]));
                at[i] += take[i];
                left[i] -= take[i];
            }
            uint64_t en[HPW];
#pragma unroll
            for (int i = 0; i < HPW; ++i) en[i] = (uint32_t)lane < left[i] ? at[i][lane] : ~0ull;
            if (last) {
#pragma unroll
                for (int i = 0; i < HPW; ++i)
                    if ((uint32_t)lane < take[i] && e[i] == H_EMPTY)
                        atomicOr(&s_top[i < NR ? 0 : 1], 1u << (i < NR ? i * HWAVES + wave : i * HWAVES + wave - HR));
            }
            // ---- rows: insert.  Probing starts on an even slot and goes up one slot at a time; the probes of the lane's
            //      hashes go out together.  Which lanes are still probing is kept as wave masks in scalar registers ----
            uint32_t ro[NR];                               // byte offset into s_key of the slot being tried, then of the slot held
            uint64_t ron[NR];
#pragma unroll
            for (int i = 0; i < NR; ++i) { ro[i] = pair_slot<LOGT>(e[i]) * 8u; ron[i] = mask_of(mine[i]); }
            for (;;) {
                uint64_t any = 0;
#pragma unroll
                for (int i = 0; i < NR; ++i) any |= ron[i];
                if (!any) break;
                unsigned long long old[NR];
#pragma unroll
                for (int i = 0; i < NR; ++i)
                    if (lanes_of(ron[i])) old[i] = atomicCAS((unsigned long long*)((char*)s_key + ro[i]), H_EMPTY, (unsigned long long)e[i]);
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    opaque(old[i]);                        // lanes outside ron[i] hold nothing; their compare bits are masked
                    ron[i] &= mask_of(old[i] != H_EMPTY) & mask_of(old[i] != e[i]);     // slot taken by another hash
                    if (lanes_of(ron[i])) ro[i] = (ro[i] + 8u) & (uint32_t)(HT * 8 - 1);
                }
            }
#pragma unroll
            for (int i = 0; i < NR; ++i)                   // 16-bit row masks, two to a word: slot s -> half (s & 1) of word s >> 1
                if (mine[i]) atomicOr(&s_mask[ro[i] >> 4], (1u << (i * HWAVES + wave)) << ((ro[i] & 8u) << 1));
            __syncthreads();
            // ---- columns: one lookup per hash, a pair of slots per probe (a key sits before the first empty slot of its
            //      probe sequence, and an odd slot is never filled before its even neighbour) ----
            uint32_t co[NC];
            uint64_t con[NC], fnd[NC], odd[NC];            // lanes still probing / that found their hash / found it in the odd slot
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                co[j] = pair_slot<LOGT>(e[NR + j]) * 8u;
                con[j] = mask_of(mine[NR + j]);
                fnd[j] = 0;
                odd[j] = 0;
            }
            for (;;) {
                uint64_t any = 0;
#pragma unroll
                for (int j = 0; j < NC; ++j) any |= con[j];
                if (!any) break;
                ulonglong2 kk[NC];
#pragma unroll
                for (int j = 0; j < NC; ++j)
                    if (lanes_of(con[j])) kk[j] = *(const ulonglong2*)((const char*)s_key + co[j]);
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    opaque(kk[j].x);                       // lanes outside con[j] hold nothing; their compare bits are masked
                    opaque(kk[j].y);
                    const uint64_t b1 = mask_of(kk[j].y == e[NR + j]) & con[j];
                    const uint64_t bh = (mask_of(kk[j].x == e[NR + j]) & con[j]) | b1;
                    fnd[j] |= bh;
                    odd[j] |= b1;
                    con[j] &= mask_of(kk[j].y != H_EMPTY) & ~bh;
                    if (lanes_of(con[j])) co[j] = (co[j] + 16u) & (uint32_t)(HT * 8 - 1);
                }
            }
            uint32_t m[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j)
                m[j] = lanes_of(fnd[j]) ? (uint32_t)((const uint16_t*)s_mask)[(co[j] >> 3) + (lanes_of(odd[j]) ? 1u : 0u)] : 0u;
            const bool hit = (fnd[0] | fnd[1] | fnd[2] | fnd[3]) != 0;
            // ---- counts: bit r of m[j] says row r shares the lane's hash of column j.  Every lane keeps 4-bit counters of
            //      its own (16 rows x 4 columns in 8 registers: 3 instructions per 8 counters, nothing divergent, no LDS);
            //      after 15 rounds they are summed over the lanes and added to s_cnt ----
            if (hit) {                                     // wave-uniform
#pragma unroll
                for (int p = 0; p < NC / 2; ++p) {
                    const uint32_t y = m[2 * p] | (m[2 * p + 1] << 16);
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[p][k] += (y >> k) & 0x11111111u;
                }
                if (++since == 15u) { flush_nibble_counts(acc, s_cnt, lane, wave); since = 0; }
            }
            __syncthreads();
            // ---- the table goes back to empty: every inserter clears the slot its hash is in (ordered before the next
            //      round's inserts by the barrier after its bound) ----
#pragma unroll
            for (int i = 0; i < NR; ++i)
                if (mine[i]) {
                    *(unsigned long long*)((char*)s_key + ro[i]) = H_EMPTY;
                    ((uint16_t*)s_mask)[ro[i] >> 3] = 0;
                }
#pragma unroll
            for (int i = 0; i < HPW; ++i) e[i] = en[i];
        }
        if (since) flush_nibble_counts(acc, s_cnt, lane, wave);
        __syncthreads();

        // the hash 2^64 - 1 (possible with scaled = 1) never went through the table
        for (int k = tid; k < HR * HC; k += HBLOCK)
            if (((s_top[0] >> (k / HC)) & 1u) && ((s_top[1] >> (k % HC)) & 1u)) s_cnt[k] += 1;
        __syncthreads();
        for (int k = tid; k < HR * HC; k += HBLOCK) {
            const uint32_t cnt = s_cnt[k];
            const uint32_t r = (uint32_t)k / HC, c = (uint32_t)k % HC;
            const uint32_t row = row0 + r, col = col0 + c;
            if (row < row_hi && col < n && cnt) {
                const uint64_t idx = (uint64_t)(it.ty * HR + r) * n + col;
                if (!symmetric) {
                    atomicAdd(&common[idx], cnt);
                } else if (col >= row) {
                    atomicAdd(&common[idx], cnt);
                    if (symmetric == 1 && col != row) atomicAdd(&common[(uint64_t)(col - row_lo) * n + row], cnt);
                }
            }
        }
    }
}


__global__ __launch_bounds__(256) void jaccard_from_counts_kernel(const uint32_t* __restrict__ common,
                                                                  const uint64_t* __restrict__ offsets, uint32_t n,
                                                                  uint32_t row_lo, uint32_t row_hi,
                                                                  double* __restrict__ out) {
    const uint64_t total = (uint64_t)(row_hi - row_lo) * n;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t row = row_lo + (uint32_t)(i / n), col = (uint32_t)(i % n);
        double v;
        if (row == col) {
            v = 1.0;                                              // compare.py:33 np.ones diagonal
        } else {
            const uint64_t ni = offsets[row + 1] - offsets[row], nj = offsets[col + 1] - offsets[col];
            const uint64_t cm = common[i];
            const uint64_t uni = ni + nj - cm;
            v = (double)cm / (double)(uni > 1 ? uni : 1);         // minhash.rs:624-631
        }
        out[i] = v;
    }
}

// Slice length: long enough that ordinary sketches (a few thousand hashes) are never cut when there
// are plenty of tiles, shorter when the tile count alone cannot fill 256 CUs x 4 workgroups.
static uint32_t pick_slice_len(uint64_t n_tiles) {
    if (n_tiles >= 2048) return 8192;
    if (n_tiles >= 512) return 2048;
    return 1024;
}

size_t compare_workspace_bytes(uint32_t n_row_tiles, uint32_t n_col_tiles) {
    const size_t tiles = (size_t)n_row_tiles * n_col_tiles;
    return 256 + tiles * sizeof(WorkItem) * (CMP_ZMAX + 1);        // heavy (x ZMAX) + light (x 1) lists
}

static hipError_t compare_launch(const uint64_t* d_hashes, const uint64_t* d_offsets, uint32_t n, uint32_t row_lo,
                                 uint32_t row_hi, int symmetric, uint32_t rb_first, uint32_t rb_stride,
                                 uint32_t n_row_tiles, uint32_t* d_common, hipStream_t stream) {
    const uint32_t ctc = HC;                                          // columns per tile
    const uint32_t n_col_tiles = (n + ctc - 1) / ctc;
    const size_t tiles = (size_t)n_row_tiles * n_col_tiles;
    // the sharded form owns whole 16-row tiles ([n_row_tiles * 16][n]); the others own exactly rows [row_lo, row_hi)
    const size_t out_rows = symmetric == 2 ? (size_t)n_row_tiles * CT : (size_t)(row_hi - row_lo);
    hipError_t e = hipMemsetAsync(d_common, 0, out_rows * n * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    void* ws = nullptr;
    e = arena_alloc(&ws, compare_workspace_bytes(n_row_tiles, n_col_tiles), stream);   // cached scratch, released in stream order
    if (e != hipSuccess) return e;
    unsigned int* counters = (unsigned int*)ws;
    WorkItem* heavy = (WorkItem*)((char*)ws + 256);
    WorkItem* light = heavy + tiles * CMP_ZMAX;
    e = hipMemsetAsync(counters, 0, 256, stream);
    if (e == hipSuccess) {
        const uint64_t work_tiles = tiles / (symmetric ? 2 : 1);
        hipLaunchKernelGGL(compare_plan_kernel, dim3((unsigned)((tiles + 255) / 256)), dim3(256), 0, stream, d_offsets, n,
                           row_lo, row_hi, symmetric, rb_first, rb_stride, n_row_tiles, n_col_tiles,
                           pick_slice_len(work_tiles), heavy, light, counters, ctc);
        const uint64_t cap = 256ull * 3;                               // 4,096 slots (load factor 1/4), 43 KiB of LDS, 80 VGPRs: 3 workgroups per CU
        const unsigned grid0 = (unsigned)(work_tiles + 1 < cap ? work_tiles + 1 : cap);
        hipLaunchKernelGGL((compare_hash_kernel<6, 12>), dim3(grid0 < 1 ? 1 : grid0), dim3(HBLOCK), 0, stream, d_hashes, d_offsets, n, row_lo, row_hi,
                           d_common, symmetric, rb_first, rb_stride, heavy, light, counters);
        e = hipGetLastError();
    }
    arena_free(ws, stream);
    return e;
}

hipError_t compare_counts_launch(const uint64_t* d_hashes, const uint64_t* d_offsets, uint32_t n, uint32_t row_lo,
                                 uint32_t row_hi, uint32_t* d_common, hipStream_t stream) {
    if (row_hi <= row_lo || n == 0) return hipSuccess;
    const int symmetric = (row_lo == 0 && row_hi == n) ? 1 : 0;
    return compare_launch(d_hashes, d_offsets, n, row_lo, row_hi, symmetric, 0u, 1u, (row_hi - row_lo + CT - 1) / CT,
                          d_common, stream);
}

// Sharded form: this launch owns the row tiles rb_first, rb_first + rb_stride, ... (rb_count of them)
// of the full n x n problem and computes only their tiles on or above the diagonal.
hipError_t compare_blocks_launch(const uint64_t* d_hashes, const uint64_t* d_offsets, uint32_t n, uint32_t rb_first,
                                 uint32_t rb_stride, uint32_t rb_count, uint32_t* d_common, hipStream_t stream) {
    if (rb_count == 0 || n == 0) return hipSuccess;
    return compare_launch(d_hashes, d_offsets, n, 0u, n, 2, rb_first, rb_stride, rb_count, d_common, stream);
}

__global__ __launch_bounds__(256) void symmetrize_kernel(uint32_t* __restrict__ m, uint32_t n) {
    // m[j][i] = m[i][j] for i < j.  32 x 32 tiles through LDS so both the read and the write are coalesced.
    __shared__ uint32_t t[32][33];
    const uint32_t bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return;
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (uint32_t r = ty; r < 32; r += 8) {
        const uint32_t i = bi * 32 + r, j = bj * 32 + tx;
        t[r][tx] = (i < n && j < n) ? m[(uint64_t)i * n + j] : 0;
    }
    __syncthreads();
    for (uint32_t r = ty; r < 32; r += 8) {
        const uint32_t j = bj * 32 + r, i = bi * 32 + tx;        // writing element (j, i), j is the row
        if (i < n && j < n && i < j) m[(uint64_t)j * n + i] = t[tx][r];
    }
}

hipError_t symmetrize_launch(uint32_t* d_common, uint32_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    dim3 grid((n + 31) / 32, (n + 31) / 32);
    hipLaunchKernelGGL(symmetrize_kernel, grid, dim3(256), 0, stream, d_common, n);
    return hipGetLastError();
}

hipError_t jaccard_from_counts_launch(const uint32_t* d_common, const uint64_t* d_offsets, uint32_t n,
                                      uint32_t row_lo, uint32_t row_hi, double* d_out, hipStream_t stream) {
    if (row_hi <= row_lo || n == 0) return hipSuccess;
    const uint64_t total = (uint64_t)(row_hi - row_lo) * n;
    const uint64_t nb = (total + 255) / 256;
    hipLaunchKernelGGL(jaccard_from_counts_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, stream,
                       d_common, d_offsets, n, row_lo, row_hi, d_out);
    return hipGetLastError();
}

}  // namespace smg
