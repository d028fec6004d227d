// qindex.hpp -- lookup of a u64 in a sorted query through a first-level table (device side).
// Shared by gather.hip (postings build, apply) and pair_ops.hip (overlaps of a query with every row of a CSR).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace smg {

constexpr uint32_t NONE32 = 0xffffffffu;

// first-level table over the sorted query: bucket b = x >> shift covers Q[T[b], T[b+1]).  Q must be followed by 4
// readable entries (owners keep a padded copy), T has buckets + 1 entries, buckets = (qmax >> shift) + 1.
// One 32-byte record per bucket: where the bucket starts in Q, how many query hashes it holds, and the first three of
// them inline.  With about one hash per bucket a lookup is then ONE dependent step -- two independent 16-byte loads from
// the same record -- instead of table entry -> query entries; fuller buckets (2 % at one hash per bucket on average)
// finish with a binary search in Q.
struct __attribute__((aligned(32))) QRec {
    uint32_t pos, cnt;
    uint64_t h[3];
};

struct QIndex {
    const uint64_t* Q;
    uint64_t nq;
    const uint32_t* T;
    uint32_t shift;
    uint64_t qmax;
    const QRec* rec = nullptr;       // optional (owners that look up many times build it); q_find prefers it
};

// table geometry for nq hashes whose largest is q_max: at most about one hash per bucket (never more than 2 * nq + 1 buckets)
__host__ __device__ inline void qindex_geometry(uint64_t nq, uint64_t q_max, uint32_t* shift, uint32_t* buckets) {
    uint32_t bucket_bits = 0;
    while (bucket_bits < 26 && (1ull << bucket_bits) < nq) ++bucket_bits;
    uint32_t value_bits = 0;
    while (value_bits < 64 && (q_max >> value_bits)) ++value_bits;
    *shift = value_bits > bucket_bits ? value_bits - bucket_bits : 0;
    *buckets = (uint32_t)(q_max >> *shift) + 1;
    // between half a hash and one hash per bucket: with up to two, one bucket in eight held more than the three hashes a
    // record carries inline (Poisson, mean 1.9) and sent its lookups into a second dependent load
    if (*buckets < nq && *shift > 0 && bucket_bits < 26) {
        --*shift;
        *buckets = (uint32_t)(q_max >> *shift) + 1;
    }
}

// Wide loads from addresses that are only 4- / 8-byte aligned.  The hardware takes them (global memory, dword
// alignment); hipcc splits them into narrower instructions unless they are spelled out.
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void load_u32_pair(const uint32_t* p, uint32_t& a, uint32_t& b) {
    u32x2_t v;
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    a = v.x; b = v.y;
}

__device__ __forceinline__ void load_u64_quad(const uint64_t* p, uint64_t& a, uint64_t& b, uint64_t& c, uint64_t& d) {
    u32x4_t v0, v1;
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v0), "=&v"(v1) : "v"(p) : "memory");
    a = (uint64_t)v0.x | ((uint64_t)v0.y << 32); b = (uint64_t)v0.z | ((uint64_t)v0.w << 32);
    c = (uint64_t)v1.x | ((uint64_t)v1.y << 32); d = (uint64_t)v1.z | ((uint64_t)v1.w << 32);
}

// The same lookup in two steps, for callers that keep several lookups in flight: q_rec_load issues the two 16-byte loads
// of the record (plain loads: the compiler schedules the loads of independent lookups back to back and places the waits),
// q_rec_match compares.  `x` must be <= qi.qmax (callers clamp, then discard the result of clamped lanes); qi.rec != null.
struct QRecVal {
    uint4 a, b;
};
__device__ __forceinline__ QRecVal q_rec_load(const QIndex& qi, uint64_t x) {
    const uint4* p = reinterpret_cast<const uint4*>(qi.rec + (x >> qi.shift));
    QRecVal v;
    v.a = p[0];
    v.b = p[1];
    return v;
}
__device__ __forceinline__ uint32_t q_rec_match(const QIndex& qi, uint64_t x, const QRecVal& v) {
    const uint32_t pos = v.a.x, cnt = v.a.y;
    const uint64_t h0 = (uint64_t)v.a.z | ((uint64_t)v.a.w << 32), h1 = (uint64_t)v.b.x | ((uint64_t)v.b.y << 32),
                   h2 = (uint64_t)v.b.z | ((uint64_t)v.b.w << 32);
    uint32_t j = NONE32;
    if (cnt > 2 && h2 == x) j = pos + 2;
    if (cnt > 1 && h1 == x) j = pos + 1;
    if (cnt > 0 && h0 == x) j = pos;
    if (__builtin_expect(cnt > 3 && j == NONE32 && x > h2, 0)) {         // rare (1.5 % of the buckets hold more than three)
        uint64_t q3, q4, q5, q6;
        load_u64_quad(qi.Q + pos + 3, q3, q4, q5, q6);                   // one more step covers buckets of up to seven
        if (q3 == x) j = pos + 3;
        else if (cnt > 4 && q4 == x) j = pos + 4;
        else if (cnt > 5 && q5 == x) j = pos + 5;
        else if (cnt > 6 && q6 == x) j = pos + 6;
        else if (cnt > 7 && x > q6) {
            uint32_t l = pos + 7, h = pos + cnt;
            const uint32_t end = h;
            while (l < h) {
                const uint32_t mid = (l + h) >> 1;
                if (qi.Q[mid] < x) l = mid + 1; else h = mid;
            }
            if (l < end && qi.Q[l] == x) j = l;
        }
    }
    return j;
}

__device__ __forceinline__ uint32_t q_find_rec(const QIndex& qi, uint64_t x) {
    return q_rec_match(qi, x, q_rec_load(qi, x));
}

// Position of x in Q, or NONE32.  The table is sized for about one query hash per bucket, so a bucket almost always
// holds <= 4: those are fetched with two 16-byte loads and compared in registers.  Three load instructions per lookup
// (table pair, two halves of the bucket): every lane of a lookup goes to a different cache line, and a CU serves such
// loads at about one line per cycle, so the number of load INSTRUCTIONS is what a lookup costs.
// Fuller buckets finish with a binary search.
__device__ __forceinline__ uint32_t q_find(const QIndex& qi, uint64_t x) {
    if (x > qi.qmax) return NONE32;
    if (qi.rec) return q_find_rec(qi, x);
    uint32_t lo, hi;
    load_u32_pair(qi.T + (x >> qi.shift), lo, hi);
    if (lo == hi) return NONE32;
    uint64_t q0, q1, q2, q3;
    load_u64_quad(qi.Q + lo, q0, q1, q2, q3);                        // padded: lo + 3 is always readable
    const uint32_t nq = (uint32_t)qi.nq;
    if (q0 == x) return lo;
    if (q1 == x) return lo + 1 < nq ? lo + 1 : NONE32;
    if (q2 == x) return lo + 2 < nq ? lo + 2 : NONE32;
    if (q3 == x) return lo + 3 < nq ? lo + 3 : NONE32;
    if (hi - lo <= 4) return NONE32;
    uint32_t l = lo + 4, h = hi;
    while (l < h) {
        const uint32_t mid = (l + h) >> 1;
        if (qi.Q[mid] < x) l = mid + 1; else h = mid;
    }
    return (l < hi && qi.Q[l] == x) ? l : NONE32;
}

// T[b] = first position of Q with Q[pos] >= b << shift, for b = 0 .. n_buckets (T[n_buckets] = nq)
__device__ __forceinline__ void qindex_fill_bucket(const uint64_t* __restrict__ Q, uint64_t nq, uint32_t shift,
                                                   uint32_t n_buckets, uint32_t* __restrict__ T, uint32_t b) {
    if (b > n_buckets) return;
    if (b == n_buckets) { T[b] = (uint32_t)nq; return; }
    const uint64_t x = (uint64_t)b << shift;
    uint64_t lo = 0, hi = nq;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (Q[mid] < x) lo = mid + 1; else hi = mid;
    }
    T[b] = (uint32_t)lo;
}

// rec[b] from the finished table: bucket b covers Q[T[b], T[b+1])
__device__ __forceinline__ void qindex_fill_record(const uint64_t* __restrict__ Q, const uint32_t* __restrict__ T,
                                                   uint32_t n_buckets, QRec* __restrict__ rec, uint32_t b) {
    if (b >= n_buckets) return;
    const uint32_t pos = T[b], cnt = T[b + 1] - pos;
    QRec r;
    r.pos = pos;
    r.cnt = cnt;
    for (int i = 0; i < 3; ++i) r.h[i] = (uint32_t)i < cnt ? Q[pos + i] : 0ull;
    rec[b] = r;
}

}  // namespace smg
