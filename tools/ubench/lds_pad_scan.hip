// Reproducer attempt for profiles/HISTORY.md 4.4 "the slice fill": does a scan over a sorted slice in LDS ever run past the two words of
// 2^64 - 1 written behind the slice, depending on the FORM of the code that fills the table slice next to it?
//
// One 1,024-thread workgroup per CU walks the ranges of a query exactly as overlap_lean_kernel does (table slice + query slice into
// LDS between two barriers, padding behind the slice, lookups: table entry -> two query hashes -> scan on in a bucket of three
// or more), with synthetic probes instead of database rows.  Every scan step at or behind the padding is counted: with the
// padding in place there can be none.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench/lds_pad_scan tools/ubench/lds_pad_scan.hip    (built here, run on the GPU box)
//   tools/ubench/lds_pad_scan            -> one line per fill form: ms, scan steps, steps behind the slice, hits
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

constexpr int THREADS = 1024, WAVES = THREADS / 64, QCAP = 11264, BUCKETS = 10240, VISITS = 25;
constexpr size_t LDS_BYTES = ((size_t)QCAP + 2) * 8 + ((size_t)BUCKETS + 4) * 4;

__device__ __forceinline__ uint64_t mask_of(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool lanes_of(uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
__device__ __forceinline__ uint32_t uniform32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// FORM 0: the guarded store (`if (i < cnt_t) s_t[i] = T[b0 + i] - p0`) + two padding slots written by two threads afterwards
// FORM 1: every slot stored, a select between the table entry and the slice's size
template <int FORM>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
void walk(const uint64_t* __restrict__ Q, const uint32_t* __restrict__ T, uint32_t n_buckets, uint32_t shift, uint32_t n_ranges,
          uint32_t bpr, unsigned long long* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
    uint64_t* s_q = lds;
    uint32_t* s_t = reinterpret_cast<uint32_t*>(s_q + QCAP + 2);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int QPER = (QCAP + THREADS - 1) / THREADS, TPER = (BUCKETS + 1 + THREADS - 1) / THREADS;
    uint32_t n_p0 = T[0], n_p1 = T[bpr < n_buckets ? bpr : n_buckets];
    unsigned long long steps = 0, behind = 0, hits = 0;
    for (uint32_t r = 0; r < n_ranges; ++r) {
        const uint32_t b0 = r * bpr, b1 = b0 + bpr < n_buckets ? b0 + bpr : n_buckets;
        const bool last = r + 1 == n_ranges;
        const uint64_t upper = last ? ~0ull : ((uint64_t)b1 << shift);
        const uint32_t p0 = uniform32(n_p0), cnt_q = uniform32(n_p1) - p0, cnt_t = b1 - b0 + 1;
        __syncthreads();
        {
            constexpr int FILL_STEP = 4;
#pragma unroll
            for (int u0 = 0; u0 < TPER; u0 += FILL_STEP) {
                uint32_t tv[FILL_STEP];
#pragma unroll
                for (int u = 0; u < FILL_STEP; ++u) {
                    const uint32_t i = (uint32_t)tid + (uint32_t)(u0 + u) * THREADS;
                    tv[u] = (u0 + u < TPER && i < cnt_t) ? T[b0 + i] : 0u;
                }
#pragma unroll
                for (int u = 0; u < FILL_STEP; ++u) {
                    const uint32_t i = (uint32_t)tid + (uint32_t)(u0 + u) * THREADS;
                    if (FORM == 0) {
                        if (u0 + u < TPER && i < cnt_t) s_t[i] = tv[u] - p0;
                    } else {
                        if (u0 + u < TPER && i < (uint32_t)BUCKETS + 4u) s_t[i] = i < cnt_t ? tv[u] - p0 : cnt_q;
                    }
                }
            }
#pragma unroll
            for (int u0 = 0; u0 < QPER; u0 += FILL_STEP) {
                uint64_t qv[FILL_STEP];
#pragma unroll
                for (int u = 0; u < FILL_STEP; ++u) {
                    const uint32_t i = (uint32_t)tid + (uint32_t)(u0 + u) * THREADS;
                    qv[u] = (u0 + u < QPER && i < cnt_q) ? Q[p0 + i] : 0ull;
                }
#pragma unroll
                for (int u = 0; u < FILL_STEP; ++u) {
                    const uint32_t i = (uint32_t)tid + (uint32_t)(u0 + u) * THREADS;
                    if (u0 + u < QPER && i < cnt_q) s_q[i] = qv[u];
                }
            }
        }
        if (FORM == 0) {
            if (tid < 2) { s_t[bpr + 1 + tid] = cnt_q; s_q[cnt_q + tid] = ~0ull; }
            if (tid >= 64 && tid < 64 + 3 && cnt_t + (uint32_t)(tid - 64) <= bpr) s_t[cnt_t + (uint32_t)(tid - 64)] = cnt_q;
        } else {
            if (tid < 2) s_q[cnt_q + tid] = ~0ull;
        }
        __syncthreads();
        if (!last) {
            const uint32_t nb0 = b1, nb1 = nb0 + bpr < n_buckets ? nb0 + bpr : n_buckets;
            n_p0 = T[nb0];
            n_p1 = T[nb1];
        }
        for (int v = 0; v < VISITS; ++v) {
            // probes: query hashes of the range (half of them + 1: not in the query), a few hashes of the next range, filler
            const uint32_t idx = ((uint32_t)lane * 151u + (uint32_t)wave * 977u + (uint32_t)v * 7919u + r * 31u + blockIdx.x * 13u) % (cnt_q + 16u);
            uint64_t e = ~0ull;
            if (idx < cnt_q) e = Q[p0 + idx] + (uint64_t)(lane & 1);
            else if (idx < cnt_q + 8u && !last) e = upper + idx;
            const uint64_t in = mask_of(e < upper);
            uint32_t kk = (uint32_t)(e >> shift) - b0;
            kk = kk < bpr ? kk : bpr;
            const uint32_t t0 = s_t[kk], t1 = s_t[kk + 1];
            const uint64_t qa = s_q[t0], qb = s_q[t0 + 1];
            uint64_t found = in & (mask_of(qa == e) | mask_of(qb == e));
            const uint64_t deep = mask_of(t1 > t0 + 2u) & mask_of(qb < e);
            if (__builtin_expect(deep != 0ull, 0)) {
                bool hit = false;
                if (lanes_of(deep))
                    for (uint32_t t = t0 + 2;; ++t) {
                        const uint64_t qv = s_q[t];
                        ++steps;
                        if (t >= cnt_q) ++behind;                 // a step at or behind the padding: the padding stops a scan AT cnt_q
                        if (qv >= e || t > (uint32_t)QCAP) { hit = qv == e; break; }
                    }
                found |= in & mask_of(hit);
            }
            hits += (unsigned long long)__popcll(found) * (lane == 0 ? 1u : 0u);
        }
    }
    // (a scan that ENDS on the padding takes one step at cnt_q: those are counted apart from the ones that go on)
    atomicAdd(&out[0], steps);
    atomicAdd(&out[1], behind);
    atomicAdd(&out[2], hits);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const uint64_t nq = 1000000, max_hash = 18446744073709552ull;
    std::mt19937_64 rng(777);
    std::vector<uint64_t> q(nq + nq / 50);
    for (auto& x : q) x = rng() % max_hash + 1;
    std::sort(q.begin(), q.end());
    q.erase(std::unique(q.begin(), q.end()), q.end());
    q.resize(nq);
    const uint64_t q_max = q.back();
    uint32_t bucket_bits = 0, value_bits = 0;
    while (bucket_bits < 26 && (1ull << bucket_bits) < nq) ++bucket_bits;
    while (value_bits < 64 && (q_max >> value_bits)) ++value_bits;
    uint32_t shift = value_bits > bucket_bits ? value_bits - bucket_bits : 0;
    uint32_t buckets = (uint32_t)(q_max >> shift) + 1;
    if (buckets < nq && shift > 0) { --shift; buckets = (uint32_t)(q_max >> shift) + 1; }
    std::vector<uint32_t> t(buckets + 1);
    {
        uint64_t pos = 0;
        for (uint32_t b = 0; b <= buckets; ++b) {
            const uint64_t lo = b == buckets ? ~0ull : ((uint64_t)b << shift);
            while (pos < nq && q[pos] < lo) ++pos;
            t[b] = (uint32_t)pos;
        }
        t[buckets] = (uint32_t)nq;
    }
    uint64_t* dq; uint32_t* dt; unsigned long long* dout;
    CK(hipMalloc(&dq, (nq + 16) * 8));
    CK(hipMalloc(&dt, (buckets + 16) * 4));
    CK(hipMalloc(&dout, 64));
    CK(hipMemcpy(dq, q.data(), nq * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, t.data(), (buckets + 1) * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)walk<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)walk<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    int n_cu = 256;
    CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# nq %llu buckets %u shift %u; %d workgroups; steps `behind` = scan steps at or past the slice's end (a scan that ends ON the padding takes one)\n",
           (unsigned long long)nq, buckets, shift, n_cu);
    const uint32_t bprs[] = {10240, 9663, 8192, 8160};
    for (uint32_t bpr : bprs) {
        const uint32_t n_ranges = (buckets + bpr - 1) / bpr;
        for (int form = 0; form < 2; ++form) {
            float best = 1e9f;
            unsigned long long res[3] = {0, 0, 0};
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(dout, 0, 64));
                CK(hipEventRecord(e0));
                if (form == 0) hipLaunchKernelGGL(walk<0>, dim3(n_cu), dim3(THREADS), LDS_BYTES, 0, dq, dt, buckets, shift, n_ranges, bpr, dout);
                else hipLaunchKernelGGL(walk<1>, dim3(n_cu), dim3(THREADS), LDS_BYTES, 0, dq, dt, buckets, shift, n_ranges, bpr, dout);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
                CK(hipMemcpy(res, dout, 24, hipMemcpyDeviceToHost));
            }
            printf("bpr %5u form %d (%s): %.3f ms, scan steps %llu, steps at/behind the slice's end %llu, hits %llu\n", bpr, form,
                   form == 0 ? "guarded store + two padding threads" : "every slot, select", best, res[0], res[1], res[2]);
        }
    }
    return 0;
}
