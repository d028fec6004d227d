// sparse_pairs.hip -- all-pairs compare through an inverted view of the collection.
//
// Same u32 |A ∩ B| matrix as compare.hip / bitindex.hip (reference: minhash.rs:539-558 count_common over the
// N(N-1)/2 loop of compare.py:36-54), third cost model.  Sort every (hash, row) of the collection by hash (one
// device radix sort of the whole CSR): the rows holding one hash sit next to each other (ascending, the sort is
// stable), so
//   * a RARE hash -- one held by at most T sketches -- contributes 1 to each of its m(m-1)/2 pairs: a lane per
//     (hash, row) element walks the rest of its run and increments the matrix (atomics; sum over hashes of m^2);
//   * a FREQUENT hash becomes one bit column; all frequent hashes together form bit rows of U_f bits per sketch
//     and go through bitindex.hip's popcount(AND) kernel.
// For collections of mostly unrelated genomes nearly every hash is rare with m = 1 or 2, and the whole compare
// costs about one sort of the collection instead of N * sum(n) merge steps; for collections drawn from one pool
// every hash is frequent and this degenerates to the bit-row path.  T is where the two costs per hash meet.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>
#include "device_api.hpp"

namespace smg {

namespace {

__global__ __launch_bounds__(256) void row_ids_kernel(const uint64_t* __restrict__ offsets, uint32_t n,
                                                      uint32_t* __restrict__ rows) {
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    for (uint64_t r = wave; r < n; r += n_waves)
        for (uint64_t i = offsets[r] + lane; i < offsets[r + 1]; i += 64) rows[i] = (uint32_t)r;
}

// per run: frequent flag, and the number of pair increments the rare runs will cost
__global__ __launch_bounds__(256) void classify_runs_kernel(const uint32_t* __restrict__ counts,
                                                            const uint64_t* __restrict__ d_n_runs, uint32_t threshold,
                                                            uint32_t* __restrict__ freq_flag,
                                                            unsigned long long* __restrict__ rare_pairs) {
    const uint64_t n_runs = *d_n_runs;
    unsigned long long pairs = 0;
    for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n_runs; u += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t m = counts[u];
        const bool freq = m > threshold;
        freq_flag[u] = freq ? 1u : 0u;
        if (!freq) pairs += (unsigned long long)m * (m - 1) / 2;
    }
    for (int off = 32; off > 0; off >>= 1) pairs += __shfl_down(pairs, off);
    if ((threadIdx.x & 63) == 0 && pairs) atomicAdd(rare_pairs, pairs);
}

// one lane per run: rare runs record their end for every element; frequent runs set their bit column
__global__ __launch_bounds__(256) void runs_apply_kernel(const uint64_t* __restrict__ run_off, const uint32_t* __restrict__ freq_flag,
                                                         const uint64_t* __restrict__ freq_rank, uint64_t n_runs,
                                                         const uint32_t* __restrict__ rows, uint32_t* __restrict__ run_end,
                                                         uint32_t* __restrict__ bits, uint32_t words_per_row) {
    for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n_runs; u += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t lo = run_off[u], hi = run_off[u + 1];
        if (freq_flag[u]) {
            const uint64_t f = freq_rank[u];
            for (uint64_t p = lo; p < hi; ++p) {
                run_end[p] = 0;                                   // not a rare element
                atomicOr(&bits[(uint64_t)rows[p] * words_per_row + (f >> 5)], 1u << (f & 31));
            }
        } else {
            for (uint64_t p = lo; p < hi; ++p) run_end[p] = (uint32_t)hi;
        }
    }
}

// the same, one lane per ELEMENT (its run found by binary search in run_off): for collections whose runs are long
// (few distinct hashes, each held by many sketches) a lane per run would serialise thousands of atomics
__global__ __launch_bounds__(256) void elements_apply_kernel(const uint64_t* __restrict__ run_off, const uint32_t* __restrict__ freq_flag,
                                                             const uint64_t* __restrict__ freq_rank, uint64_t n_runs, uint64_t total,
                                                             const uint32_t* __restrict__ rows, uint32_t* __restrict__ run_end,
                                                             uint32_t* __restrict__ bits, uint32_t words_per_row) {
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t lo = 0, hi = n_runs;                             // last run with run_off[u] <= p
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (run_off[mid] <= p) lo = mid; else hi = mid;
        }
        if (freq_flag[lo]) {
            const uint64_t f = freq_rank[lo];
            run_end[p] = 0;
            atomicOr(&bits[(uint64_t)rows[p] * words_per_row + (f >> 5)], 1u << (f & 31));
        } else {
            run_end[p] = (uint32_t)run_off[lo + 1];
        }
    }
}

// one lane per (hash, row) element of a rare run: +1 on the diagonal, +1 for every later row of the run, in the
// rows this launch owns (16-row tiles rb_first, rb_first + rb_stride, ...; output rows back to back)
__global__ __launch_bounds__(256) void rare_pairs_kernel(const uint32_t* __restrict__ rows, const uint32_t* __restrict__ run_end,
                                                         uint64_t total, uint32_t n, uint32_t rb_first, uint32_t rb_stride,
                                                         uint32_t rb_count, uint32_t* __restrict__ common, uint32_t upper_only) {
    auto local_row = [&](uint32_t r) -> uint32_t {               // 0xffffffff if the row is not owned
        const uint32_t tile = r >> 4;
        if (tile < rb_first) return 0xffffffffu;
        const uint32_t d = tile - rb_first;
        if (d % rb_stride) return 0xffffffffu;
        const uint32_t t = d / rb_stride;
        return t < rb_count ? t * 16 + (r & 15) : 0xffffffffu;
    };
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t e = run_end[p];
        if (e == 0) continue;
        const uint32_t r = rows[p];
        const uint32_t lr = local_row(r);
        if (lr != 0xffffffffu) atomicAdd(&common[(uint64_t)lr * n + r], 1u);
        for (uint64_t q = p + 1; q < e; ++q) {
            const uint32_t c = rows[q];
            if (upper_only) {                                     // the caller mirrors: only the entry above the diagonal
                const uint32_t lo = r < c ? r : c, hi = r < c ? c : r;
                const uint32_t ll = lo == r ? lr : local_row(lo);
                if (ll != 0xffffffffu) atomicAdd(&common[(uint64_t)ll * n + hi], 1u);
                continue;
            }
            if (lr != 0xffffffffu) atomicAdd(&common[(uint64_t)lr * n + c], 1u);
            const uint32_t lc = local_row(c);
            if (lc != 0xffffffffu) atomicAdd(&common[(uint64_t)lc * n + r], 1u);
        }
    }
}

size_t align256(size_t x) { return (x + 255) / 256 * 256; }
unsigned grid_for(uint64_t n, unsigned per = 256, unsigned cap = 8192) {
    const uint64_t b = (n + per - 1) / per;
    return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

size_t inverted_temp_bytes(uint64_t total) {
    size_t a = 0, b = 0, c = 0;
    (void)rocprim::radix_sort_pairs(nullptr, a, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                    (size_t)total, 0u, 64u, (hipStream_t)0);
    (void)rocprim::run_length_encode(nullptr, b, (uint64_t*)nullptr, (unsigned int)total, (uint64_t*)nullptr, (uint32_t*)nullptr,
                                     (uint64_t*)nullptr, (hipStream_t)0);
    (void)rocprim::exclusive_scan(nullptr, c, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t)0, (size_t)total + 1,
                                  rocprim::plus<uint64_t>(), (hipStream_t)0);
    size_t m = a > b ? a : b;
    return align256(m > c ? m : c) + 256;
}

// (hash, row) of the whole CSR sorted by hash -> d_rows_sorted; runs: d_counts[u] elements each, *d_n_runs of them.
// d_keys_a / d_keys_b: total u64 of scratch each; d_rows_tmp: total u32 of scratch.
hipError_t inverted_sort_launch(const uint64_t* d_hashes, const uint64_t* d_offsets, uint32_t n, uint64_t total,
                                uint64_t* d_keys_a, uint64_t* d_keys_b, uint32_t* d_rows_tmp, uint32_t* d_rows_sorted,
                                uint32_t* d_counts, uint64_t* d_n_runs, void* d_temp, size_t temp_bytes, hipStream_t stream) {
    if (total == 0 || total > 0xffffffffull) return hipErrorInvalidValue;
    hipError_t e = hipMemcpyAsync(d_keys_a, d_hashes, total * 8, hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(row_ids_kernel, dim3(grid_for(((uint64_t)n + 3) / 4, 1)), dim3(256), 0, stream, d_offsets, n, d_rows_tmp);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    size_t tb = temp_bytes;
    e = rocprim::radix_sort_pairs(d_temp, tb, d_keys_a, d_keys_b, d_rows_tmp, d_rows_sorted, (size_t)total, 0u, 64u, stream);
    if (e != hipSuccess) return e;
    tb = temp_bytes;
    return rocprim::run_length_encode(d_temp, tb, d_keys_b, (unsigned int)total, d_keys_a /* unique keys: not needed later */,
                                      d_counts, d_n_runs, stream);
}

__global__ void publish_kernel(const uint64_t* __restrict__ d_n_runs, const uint64_t* __restrict__ freq_rank,
                               unsigned long long* __restrict__ out) {
    out[0] = *d_n_runs;                 // distinct hashes
    out[1] = freq_rank[*d_n_runs];      // frequent ones
}

// Everything the host has to know in ONE read-back: d_out[0] = number of runs (distinct hashes), d_out[1] = frequent
// runs, d_out[2] = pair increments of the rare runs (d_out[2] must be zero on entry).  The number of runs stays on the
// device: flags and scans cover max_runs (= total elements) entries, of which only the first *d_n_runs mean anything.
// d_run_off[u] = first element of run u, d_freq_rank[u] = index of run u among the frequent runs.
hipError_t inverted_classify_launch(const uint32_t* d_counts, const uint64_t* d_n_runs, uint64_t max_runs, uint32_t threshold,
                                    uint32_t* d_freq_flag, uint64_t* d_run_off, uint64_t* d_freq_rank, unsigned long long* d_out,
                                    void* d_temp, size_t temp_bytes, hipStream_t stream) {
    if (max_runs == 0) return hipSuccess;
    hipLaunchKernelGGL(classify_runs_kernel, dim3(grid_for(max_runs)), dim3(256), 0, stream, d_counts, d_n_runs, threshold,
                       d_freq_flag, d_out + 2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    size_t tb = temp_bytes;
    e = rocprim::exclusive_scan(d_temp, tb, d_counts, d_run_off, (uint64_t)0, (size_t)max_runs + 1, rocprim::plus<uint64_t>(), stream);
    if (e != hipSuccess) return e;
    tb = temp_bytes;
    e = rocprim::exclusive_scan(d_temp, tb, d_freq_flag, d_freq_rank, (uint64_t)0, (size_t)max_runs + 1, rocprim::plus<uint64_t>(),
                                stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(1), 0, stream, d_n_runs, d_freq_rank, d_out);
    return hipGetLastError();
}

hipError_t inverted_apply_launch(const uint64_t* d_run_off, const uint32_t* d_freq_flag, const uint64_t* d_freq_rank,
                                 uint64_t n_runs, uint64_t total, const uint32_t* d_rows_sorted, uint32_t* d_run_end,
                                 uint32_t* d_bits, uint32_t words_per_row, hipStream_t stream) {
    if (n_runs == 0) return hipSuccess;
    if (total / n_runs >= 8) {                                    // long runs: a lane per element
        hipLaunchKernelGGL(elements_apply_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0, stream, d_run_off, d_freq_flag,
                           d_freq_rank, n_runs, total, d_rows_sorted, d_run_end, d_bits, words_per_row);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(runs_apply_kernel, dim3(grid_for(n_runs)), dim3(256), 0, stream, d_run_off, d_freq_flag, d_freq_rank,
                       n_runs, d_rows_sorted, d_run_end, d_bits, words_per_row);
    return hipGetLastError();
}

hipError_t rare_pairs_launch(const uint32_t* d_rows_sorted, const uint32_t* d_run_end, uint64_t total, uint32_t n,
                             uint32_t rb_first, uint32_t rb_stride, uint32_t rb_count, uint32_t* d_common, hipStream_t stream,
                             bool upper_only) {
    if (total == 0 || rb_count == 0) return hipSuccess;
    hipLaunchKernelGGL(rare_pairs_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0, stream, d_rows_sorted, d_run_end, total,
                       n, rb_first, rb_stride < 1 ? 1 : rb_stride, rb_count, d_common, upper_only ? 1u : 0u);
    return hipGetLastError();
}

}  // namespace smg
