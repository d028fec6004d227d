// Large PAGEABLE host buffers to and from the device at PCIe speed.
//
// The object API hands the library what the reference's FFI hands its Rust core: a list of sketch objects whose hash vectors
// live wherever the host allocator put them (src/sourmash/compare.py:326-358 -> src/core/src/ffi/minhash.rs), and numpy
// matrices to fill.  hipMemcpyAsync on pageable memory is staged by the driver copy by copy: round 4 issued one such copy per
// sketch (10,000 x 40 KB at config C4) and copied an 800 MB matrix back the same way -- the entry points took several
// times what their kernels did (Missing 2).  Here both directions go through a ring of two pinned chunks:
//   gather_to_device   worker threads copy the pieces of chunk c + 1 into pinned memory while chunk c travels as ONE H2D copy;
//   device_to_host     chunk c + 1 travels D2H while worker threads copy chunk c out of pinned memory into the caller's array
//                      (first touch of a fresh numpy array is page faults: that is why this side has threads too).
// A destination / source that is already pinned (hipHostMalloc / hipHostRegister) is copied directly.
// Callers hold the device context's lock (device_ctx.hpp), which serialises the use of the ring.
#pragma once
#include <hip/hip_runtime_api.h>
#if defined(__SSE2__)
#include <emmintrin.h>          // streaming stores; every other host (an aarch64 ROCm node) takes the memcpy path below
#endif
#include <sched.h>
#include <unistd.h>
#include <sys/mman.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "arena.hpp"
#include "device_ctx.hpp"

namespace smg {

struct HostPiece {            // `bytes` bytes at `src` belong at byte offset `dst_off` of the packed buffer
    const void* src;
    size_t dst_off, bytes;
};

class HostXfer {
  public:
    static constexpr size_t CHUNK = (size_t)32 << 20;      // per ring slot (64 MiB slots and 16 workers measured slower: 107 vs 93 ms for 4 GB)
    static HostXfer& get() { static HostXfer* x = new HostXfer(); return *x; }      // (leaked like the context: HIP may be gone at exit)

    struct Stats { uint64_t h2d_bytes = 0, d2h_bytes = 0, h2d_ns = 0, d2h_ns = 0, calls = 0; };
    Stats stats() const { return st_; }
    void reset_stats() { st_ = Stats(); }

    static unsigned workers() {
        static const unsigned n = [] {
            unsigned c = std::max(1u, std::thread::hardware_concurrency());
            cpu_set_t set;
            if (sched_getaffinity(0, sizeof(set), &set) == 0) c = std::min<unsigned>(c, (unsigned)CPU_COUNT(&set));
            return std::min(c, 8u);
        }();
        return n;
    }

    // pieces: sorted by dst_off, back to back from 0 to total_bytes (empty pieces allowed)
    void gather_to_device(void* d_dst, const std::vector<HostPiece>& pieces, size_t total_bytes, hipStream_t st) {
        if (total_bytes == 0) return;
        const uint64_t t0 = now_ns();
        ring();
        size_t first = 0;                                               // first piece that reaches into the chunk being filled
        for (size_t c = 0, off = 0; off < total_bytes; ++c, off += CHUNK) {
            const size_t len = std::min(CHUNK, total_bytes - off);
            const int slot = (int)(c & 1);
            if (c >= 2) hip_check(hipEventSynchronize(ev_[slot]), "event");           // the copy that last read this slot is through
            while (first < pieces.size() && pieces[first].dst_off + pieces[first].bytes <= off) ++first;
            fill(static_cast<char*>(buf_[slot]), pieces, first, off, len);
            hip_check(hipMemcpyAsync(static_cast<char*>(d_dst) + off, buf_[slot], len, hipMemcpyHostToDevice, st), "H2D");
            hip_check(hipEventRecord(ev_[slot], st), "event");
        }
        // the ring may be refilled by the next call at once: its first two chunks wait for nothing, so drain here
        hip_check(hipEventSynchronize(ev_[0]), "event");
        hip_check(hipEventSynchronize(ev_[1]), "event");
        st_.h2d_bytes += total_bytes; st_.h2d_ns += now_ns() - t0; st_.calls++;
    }

    // returns when the bytes are in h_dst
    void device_to_host(void* h_dst, const void* d_src, size_t bytes, hipStream_t st) {
        if (bytes == 0) return;
        const uint64_t t0 = now_ns();
        if (bytes < ((size_t)1 << 20) || is_pinned(h_dst)) {
            hip_check(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, st), "D2H");
            hip_check(hipStreamSynchronize(st), "sync");
            st_.d2h_bytes += bytes; st_.d2h_ns += now_ns() - t0;
            return;
        }
        ring();
        {   // a fresh numpy array is untouched pages: ask for huge ones where the kernel gives them on request (512 x fewer faults)
            const uintptr_t a = ((uintptr_t)h_dst + 4095) & ~(uintptr_t)4095, b = ((uintptr_t)h_dst + bytes) & ~(uintptr_t)4095;
            if (b > a) (void)madvise((void*)a, b - a, MADV_HUGEPAGE);
        }
        const size_t n_chunks = (bytes + CHUNK - 1) / CHUNK;
        auto issue = [&](size_t c) {
            const size_t off = c * CHUNK, len = std::min(CHUNK, bytes - off);
            hip_check(hipMemcpyAsync(buf_[c & 1], static_cast<const char*>(d_src) + off, len, hipMemcpyDeviceToHost, st), "D2H");
            hip_check(hipEventRecord(ev_[c & 1], st), "event");
        };
        issue(0);
        for (size_t c = 0; c < n_chunks; ++c) {
            if (c + 1 < n_chunks) issue(c + 1);                                       // travels while chunk c is copied out
            hip_check(hipEventSynchronize(ev_[c & 1]), "event");
            const size_t off = c * CHUNK, len = std::min(CHUNK, bytes - off);
            spread(static_cast<char*>(h_dst) + off, static_cast<const char*>(buf_[c & 1]), len);
            // (slot c & 1 is free again only now: chunk c + 2 is issued in the next iteration, after this copy-out)
        }
        st_.d2h_bytes += bytes; st_.d2h_ns += now_ns() - t0;
    }

  private:
    void* buf_[2] = {nullptr, nullptr};
    hipEvent_t ev_[2] = {nullptr, nullptr};
    Stats st_;

    static uint64_t now_ns() {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
    }
    void ring() {
        if (buf_[0]) return;
        for (int i = 0; i < 2; ++i) {
            hip_check(arena_pinned_alloc(&buf_[i], CHUNK), "pinned ring");
            hip_check(hipEventCreateWithFlags(&ev_[i], hipEventDisableTiming), "event");
        }
    }
    static bool is_pinned(const void* p) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
        return a.type == hipMemoryTypeHost;
    }
    // memcpy whose stores bypass the caches (SSE2 streaming stores, 16-byte aligned middle): the destination is a pinned slot the copy
    // engine reads next -- nothing on the host reads it again
    static void copy_streaming(char* dst, const char* src, size_t n) {
        if (n < 256) { memcpy(dst, src, n); return; }
#if !defined(__SSE2__)
        memcpy(dst, src, n);
#else
        const size_t head = (16 - ((uintptr_t)dst & 15)) & 15;
        memcpy(dst, src, head);
        dst += head; src += head; n -= head;
        const size_t blocks = n / 16;
        for (size_t i = 0; i < blocks; ++i) {
            __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src) + i);
            _mm_stream_si128(reinterpret_cast<__m128i*>(dst) + i, v);
        }
        memcpy(dst + blocks * 16, src + blocks * 16, n - blocks * 16);
#endif
    }
    // bytes [off, off + len) of the packed buffer, gathered from the pieces, by the worker threads (each takes a byte range)
    static void fill(char* dst, const std::vector<HostPiece>& pieces, size_t first, size_t off, size_t len) {
        auto part = [&](size_t lo, size_t hi) {                          // chunk-relative byte range
            size_t i = first;
            while (i < pieces.size() && pieces[i].dst_off + pieces[i].bytes <= off + lo) ++i;
            for (; i < pieces.size() && pieces[i].dst_off < off + hi; ++i) {
                const size_t a = std::max(pieces[i].dst_off, off + lo), b = std::min(pieces[i].dst_off + pieces[i].bytes, off + hi);
                if (b > a) copy_streaming(dst + (a - off), static_cast<const char*>(pieces[i].src) + (a - pieces[i].dst_off), b - a);
            }
#if defined(__SSE2__)
            _mm_sfence();                                               // the streamed lines are on their way before the copy engine is told
#endif
        };
        run(len, part);
    }
    static void spread(char* dst, const char* src, size_t len) {
        run(len, [&](size_t lo, size_t hi) { memcpy(dst + lo, src + lo, hi - lo); });
    }
    // The worker threads live as long as the library (leaked at exit like the context): a chunk is 0.6 ms of copying, and starting
    // and joining seven threads for each of the 128 chunks of a 4 GB collection was a tenth of its upload.  Share k of a job goes to
    // worker k; the caller takes share 0 and waits for the others.  (One job at a time: callers hold the device context's lock.)
    class Pool {
      public:
        explicit Pool(unsigned n) : owner_(getpid()) {
            for (unsigned k = 1; k < n; ++k) threads_.emplace_back([this, k] { loop(k); });
        }
        unsigned size() const { return (unsigned)threads_.size() + 1; }
        template <class F>
        void run(F&& share) {                                       // share(k) for k = 0 .. size() - 1
            if (threads_.empty()) { share(0); return; }
            if (getpid() != owner_) {                               // a forked child: the workers exist in the parent only --
                for (unsigned k = 0; k < size(); ++k) share(k);     // every share on the caller instead of a wait that never ends
                return;
            }
            std::function<void(unsigned)> f = std::forward<F>(share);
            std::lock_guard<std::mutex> one_job(run_m_);
            {
                std::lock_guard<std::mutex> l(m_);
                job_ = &f;
                pending_ = (unsigned)threads_.size();
                ++generation_;
            }
            start_.notify_all();
            f(0);
            std::unique_lock<std::mutex> l(m_);
            done_.wait(l, [this] { return pending_ == 0; });
            job_ = nullptr;
        }

      private:
        void loop(unsigned k) {
            uint64_t seen = 0;
            for (;;) {
                const std::function<void(unsigned)>* f;
                {
                    std::unique_lock<std::mutex> l(m_);
                    start_.wait(l, [&] { return generation_ != seen; });
                    seen = generation_;
                    f = job_;
                }
                (*f)(k);
                std::lock_guard<std::mutex> l(m_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
        const pid_t owner_;
        std::mutex m_, run_m_;
        std::condition_variable start_, done_;
        std::vector<std::thread> threads_;
        const std::function<void(unsigned)>* job_ = nullptr;
        uint64_t generation_ = 0;
        unsigned pending_ = 0;
    };
    static Pool& pool() { static Pool* p = new Pool(workers()); return *p; }

    template <class F>
    static void run(size_t len, F part) {
        if (len < ((size_t)2 << 20) || workers() == 1) { part(0, len); return; }
        Pool& p = pool();
        const unsigned t = p.size();
        const size_t per = ((len + t - 1) / t + 4095) & ~(size_t)4095;        // page-aligned shares
        p.run([&](unsigned k) {
            const size_t lo = std::min(len, per * k), hi = std::min(len, per * (k + 1));
            if (hi > lo) part(lo, hi);
        });
    }
};

}  // namespace smg
