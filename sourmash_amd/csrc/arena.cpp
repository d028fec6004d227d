// Device arena (see arena.hpp): size-keyed cache of driver blocks with event-ordered reuse across streams.
#include "arena.hpp"
#include <stdlib.h>
#include <chrono>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace smg {
namespace {

struct Block {
    void* p = nullptr;
    size_t size = 0;
    int device = 0;
    hipEvent_t ev = nullptr;        // recorded on `tag` when the block was released
    hipStream_t tag = nullptr;
    bool ev_valid = false;          // false: released without a usable event -> a reuse synchronises the device
    bool fresh = true;              // never used since it came from the driver
};

struct Arena {
    std::mutex mu;
    std::unordered_map<void*, Block> live;
    std::multimap<size_t, Block> cached;          // all devices; a block is matched on its device too
    std::vector<std::pair<void*, size_t>> pinned_cached;
    std::unordered_map<void*, size_t> pinned_live;
    ArenaStats st{};
    uint64_t cache_max = 64ull << 30;
    Arena() {
        if (const char* e = getenv("SMG_ARENA_CACHE_MAX")) cache_max = strtoull(e, nullptr, 10);
    }
};

Arena& A() {
    static Arena* a = new Arena();      // leaked on purpose: HIP may be gone when static destructors run
    return *a;
}

size_t round_size(size_t bytes) {
    if (bytes < 512) return 512;
    if (bytes < (1u << 20)) return (bytes + 511) & ~(size_t)511;
    return (bytes + ((2u << 20) - 1)) & ~(size_t)((2u << 20) - 1);
}

// how much larger than the request a cached block may be and still be taken
size_t slack(size_t want) {
    if (want < (1u << 20)) return want;                 // small: up to 2x
    const size_t q = want / 4;
    return q > ((size_t)8 << 20) ? q : ((size_t)8 << 20);
}

uint64_t now_ns() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void driver_free(Arena& a, Block& b) {
    const uint64_t t0 = now_ns();
    if (b.ev) (void)hipEventDestroy(b.ev);
    (void)hipFree(b.p);
    a.st.driver_ns += now_ns() - t0;
    a.st.driver_frees++;
}

// mutex held
void trim_locked(Arena& a, uint64_t keep) {
    while (a.st.cached_bytes > keep && !a.cached.empty()) {
        auto it = std::prev(a.cached.end());            // largest first
        a.st.cached_bytes -= it->second.size;
        driver_free(a, it->second);
        a.cached.erase(it);
    }
}

}  // namespace

hipError_t arena_alloc(void** out, size_t bytes, hipStream_t stream) {
    Arena& a = A();
    const size_t want = round_size(bytes);
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::unique_lock<std::mutex> lk(a.mu);
    for (auto it = a.cached.lower_bound(want); it != a.cached.end() && it->first <= want + slack(want); ++it) {
        if (it->second.device != dev) continue;
        Block b = it->second;
        a.cached.erase(it);
        a.st.cached_bytes -= b.size;
        a.st.live_bytes += b.size;
        a.st.reuse_hits++;
        const bool same = b.fresh || (b.ev_valid && b.tag == stream);
        a.live.emplace(b.p, b);
        lk.unlock();
        if (!same) {
            hipError_t e = hipSuccess;
            if (b.ev_valid) e = hipStreamWaitEvent(stream, b.ev, 0);
            if (!b.ev_valid || e != hipSuccess) { (void)hipGetLastError(); e = hipDeviceSynchronize(); }
            std::lock_guard<std::mutex> g(a.mu);
            a.st.cross_stream_waits++;
            if (e != hipSuccess) return e;
        }
        *out = b.p;
        return hipSuccess;
    }
    lk.unlock();
    Block b;
    b.size = want;
    b.device = dev;
    const uint64_t t0 = now_ns();
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {                              // give the cache back and try once more
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> g(a.mu);
            trim_locked(a, 0);
        }
        e = hipMalloc(&b.p, want);
    }
    const uint64_t dt = now_ns() - t0;
    std::lock_guard<std::mutex> g(a.mu);
    a.st.driver_ns += dt;
    if (e != hipSuccess) return e;
    a.st.driver_allocs++;
    a.st.live_bytes += want;
    if (a.st.live_bytes + a.st.cached_bytes > a.st.peak_bytes) a.st.peak_bytes = a.st.live_bytes + a.st.cached_bytes;
    b.fresh = false;
    a.live.emplace(b.p, b);
    *out = b.p;
    return hipSuccess;
}

void arena_free(void* p, hipStream_t stream) {
    if (!p) return;
    Arena& a = A();
    Block b;
    {
        std::lock_guard<std::mutex> g(a.mu);
        auto it = a.live.find(p);
        if (it == a.live.end()) return;                 // not ours (or freed twice): leave it alone
        b = it->second;
        a.live.erase(it);
        a.st.live_bytes -= b.size;
    }
    if (!b.ev && hipEventCreateWithFlags(&b.ev, hipEventDisableTiming) != hipSuccess) { b.ev = nullptr; (void)hipGetLastError(); }
    b.tag = stream;
    b.fresh = false;
    b.ev_valid = b.ev && hipEventRecord(b.ev, stream) == hipSuccess;
    if (!b.ev_valid) (void)hipGetLastError();
    std::lock_guard<std::mutex> g(a.mu);
    a.st.cached_bytes += b.size;
    a.cached.emplace(b.size, b);
    if (a.st.cached_bytes > a.cache_max) {
        // the blocks about to go may still be in use by enqueued work: hipFree synchronises the device itself
        trim_locked(a, a.cache_max);
    }
}

void arena_trim(uint64_t keep_bytes) {
    Arena& a = A();
    std::lock_guard<std::mutex> g(a.mu);
    trim_locked(a, keep_bytes);
    if (keep_bytes == 0) {
        for (auto& pb : a.pinned_cached) (void)hipHostFree(pb.first);
        a.pinned_cached.clear();
    }
}

ArenaStats arena_stats() {
    Arena& a = A();
    std::lock_guard<std::mutex> g(a.mu);
    return a.st;
}

hipError_t arena_pinned_alloc(void** p, size_t bytes) {
    Arena& a = A();
    const size_t want = (bytes + 4095) & ~(size_t)4095;
    {
        std::lock_guard<std::mutex> g(a.mu);
        for (size_t i = 0; i < a.pinned_cached.size(); ++i)
            if (a.pinned_cached[i].second >= want && a.pinned_cached[i].second <= 2 * want) {
                *p = a.pinned_cached[i].first;
                a.pinned_live.emplace(*p, a.pinned_cached[i].second);
                a.pinned_cached.erase(a.pinned_cached.begin() + (long)i);
                return hipSuccess;
            }
    }
    const hipError_t e = hipHostMalloc(p, want, hipHostMallocDefault);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> g(a.mu);
    a.pinned_live.emplace(*p, want);
    return hipSuccess;
}

void arena_pinned_free(void* p) {
    if (!p) return;
    Arena& a = A();
    void* drop = nullptr;
    {
        std::lock_guard<std::mutex> g(a.mu);
        auto it = a.pinned_live.find(p);
        if (it == a.pinned_live.end()) return;
        // page-locked memory is the host's scarce kind: the cache of released blocks is bounded (SMG_PINNED_CACHE_MAX, default
        // 2 GiB -- two 10,000 x 10,000 f64 result matrices, the common repeated shape, and the transfer ring; 4 GiB until round 6:
        // ADVICE r05); what does not fit goes back to the driver
        static const size_t cache_max = [] { const char* e = getenv("SMG_PINNED_CACHE_MAX"); return e ? (size_t)strtoull(e, nullptr, 10) : ((size_t)2 << 30); }();
        size_t cached = 0;
        for (const auto& pb : a.pinned_cached) cached += pb.second;
        if (cached + it->second <= cache_max) a.pinned_cached.emplace_back(p, it->second);
        else drop = p;
        a.pinned_live.erase(it);
    }
    if (drop) (void)hipHostFree(drop);
}

}  // namespace smg
