// Device arena: the library's only route to the driver's allocator for index and scratch memory.
//
// Why: an index build of a few milliseconds of kernels allocates a dozen buffers of up to gigabytes.  Going to the driver for
// them (hipMalloc / hipFree, and on some hosts also the stream-ordered pool of hipMallocAsync) cost 100-180 ms per build on
// the round-2 benchmark host (BENCH_r02 `index_build_ms` 183.9 against 6.9 ms of kernels).  Blocks obtained
// from the driver are therefore kept by the library and handed out again: after the first build of a given shape a rebuild
// makes NO driver call at all.  `smgpu_arena_trim` gives cached blocks back; `smgpu_arena_stats` reports what the arena did.
//
// Ordering rule (the one torch's caching allocator uses): a block freed with stream S may be reused at once by work enqueued
// on S; for any other stream the new user first waits (hipStreamWaitEvent) on an event recorded on S at release time.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

namespace smg {

struct ArenaStats {
    uint64_t driver_allocs;      // hipMalloc calls made
    uint64_t driver_frees;       // hipFree calls made (trim / out-of-memory recovery)
    uint64_t driver_ns;          // wall-clock nanoseconds spent inside them
    uint64_t reuse_hits;         // allocations served from cached blocks
    uint64_t live_bytes;         // handed out now
    uint64_t cached_bytes;       // held for reuse now
    uint64_t peak_bytes;         // most live + cached ever held
    uint64_t cross_stream_waits; // reuses that had to wait on another stream's event
};

// bytes may be 0 (returns a minimal block).  `stream`: the stream whose work will use the block first.
hipError_t arena_alloc(void** p, size_t bytes, hipStream_t stream);
// `stream`: the stream on which the block's last use was enqueued.  Never blocks.
void arena_free(void* p, hipStream_t stream);
// Release cached blocks to the driver until at most keep_bytes stay cached.
void arena_trim(uint64_t keep_bytes);
ArenaStats arena_stats();

// A few hundred bytes of pinned host memory for read-backs of scalars (a pageable destination makes the runtime stage the
// copy through its own bounce buffers); cached like device blocks.
hipError_t arena_pinned_alloc(void** p, size_t bytes);
void arena_pinned_free(void* p);

template <class T>
inline hipError_t arena_alloc_t(T** p, size_t bytes, hipStream_t stream) { return arena_alloc((void**)p, bytes, stream); }

// RAII scratch block (throws nothing: check .p)
struct ArenaBuf {
    void* p = nullptr;
    hipStream_t st = nullptr;
    ArenaBuf() = default;
    ArenaBuf(const ArenaBuf&) = delete;
    ArenaBuf& operator=(const ArenaBuf&) = delete;
    ~ArenaBuf() { if (p) arena_free(p, st); }
    hipError_t get(size_t bytes, hipStream_t stream) {
        if (p) { arena_free(p, st); p = nullptr; }
        st = stream;
        return arena_alloc(&p, bytes, stream);
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

}  // namespace smg
