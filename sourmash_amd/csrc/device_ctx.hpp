// DeviceCtx -- owns the HIP stream and grow-only device scratch buffers used by
// the host-pointer entry points of the C-ABI (kmerminhash_add_sequence, ...).
// The raw device-pointer entry points (smgpu_*_raw) bypass it entirely.
//
// There is deliberately NO CPU fallback: if no HIP device can be opened every
// k-mer / intersection operation raises SOURMASH_ERROR_CODE_INTERNAL.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <string.h>
#include <cmath>
#include <mutex>
#include <string>
#include <vector>
#include <list>
#include <unordered_map>
#include <chrono>
#include "arena.hpp"
#include "device_api.hpp"
#include "minhash_host.hpp"
#include "gather_api.hpp"
#include "pair_api.hpp"
#include "smg_errors.hpp"

namespace smg {

// A DNA ksize the device at hand cannot take (sketch_words.hip: the long-k kernel's window lives in LDS) is refused in words, at
// the entry points, instead of surfacing as a bare "invalid value" from a launch.
inline void check_dna_ksize(uint32_t k) {
    const uint32_t max_k = sketch_dna_max_k();
    if (k > max_k)
        throw err_internal("ksize " + std::to_string(k) + " is longer than the longest DNA k-mer this device's LDS holds (" + std::to_string(max_k) + ")");
}


inline void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess)
        throw err_internal(std::string("HIP failure in ") + what + ": " + hipGetErrorString(e));
}

// Grow-only device buffer.  Its block comes from the library's arena (arena.hpp) like every other device block of the library
// and goes back there -- never straight to the driver: hipFree synchronises the whole device, and a synchronisation while a
// resident gather loop of another object (or rank) is polling for its peers stalls that loop or deadlocks with it.
// `stream`: the stream the buffer's work is enqueued on (the arena orders reuse across streams by events).
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipStream_t st = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void reserve(size_t bytes, hipStream_t stream) {
        if (bytes <= cap) return;
        release();
        size_t want = bytes + bytes / 4 + 4096;
        hip_check(arena_alloc(&p, want, stream), "arena_alloc");
        cap = want;
        st = stream;
    }
    void release() {
        if (p) arena_free(p, st);
        p = nullptr; cap = 0;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// Scratch and index blocks come from the library's arena (arena.hpp): cached driver blocks, reused in stream order, so that
// an index build whose kernels take a millisecond does not spend a hundred in the driver's allocator.
struct AsyncBuf {
    void* p = nullptr;
    hipStream_t st = nullptr;
    AsyncBuf() = default;
    AsyncBuf(size_t bytes, hipStream_t stream) : st(stream) {
        hip_check(arena_alloc(&p, bytes + 256, stream), "arena_alloc");
    }
    AsyncBuf(const AsyncBuf&) = delete;
    AsyncBuf& operator=(const AsyncBuf&) = delete;
    ~AsyncBuf() { if (p) arena_free(p, st); }
    void reset(size_t bytes, hipStream_t stream) {
        if (p) arena_free(p, st);
        p = nullptr; st = stream;
        hip_check(arena_alloc(&p, bytes + 256, stream), "arena_alloc");
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

struct PairStats {
    uint64_t common = 0;       // |A ∩ B|
    uint64_t prod = 0;         // sum abundA*abundB over the intersection
    uint64_t a_sq = 0, b_sq = 0;
    uint64_t common_num = 0;   // num rule: intersection hashes that survive merge-and-truncate
};

class DeviceCtx {
  public:
    // the context if one was ever made (freeing a sketch must not create one, nor fail without a device)
    static DeviceCtx*& instance() { static DeviceCtx* ctx = nullptr; return ctx; }
    static DeviceCtx* peek() { return instance(); }
    // One context per process (leaked on purpose: HIP may already be torn down at exit).
    static DeviceCtx& get() {
        DeviceCtx*& ctx = instance();
        static std::mutex mu;
        std::lock_guard<std::mutex> g(mu);
        if (!ctx) {
            int n = 0;
            hipError_t e = hipGetDeviceCount(&n);
            if (e != hipSuccess || n <= 0)
                throw err_internal("no HIP device available: sourmash_amd has no CPU fallback for k-mer hashing "
                                   "or sketch intersection (hipGetDeviceCount: " +
                                   std::string(e == hipSuccess ? "0 devices" : hipGetErrorString(e)) + ")");
            ctx = new DeviceCtx();
            hip_check(hipStreamCreate(&ctx->stream_), "hipStreamCreate");
            ctx->scalars_.reserve(256, ctx->stream_);
        }
        return *ctx;
    }

    // recursive: an entry point that holds the context may reach settle() of a sketch with queued records (capi.cpp), which
    // takes it again on the same thread
    std::recursive_mutex& mutex() { return mu_; }
    hipStream_t stream() const { return stream_; }

    // ---- sketch a host buffer: sorted unique kept hashes (+ multiplicities) -------------------
    // thr: keep 1 <= h <= thr.  limit: only the `limit` smallest are needed (num sketches; 0 = all).
    void sketch_host(const uint8_t* seq, size_t len, uint32_t k, uint64_t seed, uint64_t thr, bool want_counts,
                     size_t limit, std::vector<uint64_t>& hashes, std::vector<uint64_t>& counts) {
        hashes.clear(); counts.clear();
        if (len < k || k == 0) return;
        // bounded chunks (k-1 overlap) so scratch stays modest whatever the record length
        const size_t CHUNK = (size_t)256 << 20;
        KmerMinHash acc;                       // merges chunk results (plain sorted-set union with counts)
        acc.max_hash = 0; acc.num = limit ? (uint32_t)limit : 0; acc.track_abundance = want_counts;
        bool multi = len > CHUNK;
        std::vector<uint64_t> hs, cs;
        for (size_t off = 0; off < len - (k - 1); off += CHUNK) {
            const size_t n = std::min(len - off, CHUNK + (size_t)(k - 1));
            sketch_chunk(seq + off, n, k, seed, thr, want_counts, limit, hs, cs);
            if (!multi) { hashes.swap(hs); counts.swap(cs); return; }
            if (acc.num == 0 && acc.max_hash == 0) acc.max_hash = UINT64_MAX;   // make the accumulator accept everything
            acc.add_sorted_batch(hs.data(), want_counts ? cs.data() : nullptr, hs.size());
        }
        hashes.swap(acc.mins);
        if (want_counts) counts.swap(acc.abunds);
    }

    // ---- per-k-mer hashes of a host buffer (0 for bad k-mers) ----------------------------------
    void kmer_hashes_host(const uint8_t* seq, size_t len, uint32_t k, uint64_t seed, std::vector<uint64_t>& out) {
        out.clear();
        if (len < k || k == 0) return;
        const size_t nk = len - k + 1;
        upload_seq(seq, len);
        out_.reserve(nk * 8, stream_);
        hip_check(hipMemsetAsync(out_.p, 0, nk * 8, stream_), "memset");
        check_dna_ksize(k);
        hip_check(kmer_hashes_launch(seq_.as<uint8_t>(), len, k, seed, out_.as<uint64_t>(), nk, stream_), "kmer_hashes");
        out.resize(nk);
        hip_check(hipMemcpyAsync(out.data(), out_.p, nk * 8, hipMemcpyDeviceToHost, stream_), "D2H");
        hip_check(hipStreamSynchronize(stream_), "sync");
    }

    // ---- position of the first byte outside ACGTacgt, or SIZE_MAX ------------------------------
    size_t first_invalid_host(const uint8_t* seq, size_t len) {
        if (len == 0) return SIZE_MAX;
        upload_seq(seq, len);
        unsigned long long* d = scalars_.as<unsigned long long>();
        hip_check(hipMemsetAsync(d, 0xff, 8, stream_), "memset");
        hip_check(first_invalid_launch(seq_.as<uint8_t>(), len, d, stream_), "first_invalid");
        unsigned long long h = 0;
        hip_check(hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, stream_), "D2H");
        hip_check(hipStreamSynchronize(stream_), "sync");
        return h == ~0ull ? SIZE_MAX : (size_t)h;
    }

    // ---- pair statistics of two sorted sketches -------------------------------------------------
    // want_list: also return the sorted intersection.  num != 0: apply the bottom-k rule.
    // ---- device mirrors of sketches (per-pair API) ---------------------------------------------------
    // The reference's per-pair entry points (ffi/minhash.rs:409-457) are called in loops over the same sketches
    // (compare.py:36-54, index/__init__.py:115-170): uploading both sketches on every call made a pair cost 54-59 us
    // where the two-pointer walk on a host core takes 7 (profiles/r02_small_calls.json).  A sketch's hashes (and
    // abundances) stay on the device under its content generation (minhash_host.hpp: gen), least recently used first out.
    struct Mirror {
        uint64_t* mins = nullptr;
        uint64_t* abunds = nullptr;
        size_t n = 0;
        uint64_t first = 0, last = 0;
        std::list<uint64_t>::iterator lru;
    };
    static constexpr size_t MIRROR_BYTES_MAX = (size_t)512 << 20;
    static constexpr size_t MIRROR_ENTRIES_MAX = 4096;

    Mirror& mirror_of(const KmerMinHash& m, bool want_abund) {
        if (m.mirrored_gen && m.mirrored_gen != m.gen) drop_mirror(mirrors_.find(m.mirrored_gen));   // the content it had before
        m.mirrored_gen = m.gen;
        auto it = mirrors_.find(m.gen);
        if (it != mirrors_.end()) {
            Mirror& mr = it->second;
            // (size and end hashes are checked as well: a writer that forgot touch() costs a re-upload, not a wrong answer)
            if (mr.n == m.size() && (mr.n == 0 || (mr.first == m.mins.front() && mr.last == m.mins.back()))) {
                lru_.splice(lru_.begin(), lru_, mr.lru);
                if (want_abund && !mr.abunds && mr.n) upload(&mr.abunds, m.abunds.data(), mr.n);
                return mr;
            }
            drop_mirror(it);
        }
        // (the most recently used mirror stays: it may be the other operand of the call being served)
        while (lru_.size() > 1 && (mirror_bytes_ + m.size() * 16 > MIRROR_BYTES_MAX || mirrors_.size() >= MIRROR_ENTRIES_MAX))
            drop_mirror(mirrors_.find(lru_.back()));
        Mirror mr;
        mr.n = m.size();
        if (mr.n) {
            mr.first = m.mins.front(); mr.last = m.mins.back();
            upload(&mr.mins, m.mins.data(), mr.n);
            if (want_abund) upload(&mr.abunds, m.abunds.data(), mr.n);
        }
        lru_.push_front(m.gen);
        mr.lru = lru_.begin();
        return mirrors_.emplace(m.gen, mr).first->second;
    }
    void upload(uint64_t** dst, const uint64_t* src, size_t n) {
        hip_check(arena_alloc((void**)dst, n * 8 + 64, stream_), "arena_alloc");
        hip_check(hipMemcpyAsync(*dst, src, n * 8, hipMemcpyHostToDevice, stream_), "H2D");
        mirror_bytes_ += n * 8;
    }
    // a sketch is being freed: its mirror goes with it (a copy that shares the content re-uploads on its next use)
    void forget(const KmerMinHash& m) {
        if (m.mirrored_gen) drop_mirror(mirrors_.find(m.mirrored_gen));
    }
    // Copies share a content generation (a clone of a mirrored sketch finds the same mirror), so the "stale" mirror one operand
    // lets go of can be the very block the other operand of the same call was just handed: its blocks are then parked until the
    // call's kernels are on the stream (a block freed and re-allocated on one stream would be overwritten by the second upload
    // in front of the launch).  `hold_gen_` is the generation the call in progress already holds pointers of.
    void drop_mirror(std::unordered_map<uint64_t, Mirror>::iterator it) {
        if (it == mirrors_.end()) return;
        Mirror& mr = it->second;
        const bool park = hold_gen_ != 0 && it->first == hold_gen_;
        if (mr.mins) { if (park) parked_.push_back(mr.mins); else arena_free(mr.mins, stream_); mirror_bytes_ -= mr.n * 8; }
        if (mr.abunds) { if (park) parked_.push_back(mr.abunds); else arena_free(mr.abunds, stream_); mirror_bytes_ -= mr.n * 8; }
        lru_.erase(mr.lru);
        mirrors_.erase(it);
    }
    // the previous call's kernels are on the stream: what it parked can go back to the arena (stream-ordered reuse)
    void release_parked() {
        for (void* p : parked_) arena_free(p, stream_);
        parked_.clear();
        hold_gen_ = 0;
    }
    struct HoldScope {                      // a pair call: parked blocks of the call before go first, the hold ends with the call
        DeviceCtx& c;
        explicit HoldScope(DeviceCtx& ctx) : c(ctx) { c.release_parked(); }
        ~HoldScope() { c.hold_gen_ = 0; }
    };

    // |a ∩ b| through the one-launch kernel with the result polled from a pinned slot; false: not applicable here
    bool pair_count_fast(const KmerMinHash& a, const KmerMinHash& b, uint64_t* common) {
        const KmerMinHash& A = a.size() <= b.size() ? a : b;
        const KmerMinHash& B = a.size() <= b.size() ? b : a;
        if (A.size() == 0) { *common = 0; return true; }
        if (B.size() > PAIR_SMALL_MAX) return false;
        if (!slot_) {
            if (hipHostMalloc((void**)&slot_, 256, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); slot_ = nullptr; return false; }
            slot_[0] = slot_[1] = 0;
        }
        HoldScope hold(*this);
        const uint64_t* dA = mirror_of(A, false).mins;
        hold_gen_ = A.gen;                                             // B may share, or have shared, A's generation
        const uint64_t* dB = mirror_of(B, false).mins;
        const unsigned long long seq = ++slot_seq_;
        volatile unsigned long long* vs = slot_;
        hip_check(pair_count_small_launch(dA, A.size(), dB, B.size(), slot_, seq, stream_), "pair_count");
        // the kernel stores the count, then the sequence number (system-scope release): poll, with a synchronise as the way out
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0; vs[0] != seq; ++spins) {
            __builtin_ia32_pause();
            if ((spins & 1023u) == 1023u &&
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > 2000.0) {
                hip_check(hipStreamSynchronize(stream_), "sync");
                break;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (vs[0] != seq) throw err_internal("pair kernel did not publish its result");
        *common = vs[1];
        return true;
    }

    PairStats pair(const KmerMinHash& a, const KmerMinHash& b, bool want_abund, bool want_list, uint64_t num,
                   std::vector<uint64_t>* list) {
        PairStats st;
        if (list) list->clear();
        if (!want_abund && !want_list && num == 0 && pair_count_fast(a, b, &st.common)) return st;
        // search the shorter list's hashes in the longer one (minhash.rs:550-554 swaps likewise)
        const KmerMinHash& A = a.size() <= b.size() ? a : b;
        const KmerMinHash& B = a.size() <= b.size() ? b : a;
        const size_t na = A.size(), nb = B.size();
        const bool ab = want_abund && a.track_abundance && b.track_abundance;
        // the sketches come from their device mirrors (uploaded on first use, kept until the sketch changes); scratch: [I list]
        pair_.reserve((na + 16) * 8, stream_);
        HoldScope hold(*this);
        Mirror& mA = mirror_of(A, ab);
        const uint64_t* dA = mA.mins;
        const uint64_t* dAa = mA.abunds;
        hold_gen_ = A.gen;                                             // B may share, or have shared, A's generation: A's blocks stay
        Mirror& mB = mirror_of(B, ab);                                 // (never evicts A: A was just used)
        const uint64_t* dB = mB.mins;
        const uint64_t* dBa = mB.abunds;
        uint64_t* dI = pair_.as<uint64_t>();
        flags_.reserve(na + 16, stream_);
        unsigned long long* sums = scalars_.as<unsigned long long>();   // [0..3] sums, [4] list size, [5] num count
        hip_check(hipMemsetAsync(sums, 0, 64, stream_), "memset");
        const bool need_list = want_list || num != 0;
        hip_check(pair_match_launch(dA, na, dB, nb, ab ? dAa : nullptr, ab ? dBa : nullptr,
                                    need_list ? flags_.as<uint8_t>() : nullptr, sums, 0, stream_), "pair_match");
        if (ab) {
            hip_check(sumsq_launch(dAa, na, sums + 2, stream_), "sumsq");
            hip_check(sumsq_launch(dBa, nb, sums + 3, stream_), "sumsq");
        }
        if (need_list && na && nb) {
            const size_t tb = select_temp_bytes(na);
            temp_.reserve(tb, stream_);
            hip_check(select_flagged(dA, flags_.as<uint8_t>(), na, dI, (uint64_t*)(sums + 4), temp_.p, tb, stream_), "select");
        }
        unsigned long long h[8] = {0};
        hip_check(hipMemcpyAsync(h, sums, 64, hipMemcpyDeviceToHost, stream_), "D2H");
        hip_check(hipStreamSynchronize(stream_), "sync");
        st.common = h[0]; st.prod = h[1];
        // sums[2]/[3] belong to the (possibly swapped) A/B
        const bool swapped = &A != &a;
        st.a_sq = swapped ? h[3] : h[2];
        st.b_sq = swapped ? h[2] : h[3];
        const uint64_t ni = need_list ? h[4] : 0;
        if (num != 0 && ni) {
            hip_check(num_rank_launch(dI, ni, dA, na, dB, nb, num, sums + 5, stream_), "num_rank");
            hip_check(hipMemcpyAsync(h, sums, 64, hipMemcpyDeviceToHost, stream_), "D2H");
            hip_check(hipStreamSynchronize(stream_), "sync");
            st.common_num = h[5];
        }
        if (want_list && list && ni) {
            list->resize(ni);
            hip_check(hipMemcpyAsync(list->data(), dI, ni * 8, hipMemcpyDeviceToHost, stream_), "D2H");
            hip_check(hipStreamSynchronize(stream_), "sync");
        }
        return st;
    }

    // ---- protein / dayhoff / hp sketches (protein.hip) ----------------------------------------------
    // Residue k-mers of `seq` (residues if is_protein, else DNA translated in six frames): sorted unique kept
    // hashes (+ multiplicities).  ksize is the stored ksize (3 x residues).
    void protein_sketch_host(const uint8_t* seq, size_t len, uint32_t ksize, uint32_t hf, uint64_t seed, bool is_protein,
                             uint64_t thr, bool want_counts, size_t limit, std::vector<uint64_t>& hashes,
                             std::vector<uint64_t>& counts) {
        hashes.clear(); counts.clear();
        const uint32_t k = ksize / 3;
        const size_t n_aa = residues_to_device(seq, len, k, hf, is_protein);
        if (n_aa < k) return;
        const size_t nw = n_aa - k + 1;
        const double expect = (double)nw * ((double)thr / 18446744073709551616.0);
        size_t cap = (size_t)(expect * 1.5 + 8.0 * std::sqrt(expect + 1.0)) + 4096;
        if (cap > nw) cap = nw;
        unsigned long long* d_cnt = scalars_.as<unsigned long long>();
        unsigned long long kept = 0;
        for (int attempt = 0; attempt < 2; ++attempt) {
            out_.reserve(cap * 8, stream_);
            hip_check(hipMemsetAsync(d_cnt, 0, 16, stream_), "memset");
            hip_check(residue_windows_launch(aa_.as<uint8_t>(), n_aa, k, seed, thr, out_.as<uint64_t>(), d_cnt, cap, false,
                                             stream_), "residue_windows");
            hip_check(hipMemcpyAsync(&kept, d_cnt, 8, hipMemcpyDeviceToHost, stream_), "D2H");
            hip_check(hipStreamSynchronize(stream_), "sync");
            if (kept <= cap) break;
            cap = (size_t)kept;
        }
        if (kept == 0) return;
        sorted_unique_to_host(kept, thr, want_counts, limit, hashes, counts);
    }

    // One hash per residue window in the reference's iteration order (signature.rs:307-393): for translated
    // DNA frame 0 forward, frame 0 reverse complement, frame 1 forward, ...
    void protein_hashes_host(const uint8_t* seq, size_t len, uint32_t ksize, uint32_t hf, uint64_t seed, bool is_protein,
                             std::vector<uint64_t>& out) {
        out.clear();
        const uint32_t k = ksize / 3;
        const size_t n_aa = residues_to_device(seq, len, k, hf, is_protein);
        if (n_aa < k) return;
        const size_t nw = n_aa - k + 1;
        out_.reserve(nw * 8, stream_);
        hip_check(hipMemsetAsync(out_.p, 0, nw * 8, stream_), "memset");
        hip_check(residue_windows_launch(aa_.as<uint8_t>(), n_aa, k, seed, ~0ull, out_.as<uint64_t>(), nullptr, nw, true,
                                         stream_), "residue_windows");
        std::vector<uint64_t> dense(nw);
        hip_check(hipMemcpyAsync(dense.data(), out_.p, nw * 8, hipMemcpyDeviceToHost, stream_), "D2H");
        hip_check(hipStreamSynchronize(stream_), "sync");
        if (is_protein) { out.swap(dense); return; }
        size_t start = 0;                                          // drop the windows that straddle a separator
        for (int s = 0; s < 6; ++s) {
            const size_t na = (len - (size_t)(s >> 1)) / 3;
            if (na >= k) out.insert(out.end(), dense.begin() + start, dense.begin() + start + (na - k + 1));
            start += na + 1;
        }
    }

  private:
    DeviceCtx() = default;

    // residues (mapped to the sketch's alphabet) of `seq` on the device -> number of bytes in aa_ (0: nothing to hash)
    size_t residues_to_device(const uint8_t* seq, size_t len, uint32_t k, uint32_t hf, bool is_protein) {
        if (k == 0 || len < k) return 0;                             // signature.rs:199-210
        if (!is_protein && len < (size_t)k * 3) return 0;           // signature.rs:259-261
        upload_seq(seq, len);
        if (is_protein) {
            aa_.reserve(len + 64, stream_);
            hip_check(residues_launch(seq_.as<uint8_t>(), len, hf, aa_.as<uint8_t>(), stream_), "residues");
            return len;
        }
        const size_t total = (size_t)translated_bytes(len);
        aa_.reserve(total + 64, stream_);
        hip_check(translate_launch(seq_.as<uint8_t>(), len, hf, aa_.as<uint8_t>(), stream_), "translate");
        return total;
    }

    // out_[0,kept) -> sorted unique (+ counts) on the host, truncated to `limit` if non-zero
    void sorted_unique_to_host(unsigned long long kept, uint64_t thr, bool want_counts, size_t limit,
                               std::vector<uint64_t>& hashes, std::vector<uint64_t>& counts) {
        unsigned long long* d_cnt = scalars_.as<unsigned long long>();
        const size_t tb = sort_unique_temp_bytes(kept);
        temp_.reserve(tb, stream_);
        uniq_.reserve((size_t)kept * 16 + 64, stream_);
        uint64_t* d_u = uniq_.as<uint64_t>();
        uint64_t* d_c = d_u + kept;
        int bits = 64;
        if (thr != ~0ull) { bits = 1; while (bits < 64 && (thr >> bits)) ++bits; }
        hip_check(sort_unique(out_.as<uint64_t>(), kept, d_u, d_c, (uint64_t*)(d_cnt + 1), temp_.p, tb, bits, stream_),
                  "sort_unique");
        unsigned long long nu = 0;
        hip_check(hipMemcpyAsync(&nu, d_cnt + 1, 8, hipMemcpyDeviceToHost, stream_), "D2H");
        hip_check(hipStreamSynchronize(stream_), "sync");
        size_t take = (size_t)nu;
        if (limit && take > limit) take = limit;        // bottom-k: only the smallest `num` can ever be kept
        hashes.resize(take);
        hip_check(hipMemcpyAsync(hashes.data(), d_u, take * 8, hipMemcpyDeviceToHost, stream_), "D2H");
        if (want_counts) {
            counts.resize(take);
            hip_check(hipMemcpyAsync(counts.data(), d_c, take * 8, hipMemcpyDeviceToHost, stream_), "D2H");
        }
        hip_check(hipStreamSynchronize(stream_), "sync");
    }

    void upload_seq(const uint8_t* seq, size_t len) {
        seq_.reserve(len + 64, stream_);
        hip_check(hipMemcpyAsync(seq_.p, seq, len, hipMemcpyHostToDevice, stream_), "H2D");
    }

    void sketch_chunk(const uint8_t* seq, size_t len, uint32_t k, uint64_t seed, uint64_t thr, bool want_counts,
                      size_t limit, std::vector<uint64_t>& hashes, std::vector<uint64_t>& counts) {
        hashes.clear(); counts.clear();
        const size_t nk = len - k + 1;
        upload_seq(seq, len);
        const double frac = (double)thr / 18446744073709551616.0;
        const double expect = (double)nk * frac;
        size_t cap = (size_t)(expect * 1.5 + 8.0 * std::sqrt(expect + 1.0)) + 4096;
        if (cap > nk) cap = nk;
        unsigned long long* d_cnt = scalars_.as<unsigned long long>();
        unsigned long long kept = 0;
        for (int attempt = 0; attempt < 2; ++attempt) {
            out_.reserve(cap * 8, stream_);
            hip_check(hipMemsetAsync(d_cnt, 0, 16, stream_), "memset");
            hip_check(sketch_dna_launch(seq_.as<uint8_t>(), len, k, seed, thr, out_.as<uint64_t>(), d_cnt, cap, stream_),
                      "sketch_dna");
            hip_check(hipMemcpyAsync(&kept, d_cnt, 8, hipMemcpyDeviceToHost, stream_), "D2H");
            hip_check(hipStreamSynchronize(stream_), "sync");
            if (kept <= cap) break;
            cap = (size_t)kept;                 // repetitive input beat the estimate: rerun with the exact size
        }
        if (kept == 0) return;
        sorted_unique_to_host(kept, thr, want_counts, limit, hashes, counts);
    }

    hipStream_t stream_ = nullptr;
    std::unordered_map<uint64_t, Mirror> mirrors_;
    std::list<uint64_t> lru_;
    size_t mirror_bytes_ = 0;
    uint64_t hold_gen_ = 0;
    std::vector<void*> parked_;
    unsigned long long* slot_ = nullptr;       // pinned host memory the one-launch pair kernel publishes into
    unsigned long long slot_seq_ = 0;
    std::recursive_mutex mu_;
    DevBuf seq_, aa_, out_, uniq_, temp_, scalars_, pair_, flags_;
};

}  // namespace smg
