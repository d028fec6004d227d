// ingest.hpp -- FASTA / FASTQ (.gz) file -> HBM -> sketches, in one streaming pass.
//
// the step right before the kernel.  The reference parses records in Python
// (screed, src/sourmash/command_sketch.py:697,746-768) and crosses the FFI once per record.  Here the host does
// no parsing at all:
//   reader threads   raw file bytes -> pinned ring buffers (plain files: several threads pread different chunks
//                    at once)
//   gzip             a single-member .gz goes to HBM COMPRESSED and is inflated there (gunzip.hpp: every block of the
//                    member decoded at once, checked against the trailer's CRC-32); what the device refuses --
//                    several members, a damaged stream -- is inflated on the host's threads (pargz.hpp) as before
//   copy stream      pinned -> HBM, overlapped with everything else
//   compute stream   fastx.hip resolves the record structure on the device (headers, line breaks, FASTQ quality
//                    lines) and compacts the sequence bytes, one separator byte per record; sketch.hip hashes the
//                    compacted chunk for every sketch of the signature (all ksizes in one pass over the file)
// Consecutive chunks are joined by a k-1 byte halo kept on the device, so every k-mer is hashed exactly once.
// Kept hashes of all chunks accumulate in HBM per sketch and are sorted / uniqued (with multiplicities for
// abundance sketches) at the end, or whenever 64M entries have piled up.
#pragma once
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <memory>
#include "pargz.hpp"
#include "gunzip.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <string>
#include <thread>
#include <vector>
#include "device_ctx.hpp"
#include "fastx_api.hpp"
#include "signature_host.hpp"

namespace smg {

struct PinnedBuf {
    uint8_t* p = nullptr;
    size_t cap = 0;
    void reserve(size_t n) {
        if (n <= cap) return;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        // (a quarter more than asked for, in whole 16 MiB: batches differ by a file or two, and pinning 60 MB takes ~100 ms --
        //  a worker that grew by a megabyte per batch spent more time here than on its files)
        n = ((n + n / 4) + ((size_t)16 << 20) - 1) & ~(((size_t)16 << 20) - 1);
        hip_check(hipHostMalloc((void**)&p, n, hipHostMallocDefault), "hipHostMalloc");
        cap = n;
    }
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
};

// Produces the raw bytes of a file as an ordered sequence of chunks in a ring of pinned buffers.
class RawChunkSource {
  public:
    static constexpr int SLOTS = 8;

    RawChunkSource(const std::string& path, size_t chunk, PinnedBuf* ring, unsigned max_readers = 6)
        : path_(path), chunk_(chunk), ring_(ring) {
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) throw Error(E_IO, "No such file or directory: " + path);
        struct stat st;
        if (fstat(fd_, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd_); throw Error(E_IO, "cannot read " + path); }
        size_ = (uint64_t)st.st_size;
        uint8_t magic[2] = {0, 0};
        const ssize_t got = ::pread(fd_, magic, 2, 0);
        gz_ = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        for (int s = 0; s < SLOTS; ++s) { owner_[s] = s; filled_[s] = -1; len_[s] = 0; }
        if (gz_) {
            // one large member of text inflates on many host threads (pargz.hpp); a worker of the many-files pipeline
            // (max_readers == 1) keeps to one thread: there the files run side by side
            try {
                pg_.reset(new ParallelGunzip(path, max_readers <= 1 ? 1u : 0u));
            } catch (const std::runtime_error& e) {
                ::close(fd_);
                throw Error(E_NIFFLER, std::string("cannot initialise gzip reader for ") + path + ": " + e.what());
            }
            threads_.emplace_back([this] { produce_gz(); });
        } else {
            const uint64_t n_chunks = (size_ + chunk_ - 1) / chunk_;
            end_seq_ = (int64_t)n_chunks;
            unsigned want = max_readers;
            const unsigned nt = (unsigned)std::min<uint64_t>(want, n_chunks);
            for (unsigned t = 0; t < nt; ++t) threads_.emplace_back([this] { produce_plain(); });
        }
    }
    ~RawChunkSource() {
        {
            std::lock_guard<std::mutex> g(mu_);
            abort_ = true;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
        pg_.reset();
        if (fd_ >= 0) ::close(fd_);
    }

    // blocks until chunk `seq` is in its slot; false when the file has no such chunk
    bool wait(int64_t seq, const uint8_t** data, size_t* len) {
        std::unique_lock<std::mutex> lk(mu_);
        const int s = (int)(seq % SLOTS);
        cv_.wait(lk, [&] { return filled_[s] == seq || (end_seq_ >= 0 && seq >= end_seq_) || failed_; });
        if (failed_) throw Error(error_code_, error_);
        if (filled_[s] != seq) return false;
        *data = ring_[s].p;
        *len = len_[s];
        return true;
    }
    // the consumer no longer needs the slot of chunk `seq`
    void release(int64_t seq) {
        {
            std::lock_guard<std::mutex> g(mu_);
            owner_[seq % SLOTS] = seq + SLOTS;
        }
        cv_.notify_all();
    }
    bool gzip() const { return gz_; }

  private:
    bool claim(int64_t seq) {                       // wait until the slot's previous tenant has been released
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return owner_[seq % SLOTS] == seq || abort_; });
        return !abort_;
    }
    void publish(int64_t seq, size_t len) {
        {
            std::lock_guard<std::mutex> g(mu_);
            len_[seq % SLOTS] = len;
            filled_[seq % SLOTS] = seq;
        }
        cv_.notify_all();
    }
    void fail(uint32_t code, const std::string& what) {
        {
            std::lock_guard<std::mutex> g(mu_);
            failed_ = true; error_code_ = code; error_ = what;
        }
        cv_.notify_all();
    }
    void produce_plain() {
        for (;;) {
            const int64_t seq = next_seq_.fetch_add(1);
            if (seq >= end_seq_) return;
            if (!claim(seq)) return;
            const uint64_t off = (uint64_t)seq * chunk_;
            const size_t want = (size_t)std::min<uint64_t>(chunk_, size_ - off);
            size_t got = 0;
            while (got < want) {
                const ssize_t r = ::pread(fd_, ring_[seq % SLOTS].p + got, want - got, (off_t)(off + got));
                if (r <= 0) { fail(E_IO, "short read on " + path_); return; }
                got += (size_t)r;
            }
            publish(seq, got);
        }
    }
    void produce_gz() {
        for (int64_t seq = 0;; ++seq) {
            if (!claim(seq)) return;
            size_t got = 0;
            try {
                got = pg_->read(ring_[seq % SLOTS].p, chunk_);            // short only at the end of the stream
            } catch (const std::runtime_error& e) {
                fail(E_NIFFLER, e.what());
                return;
            }
            if (got) publish(seq, got);
            if (got < chunk_) {                     // end of stream
                {
                    std::lock_guard<std::mutex> g(mu_);
                    end_seq_ = got ? seq + 1 : seq;
                }
                cv_.notify_all();
                return;
            }
        }
    }

    std::string path_;
    size_t chunk_;
    PinnedBuf* ring_;
    int fd_ = -1;
    std::unique_ptr<ParallelGunzip> pg_;
    uint64_t size_ = 0;
    bool gz_ = false;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<std::thread> threads_;
    std::atomic<int64_t> next_seq_{0};
    int64_t end_seq_ = -1;                          // first chunk number that does not exist (-1: not known yet)
    int64_t owner_[SLOTS], filled_[SLOTS];
    size_t len_[SLOTS];
    bool abort_ = false, failed_ = false;
    uint32_t error_code_ = 0;
    std::string error_;
};

// Buffers that survive between files: pinned ring, device chunks, scratch (allocation would otherwise dominate
// small inputs such as a single bacterial genome).
struct IngestScratch {
    size_t chunk = 0;
    PinnedBuf ring[RawChunkSource::SLOTS];
    DevBuf raw[2], comp[2], state, temp, small, gzdev;
    PinnedBuf gzfile;                               // a whole .gz file on its way to the device
    hipStream_t copy_stream = nullptr;
    hipEvent_t copied[2] = {nullptr, nullptr}, consumed[2] = {nullptr, nullptr};
    size_t temp_bytes = 0;
    size_t halo = 256;                              // bytes reserved in front of every compacted chunk: >= the longest ksize, a multiple of 256

    // The device blocks go back to the arena while the stream their work ran on still exists (its owner calls this before it
    // destroys the stream: a DevBuf released later would record its release event on a destroyed stream).  The stream is
    // drained first, so the blocks carry no pending work and are handed back untied to any stream.
    void release(hipStream_t stream) {
        if (stream) (void)hipStreamSynchronize(stream);
        if (copy_stream) (void)hipStreamSynchronize(copy_stream);
        for (DevBuf* b : {&raw[0], &raw[1], &comp[0], &comp[1], &state, &temp, &small, &gzdev}) { b->st = nullptr; b->release(); }
        for (int i = 0; i < 2; ++i) {
            if (copied[i]) (void)hipEventDestroy(copied[i]);
            if (consumed[i]) (void)hipEventDestroy(consumed[i]);
            copied[i] = consumed[i] = nullptr;
        }
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        copy_stream = nullptr;
        chunk = 0;
    }
    void prepare(size_t chunk_bytes, uint32_t kmax, hipStream_t stream) {
        if (!copy_stream) {
            hip_check(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking), "hipStreamCreate");
            for (int i = 0; i < 2; ++i) {
                hip_check(hipEventCreateWithFlags(&copied[i], hipEventDisableTiming), "hipEventCreate");
                hip_check(hipEventCreateWithFlags(&consumed[i], hipEventDisableTiming), "hipEventCreate");
            }
            small.reserve(256, stream);
        }
        const size_t need_halo = ((size_t)(kmax > 256u ? kmax : 256u) + 255) / 256 * 256;
        if (chunk_bytes <= chunk && need_halo <= halo) return;
        chunk = chunk_bytes > chunk ? chunk_bytes : chunk;
        halo = need_halo > halo ? need_halo : halo;
        for (auto& r : ring) r.reserve(chunk + 64);
        for (int i = 0; i < 2; ++i) {
            raw[i].reserve(chunk + 64, stream);
            comp[i].reserve(halo + chunk + 64, stream);
        }
        state.reserve(chunk + 64, stream);
        temp_bytes = fastx_temp_bytes(chunk);
        temp.reserve(temp_bytes, stream);
    }
};

struct IngestSlot {
    void* p = nullptr;
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// One ingest pipeline: a compute stream and the buffers that go with it.  The single-file entry points share one
// (on the context's stream, under its mutex); smgpu_sketch_files gives every worker thread its own, so many files
// stream concurrently -- what a gzip-bound `sketch` over a directory of genomes needs.
struct IngestWorker {
    hipStream_t stream = nullptr;
    bool own_stream = false;
    IngestScratch scratch;
    void init_own_stream() {
        if (!stream) { hip_check(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate"); own_stream = true; }
    }
    ~IngestWorker() {
        if (own_stream && stream) {
            scratch.release(stream);
            (void)hipStreamDestroy(stream);
        }
    }
};

// A single-member gzip file inflated on the device: -> arena block with its bytes (the caller frees it with arena_free(p, st)) and
// their number, or nullptr when the device path does not apply (not gzip, too large for one pass, SMG_GUNZIP_DEVICE=0) or refused
// the file (*why says so): the caller then reads the file through the host inflater.
inline void* gunzip_file_to_device(IngestScratch& scratch, const std::string& path, hipStream_t st, uint64_t* n_out, std::string* why,
                                   GunzipStats* stats = nullptr) {
    *n_out = 0;
    static const bool off = [] { const char* e = getenv("SMG_GUNZIP_DEVICE"); return e && e[0] == '0'; }();
    if (off) { if (why) *why = "SMG_GUNZIP_DEVICE=0"; return nullptr; }
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return nullptr;                                      // (the host path reports the error)
    struct Close { int fd; ~Close() { ::close(fd); } } close_fd{fd};
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) return nullptr;
    const uint64_t size = (uint64_t)sb.st_size;
    constexpr uint64_t MAX_FILE = (uint64_t)2 << 30;                  // (resident at once: 32 B of records per input byte, symbols + bytes of the output)
    if (size < 26 || size > MAX_FILE) { if (why) *why = "file size outside the device inflater's range"; return nullptr; }
    uint8_t magic[3] = {0, 0, 0};
    if (::pread(fd, magic, 3, 0) != 3 || magic[0] != 0x1f || magic[1] != 0x8b || magic[2] != 8) return nullptr;
    scratch.gzfile.reserve((size_t)size + GUNZIP_PAD + 64);
    scratch.gzdev.reserve((size_t)size + GUNZIP_PAD + 64, st);
    uint8_t* h = scratch.gzfile.p;
    {   // page cache -> pinned memory on a few threads (one memcpy stream moves ~6 GB/s)
        const unsigned nt = (unsigned)std::min<uint64_t>(8, (size >> 23) + 1);
        std::atomic<bool> bad(false);
        auto part = [&](unsigned t) {
            const uint64_t lo = size * t / nt, hi = size * (t + 1) / nt;
            uint64_t got = lo;
            while (got < hi) {
                const ssize_t r = ::pread(fd, h + got, (size_t)(hi - got), (off_t)got);
                if (r <= 0) { bad = true; return; }
                got += (uint64_t)r;
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; ++t) th.emplace_back(part, t);
        part(0);
        for (auto& t : th) t.join();
        if (bad) return nullptr;
    }
    memset(h + size, 0, GUNZIP_PAD);
    hip_check(hipMemcpyAsync(scratch.gzdev.p, h, (size_t)size + GUNZIP_PAD, hipMemcpyHostToDevice, st), "H2D");
    std::vector<GunzipMember> ms(1);
    ms[0].file_off = 0;
    ms[0].file_len = size;
    void* d_out = nullptr;
    gunzip_device(h, scratch.gzdev.as<uint8_t>(), size, ms, &d_out, st, stats);
    if (!ms[0].ok) {
        if (why) *why = ms[0].why;
        if (d_out) arena_free(d_out, st);
        gunzip_counters().refused++;
        return nullptr;
    }
    gunzip_counters().on_device++;
    *n_out = ms[0].out_len;
    return d_out;
}

// Sketch a sequence file into every (DNA) sketch of `mhs` on the worker's pipeline.  force == true semantics.
// inflated: the file's bytes already in HBM (a member of a batch inflated by the caller: sketch_files_parallel), or
// {nullptr, 0, true} when the device has refused the file already; default: the file is tried on the device here.
struct InflatedSlice { const void* p = nullptr; uint64_t len = 0; bool refused = false; };
inline void sketch_file_with(IngestWorker& w, std::vector<KmerMinHash*>& mhs, const std::string& path, size_t CHUNK,
                             unsigned max_readers, uint64_t* n_records, uint64_t* n_bases, const InflatedSlice* inflated = nullptr) {
    for (auto* mh : mhs)
        if (!mh->is_dna()) throw err_internal("the streaming file ingest takes DNA sketches; protein / dayhoff / hp sketches are fed record by record");
    uint32_t kmax = 0;
    for (auto* mh : mhs) kmax = std::max(kmax, mh->ksize);
    if (mhs.empty() || kmax == 0) return;
    check_dna_ksize(kmax);
    hipStream_t st = w.stream;
    IngestScratch& scratch = w.scratch;
    scratch.prepare(CHUNK, kmax, st);
    const int halo = (int)kmax - 1;

    struct Acc {                       // per sketch: unordered kept hashes since the last flush
        IngestSlot out, cnt;       // stream-ordered allocations (no device-wide sync: other workers keep running)
        size_t cap = 0;
        unsigned long long count = 0;  // host copy, exact after a sync
    };
    std::vector<Acc> acc(mhs.size());
    struct FreeAcc {
        std::vector<Acc>& v;
        hipStream_t st;
        ~FreeAcc() { for (auto& a : v) { if (a.out.p) arena_free(a.out.p, st); if (a.cnt.p) arena_free(a.cnt.p, st); a.out.p = a.cnt.p = nullptr; } }
    } free_acc{acc, st};
    for (auto& a : acc) {
        hip_check(arena_alloc(&a.cnt.p, 64, st), "arena_alloc");
        hip_check(hipMemsetAsync(a.cnt.p, 0, 64, st), "memset");
    }

    // flush: sort + unique (+ multiplicities) what has accumulated, merge it into the host container
    auto flush = [&](size_t s) {
        KmerMinHash& mh = *mhs[s];
        Acc& a = acc[s];
        hip_check(hipMemcpyAsync(&a.count, a.cnt.p, 8, hipMemcpyDeviceToHost, st), "D2H");
        hip_check(hipStreamSynchronize(st), "sync");
        if (a.count > a.cap) throw err_internal("sketch output overflow while ingesting " + path);
        if (a.count == 0) return;
        const size_t tb = sort_unique_temp_bytes(a.count);
        AsyncBuf tmp(tb, st), uniq((size_t)a.count * 16 + 64, st);
        uint64_t* d_u = uniq.as<uint64_t>();
        uint64_t* d_c = d_u + a.count;
        const uint64_t thr = mh.max_hash ? mh.max_hash : ~0ull;
        int bits = 64;
        if (thr != ~0ull) { bits = 1; while (bits < 64 && (thr >> bits)) ++bits; }
        hip_check(sort_unique(a.out.as<uint64_t>(), a.count, d_u, d_c, (uint64_t*)(a.cnt.as<unsigned long long>() + 1),
                              tmp.p, tb, bits, st), "sort_unique");
        unsigned long long nu = 0;
        hip_check(hipMemcpyAsync(&nu, a.cnt.as<unsigned long long>() + 1, 8, hipMemcpyDeviceToHost, st), "D2H");
        hip_check(hipStreamSynchronize(st), "sync");
        size_t take = (size_t)nu;
        if (mh.num && take > mh.num) take = mh.num;         // bottom-k: only the smallest `num` can ever be kept
        std::vector<uint64_t> hs(take), cs;
        hip_check(hipMemcpyAsync(hs.data(), d_u, take * 8, hipMemcpyDeviceToHost, st), "D2H");
        if (mh.track_abundance) {
            cs.resize(take);
            hip_check(hipMemcpyAsync(cs.data(), d_c, take * 8, hipMemcpyDeviceToHost, st), "D2H");
        }
        hip_check(hipMemsetAsync(a.cnt.p, 0, 16, st), "memset");
        hip_check(hipStreamSynchronize(st), "sync");
        a.count = 0;
        mh.add_sorted_batch(hs.data(), mh.track_abundance ? cs.data() : nullptr, hs.size());
    };

    // device scalars: [0,4) parser carry, [8,16) compacted length of this chunk, [16,24) records
    uint8_t* d_carry = scratch.small.as<uint8_t>();
    unsigned long long* d_n = reinterpret_cast<unsigned long long*>(scratch.small.as<uint8_t>() + 8);
    unsigned long long* d_records = d_n + 1;
    hip_check(hipMemsetAsync(scratch.small.p, 0, 64, st), "memset");
    for (int b = 0; b < 2; ++b)                      // nothing precedes the first chunk: a halo of separators
        hip_check(hipMemsetAsync(scratch.comp[b].p, '\n', scratch.halo, st), "memset");
    hip_check(hipStreamSynchronize(st), "sync");     // the copy stream must not race these

    const bool trace = getenv("SMG_INGEST_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = now();
    double t_wait = 0, t_sync = 0;
    // a gzip member inflated in HBM: the chunks below are then slices of that block (no reader threads, no copies)
    uint64_t plain_len = 0;
    std::string gz_why;
    void* d_plain = nullptr;
    bool own_plain = true;
    if (inflated && inflated->p) { d_plain = const_cast<void*>(inflated->p); plain_len = inflated->len; own_plain = false; }
    else if (!(inflated && inflated->refused)) d_plain = gunzip_file_to_device(scratch, path, st, &plain_len, &gz_why);
    struct FreePlain { void*& p; bool& own; hipStream_t st; ~FreePlain() { if (p && own) arena_free(p, st); } } free_plain{d_plain, own_plain, st};
    if (!d_plain && !gz_why.empty() && trace) fprintf(stderr, "[ingest] %s: inflated on the host (%s)\n", path.c_str(), gz_why.c_str());
    std::unique_ptr<RawChunkSource> src_host;
    if (!d_plain) src_host.reset(new RawChunkSource(path, CHUNK, scratch.ring, max_readers));
    uint8_t first_byte = 0;
    if (d_plain && plain_len) {
        hip_check(hipMemcpyAsync(&first_byte, d_plain, 1, hipMemcpyDeviceToHost, st), "D2H");
        hip_check(hipStreamSynchronize(st), "sync");
    }
    constexpr size_t FLUSH_AT = (size_t)64 << 20;            // entries; keeps scratch bounded on huge inputs
    int fastq = -1;
    uint64_t total_kept = 0;
    bool used[2] = {false, false};
    // the copy of chunk seq+1 is queued before chunk seq is parsed, so the copy engine never waits for the host
    struct Staged { const uint8_t* data = nullptr; const uint8_t* dev = nullptr; size_t len = 0; bool ok = false; };
    auto stage = [&](int64_t seq) -> Staged {
        Staged c;
        if (d_plain) {
            const uint64_t off = (uint64_t)seq * CHUNK;
            if (off >= plain_len) return c;
            c.ok = true;
            c.len = (size_t)std::min<uint64_t>(CHUNK, plain_len - off);
            c.dev = static_cast<const uint8_t*>(d_plain) + off;
            return c;
        }
        RawChunkSource& src = *src_host;
        const double t0 = now();
        c.ok = src.wait(seq, &c.data, &c.len);
        t_wait += now() - t0;
        if (!c.ok) return c;
        const int b = (int)(seq & 1);
        if (used[b]) hip_check(hipStreamWaitEvent(scratch.copy_stream, scratch.consumed[b], 0), "wait");   // parse of seq-2 read raw[b]
        hip_check(hipMemcpyAsync(scratch.raw[b].p, c.data, c.len, hipMemcpyHostToDevice, scratch.copy_stream), "H2D");
        hip_check(hipEventRecord(scratch.copied[b], scratch.copy_stream), "record");
        c.dev = scratch.raw[b].as<uint8_t>();
        return c;
    };
    Staged cur = stage(0);
    for (int64_t seq = 0; cur.ok; ++seq) {
        const int b = (int)(seq & 1);
        const size_t len = cur.len;
        if (fastq < 0) {                                      // format by the first byte of the (inflated) file
            fastq = (d_plain ? first_byte : cur.data[0]) == '@' ? 1 : 0;
            const uint8_t init[4] = {(uint8_t)(fastq ? 3 : 1), 1, 0, 0};
            hip_check(hipMemcpyAsync(d_carry, init, 4, hipMemcpyHostToDevice, st), "H2D");
            hip_check(hipStreamSynchronize(st), "sync");
        }
        // compute stream: parse + compact behind the halo that the previous chunk left in comp[b]
        if (!d_plain) hip_check(hipStreamWaitEvent(st, scratch.copied[b], 0), "wait");
        uint8_t* comp = scratch.comp[b].as<uint8_t>() + scratch.halo;
        hip_check(fastx_compact_launch(cur.dev, len, fastq, d_carry, scratch.state.as<uint8_t>(), comp,
                                       d_n, d_records, scratch.temp.p, scratch.temp_bytes, st), "fastx");
        hip_check(hipEventRecord(scratch.consumed[b], st), "record");
        used[b] = true;
        const Staged next = stage(seq + 1);                   // raw[b^1] was consumed by the parse of chunk seq-1; queued
                                                              // before the (blocking) read-backs below
        unsigned long long n_kept = 0;
        hip_check(hipMemcpyAsync(&n_kept, d_n, 8, hipMemcpyDeviceToHost, st), "D2H");
        for (auto& a : acc) hip_check(hipMemcpyAsync(&a.count, a.cnt.p, 8, hipMemcpyDeviceToHost, st), "D2H");
        const double t1 = now();
        hip_check(hipStreamSynchronize(st), "sync");          // also: every launch of the previous chunk is done
        t_sync += now() - t1;
        if (src_host) src_host->release(seq);                 // the copy is complete: the pinned slot can be refilled
        total_kept += n_kept;
        struct Pending { size_t s; uint32_t k; uint64_t thr; };
        std::vector<Pending> pending;
        for (size_t s = 0; s < mhs.size(); ++s) {
            KmerMinHash& mh = *mhs[s];
            Acc& a = acc[s];
            if (mh.num == 0 && mh.max_hash == 0) continue;
            const uint32_t k = mh.ksize;
            const size_t slen = (size_t)(k - 1) + (size_t)n_kept;   // this sketch's halo is its own k-1 bytes
            if (slen < k) continue;
            const uint64_t thr = mh.max_hash ? mh.max_hash : ~0ull;
            const double frac = (double)thr / 18446744073709551616.0;
            const size_t expect = std::min(slen, (size_t)((double)slen * frac * 1.5 + 8.0 * std::sqrt((double)slen * frac + 1.0)) + 4096);
            if (a.count > a.cap) throw err_internal("sketch output overflow while ingesting " + path);
            if (a.count && (mh.num != 0 || a.count + expect > FLUSH_AT)) flush(s);
            // Capacity covers the worst case -- every k-mer of the chunk kept (the count depends on multiplicity, not on
            // distinct k-mers: deep amplicon reads, a long tandem repeat under max_hash) -- so a launch can never
            // overflow; the statistical estimate above only decides when to flush.  8 bytes per chunk byte per sketch.
            const size_t need = (size_t)a.count + std::max(expect, slen - (size_t)(k - 1));
            if (need > a.cap) {
                const size_t ncap = std::max(need + need / 2, (size_t)1 << 16);
                void* bigger = nullptr;
                hip_check(arena_alloc(&bigger, ncap * 8, st), "arena_alloc");
                if (a.count) hip_check(hipMemcpyAsync(bigger, a.out.p, (size_t)a.count * 8, hipMemcpyDeviceToDevice, st), "D2D");
                if (a.out.p) arena_free(a.out.p, st);
                a.out.p = bigger;
                a.cap = ncap;
            }
            pending.push_back(Pending{s, k, thr});
        }
        // the launches of this chunk: the signature's ksizes in one pass where the set has a fused kernel (sketch_multi.hip), else one each
        {
            bool fused = false;
            if (pending.size() == 3 && pending.size() == mhs.size() && n_kept >= 64) {
                uint32_t ks[3];
                SketchMultiOut mo[3];
                bool ok = true;
                for (size_t q = 0; q < 3; ++q) {
                    const Pending& pn = pending[q];
                    ks[q] = pn.k;
                    mo[q] = SketchMultiOut{pn.thr, acc[pn.s].out.as<uint64_t>(), acc[pn.s].cnt.as<unsigned long long>(), acc[pn.s].cap};
                    ok = ok && mhs[pn.s]->seed == mhs[pending[0].s]->seed;
                }
                if (ok && sketch_dna_multi_supported(ks, 3)) {
                    hip_check(sketch_dna_multi_launch(comp, (uint64_t)n_kept, ks, 3, mhs[pending[0].s]->seed, mo, st), "sketch_dna_multi");
                    fused = true;
                }
            }
            for (size_t q = 0; !fused && q < pending.size(); ++q) {
                const Pending& pn = pending[q];
                Acc& a = acc[pn.s];
                hip_check(sketch_dna_launch(comp - (pn.k - 1), (size_t)(pn.k - 1) + (size_t)n_kept, pn.k, mhs[pn.s]->seed, pn.thr, a.out.as<uint64_t>(),
                                            a.cnt.as<unsigned long long>(), a.cap, st), "sketch_dna");
            }
        }
        // the next chunk's halo: the last kmax-1 bytes of the stream so far
        hip_check(fastx_halo_launch(comp, d_n, halo, scratch.comp[b ^ 1].as<uint8_t>() + scratch.halo - halo, st), "halo");
        cur = next;
    }
    unsigned long long recs = 0;
    hip_check(hipMemcpyAsync(&recs, d_records, 8, hipMemcpyDeviceToHost, st), "D2H");
    hip_check(hipStreamSynchronize(st), "sync");
    hip_check(hipStreamSynchronize(scratch.copy_stream), "sync");
    const double t_loop = now();
    if (n_records) *n_records = recs;
    if (n_bases) *n_bases = total_kept - recs;               // one separator byte per record is in the stream
    for (size_t s = 0; s < mhs.size(); ++s)
        if (acc[s].cnt.p && !(mhs[s]->num == 0 && mhs[s]->max_hash == 0)) flush(s);
    if (trace)
        fprintf(stderr, "[ingest] %s: loop %.3f s (waiting for the reader %.3f s, for the GPU %.3f s), final flush %.3f s\n",
                path.c_str(), t_loop - t_start, t_wait, t_sync, now() - t_loop);
}

// Many small files whose bytes are in HBM already (a batch of gzip members inflated together) sketched TOGETHER: the per-file
// pipeline above reads back counts after every chunk and sorts per sketch -- a dozen stream synchronisations per file, which is
// all a single genome costs.  Here every file's parse + k-mer launches are queued without a read-back (each file's compacted
// bytes land in its own region of one block pre-filled with separators, so the k-mer kernel can be given the raw length as an
// upper bound), the counts of all files come back in ONE copy, the per-sketch sorts are queued, and the sketches come back
// in one more.  done[i] = true for the files served; the others (too large, bottom-k sketches, an output estimate exceeded)
// take the per-file path.
inline void sketch_slices_batched(IngestWorker& w, const std::vector<InflatedSlice>& slices, const std::vector<uint8_t>& first_bytes,
                                  std::vector<Signature>& sigs, std::vector<uint64_t>& bases_out, std::vector<bool>& done) {
    const size_t n = slices.size();
    done.assign(n, false);
    bases_out.assign(n, 0);
    constexpr uint64_t MAX_LEN = (uint64_t)32 << 20;
    hipStream_t st = w.stream;
    std::vector<size_t> idx;
    uint64_t max_len = 0;
    uint32_t kmax = 0;
    size_t n_sk = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!slices[i].p || slices[i].len == 0 || slices[i].len > MAX_LEN) continue;
        bool ok = !sigs[i].sketches.empty();
        for (auto& mh : sigs[i].sketches) ok = ok && mh.is_dna() && mh.num == 0 && mh.max_hash != 0 && mh.ksize >= 1 && mh.ksize <= 256;
        if (!ok) continue;
        idx.push_back(i);
        max_len = std::max(max_len, slices[i].len);
        for (auto& mh : sigs[i].sketches) kmax = std::max(kmax, mh.ksize);
        n_sk = std::max(n_sk, sigs[i].sketches.size());
    }
    if (idx.empty()) return;
    // (the parser's scratch for the longest file of the batch; the worker's own chunk buffers stay at their 4 MiB)
    const size_t parse_temp_bytes = fastx_temp_bytes(max_len);
    AsyncBuf parse_temp(parse_temp_bytes, st);
    (void)kmax;
    constexpr size_t HALO = 256;
    // layout: per file a region of the compacted-bytes block, 32 bytes of scalars, and per sketch an output region + 16 bytes of counts
    struct Slot { uint64_t comp_off, out_off[8], cap[8]; };
    if (n_sk > 8) return;
    std::vector<Slot> slot(idx.size());
    uint64_t comp_bytes = 0, out_entries = 0;
    for (size_t j = 0; j < idx.size(); ++j) {
        const size_t i = idx[j];
        slot[j].comp_off = comp_bytes + HALO;
        comp_bytes += HALO + ((slices[i].len + 64 + 255) & ~255ull);
        for (size_t q = 0; q < sigs[i].sketches.size(); ++q) {
            const KmerMinHash& mh = sigs[i].sketches[q];
            const double frac = (double)mh.max_hash / 18446744073709551616.0;
            const double len = (double)slices[i].len;
            const uint64_t cap = std::min<uint64_t>(slices[i].len, (uint64_t)(len * frac * 2.0 + 16.0 * std::sqrt(len * frac + 1.0)) + 4096);
            slot[j].out_off[q] = out_entries;
            slot[j].cap[q] = cap;
            out_entries += cap;
        }
    }
    const size_t per_file_scalars = 32 + 16 * 8;                       // carry (4) | n_kept (8 @ 8) | records (8 @ 16); then 8 x {count, unique}
    AsyncBuf comp(comp_bytes + 256, st), outs(out_entries * 8 + 256, st), small(idx.size() * per_file_scalars + 64, st);
    hip_check(hipMemsetAsync(comp.p, '\n', comp_bytes + 256, st), "memset");
    std::vector<uint8_t> h_small(idx.size() * per_file_scalars, 0);
    std::vector<int> fastq(idx.size());
    for (size_t j = 0; j < idx.size(); ++j) {
        fastq[j] = first_bytes[idx[j]] == '@' ? 1 : 0;
        uint8_t* c = h_small.data() + j * per_file_scalars;
        c[0] = (uint8_t)(fastq[j] ? 3 : 1); c[1] = 1;
    }
    hip_check(hipMemcpyAsync(small.p, h_small.data(), h_small.size(), hipMemcpyHostToDevice, st), "H2D");
    for (size_t j = 0; j < idx.size(); ++j) {
        const size_t i = idx[j];
        uint8_t* sc = small.as<uint8_t>() + j * per_file_scalars;
        uint8_t* cp = comp.as<uint8_t>() + slot[j].comp_off;
        hip_check(fastx_compact_launch(static_cast<const uint8_t*>(slices[i].p), slices[i].len, fastq[j], sc, nullptr, cp,
                                       reinterpret_cast<unsigned long long*>(sc + 8), reinterpret_cast<unsigned long long*>(sc + 16),
                                       parse_temp.p, parse_temp_bytes, st, true), "fastx");
        // the signature's ksizes in one pass where the set has a fused kernel (sketch_multi.hip: 21 / 31 / 51), else a launch each
        bool fused = false;
        {
            const auto& sk = sigs[i].sketches;
            uint32_t ks[3];
            SketchMultiOut mo[3];
            bool same_seed = sk.size() == 3;
            for (size_t q = 0; same_seed && q < 3; ++q) {
                ks[q] = sk[q].ksize;
                mo[q] = SketchMultiOut{sk[q].max_hash, outs.as<uint64_t>() + slot[j].out_off[q], reinterpret_cast<unsigned long long*>(sc + 32 + 16 * q), slot[j].cap[q]};
                same_seed = sk[q].seed == sk[0].seed;
            }
            if (same_seed && sketch_dna_multi_supported(ks, 3)) {
                hip_check(sketch_dna_multi_launch(cp, slices[i].len, ks, 3, sk[0].seed, mo, st), "sketch_dna_multi");
                fused = true;
            }
        }
        for (size_t q = 0; !fused && q < sigs[i].sketches.size(); ++q) {
            const KmerMinHash& mh = sigs[i].sketches[q];
            const uint32_t k = mh.ksize;
            hip_check(sketch_dna_launch(cp - (k - 1), (uint64_t)(k - 1) + slices[i].len, k, mh.seed, mh.max_hash, outs.as<uint64_t>() + slot[j].out_off[q],
                                        reinterpret_cast<unsigned long long*>(sc + 32 + 16 * q), slot[j].cap[q], st), "sketch_dna");
        }
    }
    hip_check(hipMemcpyAsync(h_small.data(), small.p, h_small.size(), hipMemcpyDeviceToHost, st), "D2H");
    hip_check(hipStreamSynchronize(st), "sync");
    // sorts: only where the estimate held
    std::vector<bool> fits(idx.size(), true);
    uint64_t uniq_entries = 0, max_count = 0;
    std::vector<uint64_t> uoff(idx.size() * 8, 0);
    auto scalar = [&](size_t j, size_t off) { uint64_t v; memcpy(&v, h_small.data() + j * per_file_scalars + off, 8); return v; };
    for (size_t j = 0; j < idx.size(); ++j) {
        const size_t i = idx[j];
        for (size_t q = 0; q < sigs[i].sketches.size(); ++q) if (scalar(j, 32 + 16 * q) > slot[j].cap[q]) fits[j] = false;
        if (!fits[j]) continue;
        for (size_t q = 0; q < sigs[i].sketches.size(); ++q) {
            const uint64_t c = scalar(j, 32 + 16 * q);
            uoff[j * 8 + q] = uniq_entries;
            uniq_entries += 2 * c;
            max_count = std::max(max_count, c);
        }
    }
    // every list that fits, sorted in ONE radix sort: the list's number rides in the key bits above the hashes
    int hbits = 1;
    {
        uint64_t thr = 0;
        for (size_t j = 0; j < idx.size(); ++j) for (auto& mh : sigs[idx[j]].sketches) thr = std::max(thr, mh.max_hash);
        while (hbits < 64 && (thr >> hbits)) ++hbits;
    }
    std::vector<TagSegment> segs;
    std::vector<std::pair<size_t, size_t>> seg_of;                     // (j, q) of every segment
    uint64_t tagged = 0;
    for (size_t j = 0; j < idx.size(); ++j) {
        if (!fits[j]) continue;
        for (size_t q = 0; q < sigs[idx[j]].sketches.size(); ++q) {
            const uint64_t c = scalar(j, 32 + 16 * q);
            segs.push_back(TagSegment{slot[j].out_off[q], tagged, c});
            seg_of.emplace_back(j, q);
            tagged += c;
        }
    }
    const bool one_sort = hbits < 64 && segs.size() <= ((uint64_t)1 << (64 - hbits)) && tagged < 0xffffffffull;
    if (tagged && one_sort) {
        const size_t tb = sort_unique_temp_bytes(tagged);
        AsyncBuf tmp(tb, st), keys(tagged * 8 + 256, st), uniq(tagged * 16 + 256, st), d_segs(segs.size() * sizeof(TagSegment), st);
        hip_check(hipMemcpyAsync(d_segs.p, segs.data(), segs.size() * sizeof(TagSegment), hipMemcpyHostToDevice, st), "H2D");
        hip_check(tag_gather_launch(outs.as<uint64_t>(), d_segs.as<TagSegment>(), (uint32_t)segs.size(), keys.as<uint64_t>(), hbits, st), "tag_gather");
        uint64_t* d_u = uniq.as<uint64_t>();
        unsigned long long* d_nu = reinterpret_cast<unsigned long long*>(small.as<uint8_t>());       // (the first file's carry bytes have done their work)
        hip_check(sort_unique(keys.as<uint64_t>(), tagged, d_u, d_u + tagged, reinterpret_cast<uint64_t*>(d_nu), tmp.p, tb, 64, st), "sort_unique");
        unsigned long long nu = 0;
        hip_check(hipMemcpyAsync(&nu, d_nu, 8, hipMemcpyDeviceToHost, st), "D2H");
        hip_check(hipStreamSynchronize(st), "sync");
        std::vector<uint64_t> h_u((size_t)nu), h_c((size_t)nu);
        if (nu) {
            hip_check(hipMemcpyAsync(h_u.data(), d_u, (size_t)nu * 8, hipMemcpyDeviceToHost, st), "D2H");
            hip_check(hipMemcpyAsync(h_c.data(), d_u + tagged, (size_t)nu * 8, hipMemcpyDeviceToHost, st), "D2H");
            hip_check(hipStreamSynchronize(st), "sync");
        }
        const uint64_t mask = ((uint64_t)1 << hbits) - 1;
        size_t at = 0;
        while (at < (size_t)nu) {                                      // runs of equal list number, in list order
            const uint64_t tag = h_u[at] >> hbits;
            size_t end = at;
            while (end < (size_t)nu && (h_u[end] >> hbits) == tag) { h_u[end] &= mask; ++end; }
            KmerMinHash& mh = sigs[idx[seg_of[tag].first]].sketches[seg_of[tag].second];
            mh.add_sorted_batch(h_u.data() + at, mh.track_abundance ? h_c.data() + at : nullptr, end - at);
            at = end;
        }
    } else if (uniq_entries) {
        const size_t tb = sort_unique_temp_bytes(max_count);
        AsyncBuf tmp(tb, st), uniq(uniq_entries * 8 + 256, st);
        for (size_t j = 0; j < idx.size(); ++j) {
            if (!fits[j]) continue;
            const size_t i = idx[j];
            uint8_t* sc = small.as<uint8_t>() + j * per_file_scalars;
            for (size_t q = 0; q < sigs[i].sketches.size(); ++q) {
                const uint64_t c = scalar(j, 32 + 16 * q);
                if (!c) continue;
                const uint64_t thr = sigs[i].sketches[q].max_hash;
                int bits = 1;
                while (bits < 64 && (thr >> bits)) ++bits;
                uint64_t* d_u = uniq.as<uint64_t>() + uoff[j * 8 + q];
                hip_check(sort_unique(outs.as<uint64_t>() + slot[j].out_off[q], c, d_u, d_u + c, reinterpret_cast<uint64_t*>(sc + 32 + 16 * q + 8),
                                      tmp.p, tb, bits, st), "sort_unique");
            }
        }
        std::vector<uint64_t> h_uniq(uniq_entries);
        hip_check(hipMemcpyAsync(h_small.data(), small.p, h_small.size(), hipMemcpyDeviceToHost, st), "D2H");
        hip_check(hipMemcpyAsync(h_uniq.data(), uniq.p, uniq_entries * 8, hipMemcpyDeviceToHost, st), "D2H");
        hip_check(hipStreamSynchronize(st), "sync");
        for (size_t j = 0; j < idx.size(); ++j) {
            if (!fits[j]) continue;
            const size_t i = idx[j];
            for (size_t q = 0; q < sigs[i].sketches.size(); ++q) {
                const uint64_t c = scalar(j, 32 + 16 * q);
                if (!c) continue;
                const uint64_t nu = scalar(j, 32 + 16 * q + 8);
                KmerMinHash& mh = sigs[i].sketches[q];
                const uint64_t* hu = h_uniq.data() + uoff[j * 8 + q];
                mh.add_sorted_batch(hu, mh.track_abundance ? hu + c : nullptr, (size_t)nu);
            }
        }
    }
    for (size_t j = 0; j < idx.size(); ++j) {
        if (!fits[j]) continue;
        done[idx[j]] = true;
        bases_out[idx[j]] = scalar(j, 8) - scalar(j, 16);             // one separator byte per record is in the compacted stream
    }
}

// single-file entry point: the shared pipeline on the context's stream
inline void sketch_file_into(std::vector<KmerMinHash*>& mhs, const std::string& path, uint64_t* n_records,
                             uint64_t* n_bases) {
    DeviceCtx& ctx = DeviceCtx::get();
    std::lock_guard<std::recursive_mutex> g(ctx.mutex());
    static IngestWorker& shared = *new IngestWorker();   // guarded by the context mutex; leaked on purpose like the
                                                         // context (its pinned buffers must not be freed after HIP is gone)
    shared.stream = ctx.stream();
    // 32 MiB chunks; SMG_INGEST_CHUNK (bytes) overrides it so tests can force many chunk boundaries
    size_t chunk = (size_t)32 << 20;
    if (const char* e = getenv("SMG_INGEST_CHUNK")) { const long v = atol(e); if (v >= 256) chunk = (size_t)v; }
    sketch_file_with(shared, mhs, path, chunk, 6, n_records, n_bases);
}

// Many files at once: `threads` workers, each with its own stream, pinned ring and device chunks (4 MiB pieces: the
// inputs are typically single genomes), pull files from a shared counter.  out[i] is the signature of paths[i].
inline void sketch_files_parallel(const std::vector<std::string>& paths, const ComputeParameters& params, unsigned threads,
                                  std::vector<Signature>& out, uint64_t* total_bases) {
    (void)DeviceCtx::get();                         // fail early without a GPU
    out.assign(paths.size(), Signature());
    if (threads == 0) threads = std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
    if (threads > paths.size()) threads = (unsigned)std::max<size_t>(paths.size(), 1);
    std::atomic<size_t> next(0);
    std::atomic<uint64_t> bases(0);
    std::mutex err_mutex;
    std::vector<Error> errors;
    int device = 0;
    (void)hipGetDevice(&device);
    // files are claimed a batch at a time: the gzip members of a batch (single genomes: a few dozen deflate blocks each) are
    // inflated on the device in ONE pass (gunzip.hpp) -- a member by itself would keep a few dozen wavefronts busy
    // (a batch costs ~20 ms of dependent steps however few files it holds and ~0.3 ms of launches per file: up to 64 files a
    //  batch.  Round 6, 256 E. coli genomes: 4 workers 0.075 s, 16 workers 0.13 s -- the batches' kernels (inflate pass 1 fills every
    //  CU by itself) only queue behind each other and the workers contend for the runtime's locks: at most SMG_INGEST_WORKERS (4)
    //  pipelines run side by side whatever `threads` says, and the files are dealt in two rounds of batches per worker)
    static const unsigned max_workers = [] { const char* e = getenv("SMG_INGEST_WORKERS"); const long v = e ? atol(e) : 0; return (unsigned)(v >= 1 && v <= 64 ? v : 4); }();
    if (threads > max_workers) threads = max_workers;
    const size_t n_batches = (size_t)threads * 2;
    const size_t per_batch = threads <= 1 ? std::min<size_t>(64, paths.size())
                                          : std::max<size_t>(1, std::min<size_t>(64, (paths.size() + n_batches - 1) / n_batches));
    constexpr uint64_t BATCH_FILE_MAX = (uint64_t)64 << 20;          // larger files go by themselves (sketch_file_with inflates them)
    static const bool device_gunzip = [] { const char* e = getenv("SMG_GUNZIP_DEVICE"); return !(e && e[0] == '0'); }();
    // Workers outlive the call (pinning 100 MB of host memory and creating a stream cost tens of milliseconds, and the runtime
    // serialises them against every other thread's work): a pool, leaked on purpose like the context.
    struct WorkerPool {
        std::mutex mu;
        std::vector<IngestWorker*> idle;
        IngestWorker* take() {
            { std::lock_guard<std::mutex> g(mu); if (!idle.empty()) { IngestWorker* w = idle.back(); idle.pop_back(); return w; } }
            IngestWorker* w = new IngestWorker();
            w->init_own_stream();
            return w;
        }
        void give(IngestWorker* w) { std::lock_guard<std::mutex> g(mu); idle.push_back(w); }
    };
    static WorkerPool& workers = *new WorkerPool();
    auto run = [&]() {
        (void)hipSetDevice(device);
        IngestWorker* wp = nullptr;
        struct Return { WorkerPool& pool; IngestWorker*& w; ~Return() { if (w) { (void)hipStreamSynchronize(w->stream); pool.give(w); } } } give_back{workers, wp};
        try {
            wp = workers.take();
            IngestWorker& w = *wp;
            for (;;) {
                const size_t i0 = next.fetch_add(per_batch);
                if (i0 >= paths.size()) break;
                const size_t i1 = std::min(paths.size(), i0 + per_batch);
                static const bool trace = getenv("SMG_INGEST_TRACE") != nullptr;
                auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
                const double t_begin = now();
                double t_read = 0, t_inflate = 0, t_sketch = 0;
                GunzipStats gstats;
                // ---- the batch's gzip files -> pinned memory -> HBM -> inflated ----
                std::vector<GunzipMember> ms;
                std::vector<size_t> owner;                            // ms[k] is paths[owner[k]]
                std::vector<InflatedSlice> slice(i1 - i0);
                std::vector<uint8_t> first(i1 - i0, 0);
                void* d_out = nullptr;
                struct FreeOut { void*& p; hipStream_t st; ~FreeOut() { if (p) arena_free(p, st); } } free_out{d_out, w.stream};
                if (device_gunzip && i1 - i0 > 1) {
                    std::vector<int> fds;
                    uint64_t total = 0;
                    for (size_t i = i0; i < i1; ++i) {
                        const int fd = ::open(paths[i].c_str(), O_RDONLY);
                        struct stat sb;
                        uint8_t magic[3] = {0, 0, 0};
                        if (fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && (uint64_t)sb.st_size >= 26 && (uint64_t)sb.st_size <= BATCH_FILE_MAX &&
                            ::pread(fd, magic, 3, 0) == 3 && magic[0] == 0x1f && magic[1] == 0x8b && magic[2] == 8) {
                            GunzipMember m;
                            m.file_off = total;
                            m.file_len = (uint64_t)sb.st_size;
                            total += (m.file_len + 7) & ~7ull;
                            ms.push_back(m);
                            owner.push_back(i);
                            fds.push_back(fd);
                        } else if (fd >= 0) ::close(fd);             // (not a gzip file, or one for the single-file path: it opens the file itself)
                    }
                    if (!ms.empty()) {
                        w.scratch.gzfile.reserve((size_t)total + GUNZIP_PAD + 64);
                        w.scratch.gzdev.reserve((size_t)total + GUNZIP_PAD + 64, w.stream);
                        uint8_t* h = w.scratch.gzfile.p;
                        bool read_ok = true;
                        for (size_t k = 0; k < ms.size(); ++k) {
                            uint64_t got = 0;
                            while (got < ms[k].file_len) {
                                const ssize_t r = ::pread(fds[k], h + ms[k].file_off + got, (size_t)(ms[k].file_len - got), (off_t)got);
                                if (r <= 0) { read_ok = false; break; }
                                got += (uint64_t)r;
                            }
                            ::close(fds[k]);
                            const uint64_t end = ms[k].file_off + ms[k].file_len;
                            memset(h + end, 0, (size_t)((((end + 7) & ~7ull)) - end));
                        }
                        t_read = now() - t_begin;
                        if (read_ok) {
                            memset(h + total, 0, GUNZIP_PAD);
                            hip_check(hipMemcpyAsync(w.scratch.gzdev.p, h, (size_t)total + GUNZIP_PAD, hipMemcpyHostToDevice, w.stream), "H2D");
                            gunzip_device(h, w.scratch.gzdev.as<uint8_t>(), total, ms, &d_out, w.stream, trace ? &gstats : nullptr);
                            t_inflate = now() - t_begin - t_read;
                            for (size_t k = 0; k < ms.size(); ++k) {
                                InflatedSlice& sl = slice[owner[k] - i0];
                                if (ms[k].ok) {
                                    sl.p = (const uint8_t*)d_out + ms[k].out_off; sl.len = ms[k].out_len; first[owner[k] - i0] = ms[k].first_byte;
                                    gunzip_counters().on_device++;
                                }
                                else { sl.refused = true; gunzip_counters().refused++; }
                            }
                        }
                    }
                }
                std::vector<Signature> sigs;
                for (size_t i = i0; i < i1; ++i) sigs.push_back(Signature::from_params(params));
                std::vector<bool> done(i1 - i0, false);
                std::vector<uint64_t> file_bases(i1 - i0, 0);
                const double t_s0 = now();
                if (d_out) sketch_slices_batched(w, slice, first, sigs, file_bases, done);
                t_sketch = now() - t_s0;
                for (size_t i = i0; i < i1; ++i) {
                    Signature& sig = sigs[i - i0];
                    uint64_t recs = 0, b = file_bases[i - i0];
                    if (!done[i - i0]) {
                        std::vector<KmerMinHash*> mhs;
                        for (auto& mh : sig.sketches) mhs.push_back(&mh);
                        const InflatedSlice& sl = slice[i - i0];
                        sketch_file_with(w, mhs, paths[i], (size_t)4 << 20, 1, &recs, &b, (sl.p || sl.refused) ? &sl : nullptr);
                    }
                    sig.filename = paths[i];
                    bases += b;
                    out[i] = std::move(sig);
                }
                if (trace)
                    fprintf(stderr, "[ingest] batch of %zu files: %.1f ms (read %.1f, H2D + inflate %.1f [scan %.1f pass1 %.1f link %.1f pass2 %.1f finish %.1f], "
                                    "batched sketch %.1f, the rest %.1f)\n", i1 - i0, now() - t_begin, t_read, t_inflate, gstats.scan_ms, gstats.pass1_ms,
                            gstats.link_ms, gstats.pass2_ms, gstats.finish_ms, t_sketch, now() - t_begin - t_read - t_inflate - t_sketch);
            }
            hip_check(hipStreamSynchronize(w.stream), "sync");
        } catch (const Error& e) {
            std::lock_guard<std::mutex> g(err_mutex);
            errors.push_back(e);
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads; ++t) pool.emplace_back(run);
    run();
    for (auto& t : pool) t.join();
    if (!errors.empty()) throw errors.front();
    if (total_bases) *total_bases = bases.load();
}

}  // namespace smg
