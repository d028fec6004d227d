// ingest.hpp -- FASTA / FASTQ (.gz) file -> device -> sketches, in one streaming pass.
//
// the step right before the kernel.  The reference parses records in
// Python (screed, src/sourmash/command_sketch.py:697,746-768) and crosses the FFI once per record;
// here a native reader strips headers and line breaks, writes the records back to back with ONE
// separator byte between them (a byte outside ACGT kills exactly the k-mers that would span two
// records), and ships 64 MiB chunks through two pinned staging buffers so that parsing chunk i+1
// overlaps the H2D copy and the kernels of chunk i.  Consecutive chunks overlap by kmax-1 bytes, so
// every k-mer is hashed exactly once.  Kept hashes of all chunks accumulate in HBM per sketch and
// are sorted / uniqued (with multiplicities for abundance sketches) once at the end.
#pragma once
#include <zlib.h>
#include <algorithm>
#include <string>
#include <vector>
#include "device_ctx.hpp"
#include "signature_host.hpp"

namespace smg {

// Streaming FASTA/FASTQ reader (gz or plain: zlib's gzread handles both).  fill() appends sequence
// bytes and '\n' record separators to dst until it is full or the file ends.
class SeqFileReader {
  public:
    explicit SeqFileReader(const std::string& path) : buf_(1 << 20) {
        f_ = gzopen(path.c_str(), "rb");
        if (!f_) throw Error(E_IO, "No such file or directory: " + path);
        gzbuffer(f_, 1 << 20);
    }
    ~SeqFileReader() { if (f_) gzclose(f_); }

    uint64_t n_records = 0, n_bases = 0;

    // returns bytes written; 0 at end of file
    size_t fill(uint8_t* dst, size_t cap) {
        size_t w = 0;
        while (w < cap) {
            if (pos_ == len_) {
                if (eof_) break;
                const int got = gzread(f_, buf_.data(), (unsigned)buf_.size());
                if (got < 0) throw Error(E_NIFFLER, "error while reading sequence file");
                if (got == 0) { eof_ = true; break; }
                pos_ = 0; len_ = (size_t)got;
            }
            const uint8_t* p = buf_.data() + pos_;
            const size_t avail = len_ - pos_;
            if (at_line_start_) {
                at_line_start_ = false;
                const uint8_t c = *p;
                if (format_ == UNKNOWN) format_ = (c == '@') ? FASTQ : FASTA;
                bool header;
                if (format_ == FASTA) header = (c == '>');
                else header = (line_in_record_ == 0);
                if (header) {
                    if (emitted_since_sep_) { dst[w++] = '\n'; emitted_since_sep_ = false; }
                    ++n_records;
                    seq_line_ = false;
                } else {
                    seq_line_ = (format_ == FASTA) || (line_in_record_ == 1);
                }
                if (w == cap) break;     // the separator filled dst; the line itself is handled next call
            }
            const uint8_t* nl = (const uint8_t*)memchr(p, '\n', avail);
            const size_t span = nl ? (size_t)(nl - p) : avail;
            size_t take = span;
            if (seq_line_) {
                take = std::min(span, cap - w);
                size_t copy = take;
                if (copy && p[copy - 1] == '\r') --copy;      // CRLF files
                memcpy(dst + w, p, copy);
                w += copy; n_bases += copy;
                if (copy) emitted_since_sep_ = true;
            }
            pos_ += take;
            if (take < span) break;                          // dst full mid-line; resume here next call
            if (!nl) continue;                               // the line continues in the next read block
            ++pos_;                                          // the newline itself
            at_line_start_ = true;
            if (format_ == FASTQ) line_in_record_ = (line_in_record_ + 1) & 3;
        }
        return w;
    }

  private:
    enum Format { UNKNOWN, FASTA, FASTQ } format_ = UNKNOWN;
    gzFile f_ = nullptr;
    std::vector<uint8_t> buf_;
    size_t pos_ = 0, len_ = 0;
    bool eof_ = false, at_line_start_ = true, seq_line_ = false, emitted_since_sep_ = false;
    int line_in_record_ = 0;
};

struct PinnedBuf {
    uint8_t* p = nullptr;
    size_t cap = 0;
    void reserve(size_t n) {
        if (n <= cap) return;
        if (p) (void)hipHostFree(p);
        hip_check(hipHostMalloc((void**)&p, n, hipHostMallocDefault), "hipHostMalloc");
        cap = n;
    }
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
};

// Sketch a sequence file into every (DNA) sketch of `sigs`.  force == true semantics.
inline void sketch_file_into(std::vector<KmerMinHash*>& mhs, const std::string& path, uint64_t* n_records,
                             uint64_t* n_bases) {
    for (auto* mh : mhs)
        if (!mh->is_dna()) throw err_internal("sourmash_amd accelerates DNA sketches only");
    uint32_t kmax = 0;
    for (auto* mh : mhs) kmax = std::max(kmax, mh->ksize);
    if (mhs.empty() || kmax == 0) return;
    DeviceCtx& ctx = DeviceCtx::get();
    std::lock_guard<std::mutex> g(ctx.mutex());
    hipStream_t st = ctx.stream();

    // 64 MiB chunks; SMG_INGEST_CHUNK (bytes) overrides it so tests can force many chunk boundaries
    size_t CHUNK = (size_t)64 << 20;
    if (const char* e = getenv("SMG_INGEST_CHUNK")) { const long v = atol(e); if (v >= 256) CHUNK = (size_t)v; }
    const size_t halo = kmax - 1;
    PinnedBuf pin[2];
    DevBuf dseq[2];
    hipEvent_t done[2];
    for (int i = 0; i < 2; ++i) {
        pin[i].reserve(CHUNK + halo + 64);
        dseq[i].reserve(CHUNK + halo + 64);
        hip_check(hipEventCreateWithFlags(&done[i], hipEventDisableTiming), "hipEventCreate");
    }
    struct Acc {                       // per sketch: unordered kept hashes since the last flush
        DevBuf out, cnt;
        size_t cap = 0;
        unsigned long long count = 0;  // host copy, exact after a sync
    };
    std::vector<Acc> acc(mhs.size());
    for (auto& a : acc) { a.cnt.reserve(64); hip_check(hipMemsetAsync(a.cnt.p, 0, 64, st), "memset"); }

    // flush: sort + unique (+ multiplicities) what has accumulated, merge it into the host container
    auto flush = [&](size_t s) {
        KmerMinHash& mh = *mhs[s];
        Acc& a = acc[s];
        hip_check(hipMemcpyAsync(&a.count, a.cnt.p, 8, hipMemcpyDeviceToHost, st), "D2H");
        hip_check(hipStreamSynchronize(st), "sync");
        if (a.count > a.cap) throw err_internal("sketch output overflow while ingesting " + path);
        if (a.count == 0) return;
        DevBuf uniq, tmp;
        struct Free { DevBuf& b; ~Free() { if (b.p) (void)hipFree(b.p); } } f1{uniq}, f2{tmp};
        const size_t tb = sort_unique_temp_bytes(a.count);
        tmp.reserve(tb);
        uniq.reserve((size_t)a.count * 16 + 64);
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

    SeqFileReader rd(path);
    std::vector<uint8_t> carry;        // last kmax-1 bytes of the previous chunk
    bool pending[2] = {false, false};
    constexpr size_t FLUSH_AT = (size_t)64 << 20;            // entries; keeps scratch bounded on huge inputs
    for (int b = 0;; b ^= 1) {
        if (pending[b]) { hip_check(hipEventSynchronize(done[b]), "event sync"); pending[b] = false; }
        memcpy(pin[b].p, carry.data(), carry.size());
        const size_t got = rd.fill(pin[b].p + carry.size(), CHUNK);
        if (got == 0) break;
        const size_t len = carry.size() + got;
        hip_check(hipMemcpyAsync(dseq[b].p, pin[b].p, len, hipMemcpyHostToDevice, st), "H2D");
        for (size_t s = 0; s < mhs.size(); ++s) {
            KmerMinHash& mh = *mhs[s];
            Acc& a = acc[s];
            if (mh.num == 0 && mh.max_hash == 0) continue;
            const uint32_t k = mh.ksize;
            // this sketch only needs k-1 bytes of overlap: skip the rest of the carried prefix so that no
            // k-mer is hashed in two chunks (matters for abundances)
            const size_t skip = carry.size() > (size_t)(k - 1) ? carry.size() - (k - 1) : 0;
            const size_t slen = len - skip;
            const uint64_t thr = mh.max_hash ? mh.max_hash : ~0ull;
            const double frac = (double)thr / 18446744073709551616.0;
            const size_t expect = std::min(slen, (size_t)((double)slen * frac * 1.5 + 8.0 * std::sqrt((double)slen * frac + 1.0)) + 4096);
            hip_check(hipMemcpyAsync(&a.count, a.cnt.p, 8, hipMemcpyDeviceToHost, st), "D2H");
            hip_check(hipStreamSynchronize(st), "sync");
            if (a.count > a.cap) throw err_internal("sketch output overflow while ingesting " + path);
            if (a.count && (mh.num != 0 || a.count + expect > FLUSH_AT)) flush(s);
            const size_t need = (size_t)a.count + expect;
            if (need > a.cap) {
                const size_t ncap = std::max(need + need / 2, (size_t)1 << 16);
                DevBuf bigger;
                bigger.reserve(ncap * 8);
                if (a.count) hip_check(hipMemcpyAsync(bigger.p, a.out.p, (size_t)a.count * 8, hipMemcpyDeviceToDevice, st), "D2D");
                hip_check(hipStreamSynchronize(st), "sync");
                if (a.out.p) (void)hipFree(a.out.p);
                a.out = bigger; bigger.p = nullptr; bigger.cap = 0;
                a.cap = ncap;
            }
            hip_check(sketch_dna_launch(dseq[b].as<uint8_t>() + skip, slen, k, mh.seed, thr, a.out.as<uint64_t>(),
                                        a.cnt.as<unsigned long long>(), a.cap, st), "sketch_dna");
        }
        hip_check(hipEventRecord(done[b], st), "event record");
        pending[b] = true;
        const size_t keep = std::min(halo, len);             // the next chunk starts with these bytes
        carry.assign(pin[b].p + len - keep, pin[b].p + len);
    }
    for (int i = 0; i < 2; ++i)
        if (pending[i]) hip_check(hipEventSynchronize(done[i]), "event sync");
    if (n_records) *n_records = rd.n_records;
    if (n_bases) *n_bases = rd.n_bases;
    for (size_t s = 0; s < mhs.size(); ++s)
        if (acc[s].cnt.p && !(mhs[s]->num == 0 && mhs[s]->max_hash == 0)) flush(s);
    for (auto& a : acc) { if (a.out.p) (void)hipFree(a.out.p); if (a.cnt.p) (void)hipFree(a.cnt.p); a.out.p = a.cnt.p = nullptr; }
    for (int i = 0; i < 2; ++i) { (void)hipEventDestroy(done[i]); if (dseq[i].p) (void)hipFree(dseq[i].p); dseq[i].p = nullptr; }
}

}  // namespace smg
