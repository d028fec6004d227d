// pargz.hpp -- gzip inflate of ONE large member on many host threads.
//
// Why: `sourmash sketch` reads genome.fna.gz (src/sourmash/command_sketch.py:697,746-768 through screed; the Rust bench
// inflates with niffler, src/core/benches/compute.rs:35-38).  One zlib stream inflates ~0.2-0.4 GB/s: a single 1 GB .fna.gz
// fed the sketch kernel 1 % of what it can take (weakness 9).  A deflate stream has no index, but:
//   * block starts can be FOUND: a worker scans bit positions from its span's nominal start for a dynamic-Huffman block
//     header whose code-length tables are complete prefix codes, and confirms by inflating 128 KB from there;
//   * a block can be DECODED without the 32 KB of history in front of it, if window references are kept symbolic: the span
//     is inflated (zlib, raw mode, inflatePrime for the bit offset) against a preset dictionary whose bytes all have the top
//     bit set and encode bits 8..14 of their own position -- FASTA / FASTQ text has no byte >= 0x80, so exactly the output
//     bytes copied (directly or through later copies) out of the unknown window come out >= 0x80 -- and once more, as far
//     as such bytes reach, against a dictionary encoding bits 0..7: together the window position each of them stands for.
//     A DATA byte >= 0x80 (a UTF-8 header, a binary file) reads the same in both passes, and so do the markers of 128 window
//     positions (one per value of the high bits); the position code is permuted so that these are the OLDEST 128 bytes of the
//     window -- distances zlib-family encoders never emit (MAX_DIST = 32768 - 262) -- and a span that does hold such a byte
//     is inflated a third time against an all-zero dictionary, where a marker reads 0 and a data byte reads itself: a span
//     with a data byte >= 0x80 is refused BEFORE any of it is delivered, and the reader falls back;
//   * when the span in front is finished, its last 32 KB are the window and the marked bytes are filled in.
// Every span must end exactly on the bit where the next one begins (zlib's Z_BLOCK reports block ends), and the member's
// CRC-32 and length (gzip trailer) are checked over the assembled output: an input that is not pure 7-bit text, a
// mis-detected block start or anything else that does not add up makes the reader start over with the plain sequential
// zlib stream -- the result is the sequential result or an error, never something else.
//
// Multi-member files (bgzip) and files smaller than two spans take the sequential path as well.
#pragma once
#include <fcntl.h>
#include <sched.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace smg {

namespace pargz_detail {

constexpr uint32_t WIN = 32768;

// little-endian bit reader over a byte range (deflate's bit order)
struct Bits {
    const uint8_t* p;
    uint64_t n_bits, pos;
    bool ok(uint64_t need) const { return pos + need <= n_bits; }
    uint32_t get(int k) {
        uint32_t v = 0;
        for (int i = 0; i < k; ++i, ++pos) v |= (uint32_t)((p[pos >> 3] >> (pos & 7)) & 1u) << i;
        return v;
    }
};

// Kraft sum of a set of code lengths == 1 (complete) -- or a single code of length 1 for the distance tree (RFC 1951 3.2.7)
inline bool complete_code(const uint8_t* len, int n, int max_bits, bool allow_single) {
    uint32_t left = 1u << max_bits, used = 0;
    for (int i = 0; i < n; ++i)
        if (len[i]) {
            const uint32_t w = 1u << (max_bits - len[i]);
            if (w > left) return false;
            left -= w;
            ++used;
        }
    if (left == 0) return used > 0;
    return allow_single && used == 1 && left == (1u << (max_bits - 1));
}

// Does a dynamic-Huffman, non-final block header start at bit `bit`?  (cheap structural test; the caller confirms by inflating)
inline bool plausible_dynamic_header(const uint8_t* data, uint64_t n_bytes, uint64_t bit) {
    // Fast rejection on one unaligned 64-bit read (7 of 8 positions fail the 3 header bits, almost all others the Kraft
    // sum of the code-length code): the scan visits ~10^5 bit positions per block start it finds.
    if ((bit >> 3) + 16 < n_bytes) {
        uint64_t w0, w1;
        memcpy(&w0, data + (bit >> 3), 8);
        memcpy(&w1, data + (bit >> 3) + 8, 8);
        const unsigned sh = (unsigned)(bit & 7);
        const uint64_t w = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
        if ((w & 7u) != 4u) return false;                       // BFINAL = 0, BTYPE = 10b (bits: 0, then 0 1)
        const unsigned hlit = (unsigned)((w >> 3) & 31u), hdist = (unsigned)((w >> 8) & 31u), hclen = (unsigned)((w >> 13) & 15u) + 4;
        if (hlit > 29 || hdist > 29) return false;
        // Kraft sum over the code-length code (lengths 1..7): sum 2^(7 - len) must be exactly 128
        unsigned kraft = 0, nz = 0;
        uint64_t pos = bit + 17;
        for (unsigned i = 0; i < hclen; ++i, pos += 3) {
            const uint64_t byte = pos >> 3;
            const unsigned v = (unsigned)(((data[byte] | ((unsigned)data[byte + 1] << 8)) >> (pos & 7)) & 7u);
            if (v) { kraft += 1u << (7 - v); ++nz; }
        }
        if (kraft != 128u || nz == 0) return false;
    }
    Bits b{data, n_bytes * 8, bit};
    if (!b.ok(17)) return false;
    if (b.get(1) != 0) return false;                       // BFINAL: a span never starts at the last block
    if (b.get(2) != 2) return false;                       // BTYPE 10: dynamic Huffman
    const int hlit = (int)b.get(5) + 257, hdist = (int)b.get(5) + 1, hclen = (int)b.get(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    if (!b.ok((uint64_t)hclen * 3)) return false;
    for (int i = 0; i < hclen; ++i) cl[order[i]] = (uint8_t)b.get(3);
    if (!complete_code(cl, 19, 7, false)) return false;
    // canonical code of the code-length alphabet
    uint16_t count[8] = {0}, next[8] = {0}, code[19];
    for (int i = 0; i < 19; ++i) count[cl[i]]++;
    count[0] = 0;
    for (int l = 1, c = 0; l <= 7; ++l) { c = (c + count[l - 1]) << 1; next[l] = (uint16_t)c; }
    for (int i = 0; i < 19; ++i) code[i] = cl[i] ? next[cl[i]]++ : 0;
    uint8_t lens[286 + 30];
    int got = 0, prev = 0;
    while (got < hlit + hdist) {
        // decode one symbol bit by bit (MSB-first codes)
        uint32_t acc = 0;
        int sym = -1;
        for (int l = 1; l <= 7 && sym < 0; ++l) {
            if (!b.ok(1)) return false;
            acc = (acc << 1) | b.get(1);
            for (int i = 0; i < 19; ++i)
                if (cl[i] == l && code[i] == acc) { sym = i; break; }
        }
        if (sym < 0) return false;
        if (sym < 16) { lens[got++] = (uint8_t)sym; prev = sym; continue; }
        int rep, val = 0;
        if (sym == 16) { if (got == 0 || !b.ok(2)) return false; rep = 3 + (int)b.get(2); val = prev; }
        else if (sym == 17) { if (!b.ok(3)) return false; rep = 3 + (int)b.get(3); }
        else { if (!b.ok(7)) return false; rep = 11 + (int)b.get(7); }
        if (got + rep > hlit + hdist) return false;
        for (int i = 0; i < rep; ++i) lens[got++] = (uint8_t)val;
        prev = val;
    }
    if (lens[256] == 0) return false;                      // no end-of-block code
    if (!complete_code(lens, hlit, 15, false)) return false;
    if (!complete_code(lens + hlit, hdist, 15, true)) {
        bool none = true;
        for (int i = 0; i < hdist; ++i) none = none && lens[hlit + i] == 0;
        if (!none) return false;                           // (no distance codes at all is allowed: literals only)
    }
    return true;
}

struct RawInflate {
    z_stream zs;
    bool live = false;
    ~RawInflate() { if (live) inflateEnd(&zs); }
    // start a raw deflate decoder at bit `bit` of data[0, n) with a preset 32 KB window (may be null)
    bool open(const uint8_t* data, uint64_t n, uint64_t bit, const uint8_t* dict) {
        if (live) { inflateEnd(&zs); live = false; }
        memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, -15) != Z_OK) return false;
        live = true;
        if (dict && inflateSetDictionary(&zs, dict, WIN) != Z_OK) return false;
        const uint64_t byte = bit >> 3;
        const int skip = (int)(bit & 7);
        if (byte >= n) return false;
        uint64_t from = byte;
        if (skip) {
            if (inflatePrime(&zs, 8 - skip, data[byte] >> skip) != Z_OK) return false;
            from = byte + 1;
        }
        base_ = data;
        zs.next_in = const_cast<Bytef*>(data + from);
        avail_total_ = n - from;
        zs.avail_in = 0;
        return true;
    }
    // bit position (in the data passed to open) of the next unread bit
    uint64_t bit_pos() const { return (uint64_t)(zs.next_in - base_) * 8 - (uint64_t)(zs.data_type & 63); }
    void feed() {
        if (zs.avail_in == 0 && avail_total_) {
            const uint64_t k = std::min<uint64_t>(avail_total_, 1u << 30);
            zs.avail_in = (uInt)k;
            avail_total_ -= k;
        }
    }
    const uint8_t* base_ = nullptr;
    uint64_t avail_total_ = 0;
};

}  // namespace pargz_detail

// Decompressed bytes of a gzip file, in order, through read(); many threads when the file is one large member of text.
class ParallelGunzip {
  public:
    // span_bytes: compressed bytes per work unit; threads: 0 = hardware concurrency (at most 32)
    explicit ParallelGunzip(const std::string& path, unsigned threads = 0, size_t span_bytes = (size_t)1 << 20)
        : path_(path), span_(span_bytes) {
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) throw std::runtime_error("cannot open " + path);
        struct stat st;
        if (fstat(fd_, &st) != 0) { ::close(fd_); throw std::runtime_error("cannot stat " + path); }
        size_ = (uint64_t)st.st_size;
        if (threads == 0) threads = usable_cpus();
        n_threads_ = std::max(1u, std::min(threads, 32u));
        parallel_ = false;
        if (getenv("SMG_GUNZIP_SEQUENTIAL") == nullptr && n_threads_ > 1 && size_ >= 4 * span_) {
            map_ = (const uint8_t*)mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
            if (map_ == MAP_FAILED) map_ = nullptr;
            if (map_ && parse_header()) parallel_ = true;
        }
        if (parallel_) start_parallel();
        else open_sequential();
    }
    ~ParallelGunzip() {
        stop_workers();
        if (parallel_ && getenv("SMG_GUNZIP_TRACE"))
            fprintf(stderr, "[gunzip] %s: %u threads, %llu spans; worker ms: block search %.0f, first pass %.0f, second pass %.0f, third pass %.0f, waiting for "
                            "the span in front %.0f, crc %.0f; consumer ms: waiting %.0f, copying %.0f\n", path_.c_str(), n_threads_,
                    (unsigned long long)n_spans_, t_search_ * 1e-6, t_pass1_ * 1e-6, t_pass2_ * 1e-6, t_pass3_ * 1e-6, t_wait_tail_ * 1e-6, t_crc_ * 1e-6,
                    t_consumer_wait_ * 1e-6, t_copy_ * 1e-6);
        for (Span* c : spans_) delete c;
        for (Span* c : free_spans_) delete c;
        if (gzf_) gzclose(gzf_);
        if (map_) munmap((void*)map_, size_);
        if (fd_ >= 0) ::close(fd_);
    }
    ParallelGunzip(const ParallelGunzip&) = delete;
    ParallelGunzip& operator=(const ParallelGunzip&) = delete;

    // up to `want` decompressed bytes; 0 at the end of the stream; throws on a corrupt file
    size_t read(uint8_t* dst, size_t want) {
        if (!parallel_) return read_sequential(dst, want);
        size_t got = 0;
        while (got < want) {
            if (!cur_) {
                cur_ = next_ready();
                if (!cur_) {
                    if (fell_back_) return got + read_sequential(dst + got, want - got);
                    break;
                }
                cur_off_ = 0;
            }
            const size_t k = std::min(want - got, cur_->out.size() - cur_off_);
            const uint64_t tcp = now_ns();
            big_copy(dst + got, cur_->out.data() + cur_off_, k);
            t_copy_ += now_ns() - tcp;
            got += k;
            cur_off_ += k;
            if (cur_off_ == cur_->out.size()) { recycle(cur_); cur_ = nullptr; }
        }
        return got;
    }
    // The consumer is one thread: a plain memcpy of every span into the caller's buffer (pinned memory in the ingest path) would
    // cap the reader at one core's copy rate, a third of what sixteen inflating threads deliver.  Large copies go out in four pieces.
    static void big_copy(uint8_t* dst, const uint8_t* src, size_t n) {
        constexpr size_t PIECE = (size_t)2 << 20;
        if (n == 0) return;                                          // (an empty span has no buffer at all)
        if (n < 2 * PIECE) { memcpy(dst, src, n); return; }
        const unsigned parts = 4;
        const size_t per = (n / parts + 63) & ~(size_t)63;
        std::thread th[parts - 1];
        for (unsigned t = 1; t < parts; ++t) {
            const size_t lo = std::min(n, per * t), hi = t + 1 == parts ? n : std::min(n, per * (t + 1));
            th[t - 1] = std::thread([=] { if (hi > lo) memcpy(dst + lo, src + lo, hi - lo); });
        }
        memcpy(dst, src, std::min(n, per));
        for (unsigned t = 1; t < parts; ++t) th[t - 1].join();
    }
    bool parallel() const { return parallel_ && !fell_back_; }
    // why the parallel form was not used or was abandoned ("" otherwise)
    const std::string& fallback_reason() const { return reason_; }

    // CPUs this process may really use: affinity mask, capped by the cgroup's CPU quota (a container that sees 256 CPUs
    // may be granted 16)
    static unsigned usable_cpus() {
        unsigned n = std::thread::hardware_concurrency();
        if (n == 0) n = 1;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min<unsigned>(n, (unsigned)CPU_COUNT(&set));
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64] = {0};
            unsigned long period = 0;
            if (fscanf(f, "%63s %lu", q, &period) == 2 && strcmp(q, "max") != 0 && period) {
                const unsigned long quota = strtoul(q, nullptr, 10);
                if (quota) n = std::min<unsigned>(n, (unsigned)std::max<unsigned long>(1, (quota + period - 1) / period));
            }
            fclose(f);
        }
        return n;
    }

  private:
    static constexpr uint64_t AT_END = UINT64_MAX - 1;         // "no block start from here to the end of the stream"
    struct Span {
        uint64_t index = 0;
        uint64_t nominal_bit = 0;                // where the search for the block start began
        uint64_t start_bit = 0, end_bit = 0;     // [start, end) in deflate-stream bits; end = start of the next span
        bool final_span = false;                 // runs to the end of the deflate stream
        std::vector<uint8_t> out, lowbits;       // output with marker bytes; low position bits of the marked bytes' prefix
        uint64_t marked_until = 0;               // output offset behind the last marked byte
        uint32_t crc = 0;                        // CRC-32 of the resolved output
        bool ok = false;
        std::string why;
        bool done = false;
    };

    // ---- sequential fallback (the plain zlib stream) ----
    void open_sequential() {
        gzf_ = gzdopen(dup(fd_), "rb");
        if (!gzf_) throw std::runtime_error("cannot initialise gzip reader for " + path_);
        gzbuffer(gzf_, 1 << 20);
    }
    size_t read_sequential(uint8_t* dst, size_t want) {
        if (!gzf_) open_sequential();
        if (seq_skip_) {                                             // bytes the parallel form already delivered: read past them,
            std::vector<uint8_t> tmp(1 << 20);                       // and make sure they ARE what was delivered
            uint32_t crc = 0;
            while (seq_skip_) {
                const int r = gzread(gzf_, tmp.data(), (unsigned)std::min<uint64_t>(tmp.size(), seq_skip_));
                if (r <= 0) throw std::runtime_error("error while reading sequence file " + path_);
                crc = (uint32_t)crc32_z(crc, tmp.data(), (size_t)r);
                seq_skip_ -= (uint64_t)r;
            }
            if (crc != crc_)
                throw std::runtime_error("gzip reader: the bytes already delivered from " + path_ + " differ from the sequential stream "
                                         "(set SMG_GUNZIP_SEQUENTIAL=1)");
        }
        size_t got = 0;
        while (got < want) {
            const int r = gzread(gzf_, dst + got, (unsigned)std::min<size_t>(want - got, 1u << 30));
            if (r < 0) throw std::runtime_error("error while reading sequence file " + path_);
            if (r == 0) {
                int errnum = Z_OK;
                (void)gzerror(gzf_, &errnum);                         // a stream that stops short of its end marker reads as EOF + Z_BUF_ERROR
                if (errnum != Z_OK && errnum != Z_STREAM_END)
                    throw std::runtime_error("error while reading sequence file " + path_ + " (truncated or corrupt gzip stream)");
                break;
            }
            got += (size_t)r;
        }
        return got;
    }

    // ---- gzip framing ----
    bool parse_header() {
        const uint8_t* p = map_;
        if (size_ < 18 + 8 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8) return false;
        const uint8_t flg = p[3];
        uint64_t o = 10;
        if (flg & 4) { if (o + 2 > size_) return false; o += 2 + (uint64_t)(p[o] | (p[o + 1] << 8)); }
        if (flg & 8) { while (o < size_ && p[o]) ++o; ++o; }
        if (flg & 16) { while (o < size_ && p[o]) ++o; ++o; }
        if (flg & 2) o += 2;
        if (o + 8 >= size_) return false;
        deflate_ = map_ + o;
        deflate_len_ = size_ - o - 8;                                 // if this is the only member: everything up to the trailer
        want_crc_ = (uint32_t)map_[size_ - 8] | ((uint32_t)map_[size_ - 7] << 8) | ((uint32_t)map_[size_ - 6] << 16) | ((uint32_t)map_[size_ - 5] << 24);
        want_isize_ = (uint32_t)map_[size_ - 4] | ((uint32_t)map_[size_ - 3] << 8) | ((uint32_t)map_[size_ - 2] << 16) | ((uint32_t)map_[size_ - 1] << 24);
        return true;
    }

    // ---- the parallel machinery ----
    void start_parallel() {
        n_spans_ = std::max<uint64_t>(1, deflate_len_ / span_);       // the last span takes the remainder (between one and two spans:
                                                                      // a short tail may hold nothing but the final block)
        tails_.assign(n_spans_, std::vector<uint8_t>());
        tail_state_.assign(n_spans_, 0);
        in_flight_max_ = n_threads_ + 4;                             // (every span in flight is ~6 MB of freshly touched memory)
        static_dicts();
        for (unsigned t = 0; t < n_threads_; ++t) workers_.emplace_back([this] { work(); });
    }
    void stop_workers() {
        {
            std::lock_guard<std::mutex> g(mu_);
            abort_ = true;
        }
        cv_.notify_all();
        tcv_.notify_all();
        bcv_.notify_all();
        for (auto& t : workers_) t.join();
        workers_.clear();
    }
    void static_dicts() {
        dict_hi_.resize(pargz_detail::WIN);
        dict_lo_.resize(pargz_detail::WIN);
        dict_zero_.assign(pargz_detail::WIN, 0);
        for (uint32_t i = 0; i < pargz_detail::WIN; ++i) {
            const uint32_t c = position_code(i);
            dict_hi_[i] = (uint8_t)(0x80u | (c >> 8));               // top bit: "from the unknown window"; bits 8..14 of the code
            dict_lo_[i] = (uint8_t)(c & 0xffu);
        }
    }
    // Window position <-> 15-bit code (an involution).  The two marker bytes of code c are 0x80 | c >> 8 and c & 0xff; they are
    // EQUAL -- like the two readings of a data byte >= 0x80 -- for the 128 codes h << 8 | 0x80 | h.  Those codes go to window
    // positions 0..127, the oldest bytes of the window, which only a match at the very start of a span with a distance above
    // 32,640 can reach (zlib, pigz, bgzip never look back further than 32,506); every other position keeps its own number.
  public:
    static uint32_t position_code(uint32_t p) {
        if (p < 128u) return (p << 8) | 0x80u | p;
        const uint32_t h = p >> 8;
        if ((p & 0xffu) == (0x80u | h)) return h;
        return p;
    }
  private:

    // first confirmed block start at or after `bit` (searching at most `limit_bits` further); UINT64_MAX if none
    uint64_t find_block(uint64_t bit, uint64_t limit_bits) {
        using namespace pargz_detail;
        const uint64_t end = std::min<uint64_t>(bit + limit_bits, deflate_len_ * 8 - 64);
        std::vector<uint8_t> probe(128 << 10);
        for (uint64_t b = bit; b < end; ++b) {
            if (!plausible_dynamic_header(deflate_, deflate_len_, b)) continue;
            // confirm: 128 KB must inflate from here without an error (history: the marker dictionary)
            RawInflate ri;
            if (!ri.open(deflate_, deflate_len_, b, dict_hi_.data())) continue;
            ri.zs.next_out = probe.data();
            ri.zs.avail_out = (uInt)probe.size();
            int rc = Z_OK;
            while (rc == Z_OK && ri.zs.avail_out) {
                ri.feed();
                rc = inflate(&ri.zs, Z_NO_FLUSH);
                if (rc == Z_BUF_ERROR && ri.zs.avail_in == 0 && ri.avail_total_ == 0) break;
            }
            if (rc == Z_OK || rc == Z_STREAM_END || (rc == Z_BUF_ERROR && ri.zs.avail_out == 0)) return b;
        }
        return UINT64_MAX;
    }

    // start of span i (i >= 1): the first confirmed block start at or after its nominal position; searched once, by whichever
    // worker asks first (the span's own worker and the one in front of it both need it)
    uint64_t boundary(uint64_t i) {
        {
            std::unique_lock<std::mutex> lk(bmu_);
            if (bounds_.size() < n_spans_ + 1) { bounds_.assign(n_spans_ + 1, 0); bstate_.assign(n_spans_ + 1, 0); }
            bcv_.wait(lk, [&] { return bstate_[i] != 1 || failed_ || abort_; });
            if (bstate_[i] == 2) return bounds_[i];
            if (bstate_[i] == 1) return UINT64_MAX;                  // the worker that was searching gave up: so does this span
            bstate_[i] = 1;
        }
        // (a block of 16,384 symbols can be longer than a small span: look up to 256 KB ahead; two spans may then share a start,
        //  and the one in front is empty)
        const uint64_t look = std::max<uint64_t>(span_, 256u << 10) * 8;
        const uint64_t ts = now_ns();
        uint64_t b = find_block(i * span_ * 8, look);
        t_search_ += now_ns() - ts;
        // nothing up to the end of the stream: what is left holds no further (non-final) block start -- the span in front
        // runs to the end, and this one and the ones behind it are empty
        if (b == UINT64_MAX && i * span_ * 8 + look + 64 >= deflate_len_ * 8) b = AT_END;
        {
            std::lock_guard<std::mutex> g(bmu_);
            bounds_[i] = b;
            bstate_[i] = 2;
        }
        bcv_.notify_all();
        return b;
    }

    // Inflate span s (start_bit known) until the block boundary that is span s+1's start.
    void decode(Span& s) {
        using namespace pargz_detail;
        s.ok = false;
        // where must this span stop?  the first confirmed block start at or after the next nominal boundary
        uint64_t stop = UINT64_MAX;
        if (!s.final_span) {
            stop = boundary(s.index + 1);
            if (stop == UINT64_MAX) { s.why = "no block start found for span " + std::to_string(s.index + 1); return; }
            if (stop == AT_END) { s.final_span = true; stop = UINT64_MAX; }
        }
        s.end_bit = stop;
        const bool first = s.index == 0;
        if (!first && s.start_bit == stop) {                         // nothing of its own: the next span starts at the same block
            s.out.clear();
            s.marked_until = 0;
            s.lowbits.clear();
            s.ok = true;
            return;
        }
        if (!first && s.start_bit > stop) { s.why = "span boundaries out of order"; return; }
        RawInflate ri;
        if (!ri.open(deflate_, deflate_len_, s.start_bit, first ? nullptr : dict_hi_.data())) { s.why = "inflate init"; return; }
        const size_t cap = span_ * 5 + (1 << 20);
        if (s.out.size() < cap) s.out.resize(cap);                 // (a recycled span keeps its buffer: no zero-fill, no page faults)
        size_t produced = 0;
        bool reached = false;
        const uint64_t t1 = now_ns();
        for (;;) {
            if (produced == s.out.size()) s.out.resize(s.out.size() * 2);
            ri.zs.next_out = s.out.data() + produced;
            ri.zs.avail_out = (uInt)std::min<size_t>(s.out.size() - produced, 1u << 30);
            ri.feed();
            const size_t before = ri.zs.avail_out;
            const int rc = inflate(&ri.zs, Z_BLOCK);
            produced += before - ri.zs.avail_out;
            if (rc == Z_STREAM_END) {
                if (!s.final_span) { s.why = "stream ended inside span " + std::to_string(s.index); return; }
                // the deflate stream must end exactly where the trailer begins (a single member)
                const uint64_t used = (uint64_t)(ri.zs.next_in - deflate_);
                if (used != deflate_len_) { s.why = "more than one gzip member"; return; }
                reached = true;
                break;
            }
            if (rc != Z_OK && rc != Z_BUF_ERROR) { s.why = "corrupt deflate data in span " + std::to_string(s.index); return; }
            if (rc == Z_BUF_ERROR && ri.zs.avail_in == 0 && ri.avail_total_ == 0 && ri.zs.avail_out) { s.why = "truncated deflate stream"; return; }
            if ((ri.zs.data_type & 128) && !(ri.zs.data_type & 64)) {           // at a block boundary
                const uint64_t at = ri.bit_pos();
                if (at == stop) { reached = true; break; }
                if (at > stop) { s.why = "span " + std::to_string(s.index) + " ran past the start found for the next one"; return; }
            }
        }
        if (!reached) { s.why = "span did not end on a block boundary"; return; }
        t_pass1_ += now_ns() - t1;
        s.out.resize(produced);
        s.marked_until = 0;
        s.lowbits.clear();
        if (!first) {
            // marked bytes (>= 0x80) stand for window positions; how far do they reach?
            size_t last = 0;
            const uint8_t* o = s.out.data();
            {   // eight bytes at a time from the back: the marked bytes sit near the front of a span
                size_t i = produced;
                while (i > 0 && (i & 7)) { --i; if (o[i] & 0x80u) { last = i + 1; break; } }
                while (!last && i >= 8) {
                    uint64_t w;
                    memcpy(&w, o + i - 8, 8);
                    if (w & 0x8080808080808080ull) {
                        for (size_t k = i; k > i - 8; --k)
                            if (o[k - 1] & 0x80u) { last = k; break; }
                        break;
                    }
                    i -= 8;
                }
            }
            s.marked_until = last;
            const uint64_t t2 = now_ns();
            if (last) {
                // second pass over that prefix with the low-bits dictionary
                RawInflate r2;
                if (!r2.open(deflate_, deflate_len_, s.start_bit, dict_lo_.data())) { s.why = "inflate init"; return; }
                s.lowbits.resize(last);
                size_t got = 0;
                while (got < last) {
                    r2.zs.next_out = s.lowbits.data() + got;
                    r2.zs.avail_out = (uInt)std::min<size_t>(last - got, 1u << 30);
                    r2.feed();
                    const size_t before = r2.zs.avail_out;
                    const int rc = inflate(&r2.zs, Z_NO_FLUSH);
                    got += before - r2.zs.avail_out;
                    if (rc == Z_STREAM_END) break;
                    if (rc != Z_OK && !(rc == Z_BUF_ERROR && r2.zs.avail_out == 0)) { s.why = "second pass failed"; return; }
                }
                if (got < last) { s.why = "second pass came up short"; return; }
                t_pass2_ += now_ns() - t2;
                // A byte that reads the same (>= 0x80) in both passes is either a data byte of the file or the marker of one
                // of the 128 oldest window positions: a third pass against zeros tells them apart (marker -> 0, data -> itself).
                size_t amb_end = 0;
                for (size_t i = last; i > 0; --i)
                    if ((o[i - 1] & 0x80u) && s.lowbits[i - 1] == o[i - 1]) { amb_end = i; break; }
                if (amb_end) {
                    const uint64_t t3 = now_ns();
                    RawInflate r3;
                    if (!r3.open(deflate_, deflate_len_, s.start_bit, dict_zero_.data())) { s.why = "inflate init"; return; }
                    std::vector<uint8_t> plain(amb_end);
                    size_t got3 = 0;
                    while (got3 < amb_end) {
                        r3.zs.next_out = plain.data() + got3;
                        r3.zs.avail_out = (uInt)std::min<size_t>(amb_end - got3, 1u << 30);
                        r3.feed();
                        const size_t before = r3.zs.avail_out;
                        const int rc = inflate(&r3.zs, Z_NO_FLUSH);
                        got3 += before - r3.zs.avail_out;
                        if (rc == Z_STREAM_END) break;
                        if (rc != Z_OK && !(rc == Z_BUF_ERROR && r3.zs.avail_out == 0)) { s.why = "third pass failed"; return; }
                    }
                    if (got3 < amb_end) { s.why = "third pass came up short"; return; }
                    for (size_t i = 0; i < amb_end; ++i)
                        if ((o[i] & 0x80u) && s.lowbits[i] == o[i] && plain[i] != 0) {
                            s.why = "the file holds bytes >= 0x80 (not 7-bit text)";
                            return;
                        }
                    t_pass3_ += now_ns() - t3;
                }
            }
        }
        s.ok = true;
    }

    // fill the marked bytes of s from the 32 KB in front of it (the resolved tail of the previous span)
    static void resolve(Span& s, const std::vector<uint8_t>& window, size_t from, size_t to) {
        uint8_t* o = s.out.data();
        if (to > (size_t)s.marked_until) to = (size_t)s.marked_until;
        for (size_t i = from; i < to; ++i)
            if (o[i] & 0x80u) {
                const uint32_t pos = position_code(((uint32_t)(o[i] & 0x7fu) << 8) | s.lowbits[i]);
                o[i] = window[pos];
            }
    }

    // Second half of a span's work, still on the worker: wait for the (final) last 32 KB of the span in front, fill the
    // marked bytes in, publish this span's own last 32 KB for the next one, and checksum the span.  Only the few marked bytes
    // make spans depend on each other; inflating and checksumming run on all threads.
    void finish(Span& s) {
        using namespace pargz_detail;
        std::vector<uint8_t> window;
        const uint64_t tw = now_ns();
        if (s.index > 0) {
            std::unique_lock<std::mutex> lk(tmu_);
            tcv_.wait(lk, [&] { return tail_state_[s.index - 1] != 0 || abort_ || failed_; });
            if (tail_state_[s.index - 1] != 1) s.ok = false;            // the span in front failed (or everything was called off)
            else window = tails_[s.index - 1];
        }
        t_wait_tail_ += now_ns() - tw;
        // The spans depend on each other only through their last 32 KB: those are filled in and published FIRST (a few
        // microseconds per link of the chain); the rest of the span follows while the next span is already being resolved.
        const size_t n_out = s.out.size();
        const size_t tail_from = n_out >= WIN ? n_out - WIN : 0;
        if (s.ok && s.index > 0) resolve(s, window, tail_from, n_out);
        {
            std::vector<uint8_t> tail;
            bool tail_ok = s.ok;
            if (tail_ok) {
                uint64_t any = 0;
                for (size_t i = tail_from; i < n_out; ++i) any |= s.out[i];
                if (any & 0x80u) { s.ok = tail_ok = false; s.why = "the file holds bytes >= 0x80 (not 7-bit text)"; }
            }
            if (tail_ok) {
                if (n_out >= WIN) tail.assign(s.out.end() - WIN, s.out.end());
                else {
                    tail.assign(WIN, 0);
                    const size_t keep = WIN - n_out;
                    if (!window.empty()) memcpy(tail.data(), window.data() + (WIN - keep), keep);
                    if (n_out) memcpy(tail.data() + keep, s.out.data(), n_out);   // (an empty span: the tail is the window)
                }
            }
            {
                std::lock_guard<std::mutex> g(tmu_);
                tails_[s.index].swap(tail);
                tail_state_[s.index] = tail_ok ? 1 : 2;
                if (s.index > 0) std::vector<uint8_t>().swap(tails_[s.index - 1]);   // nobody else needs it
            }
            tcv_.notify_all();
        }
        if (s.ok) {
            if (s.index > 0) resolve(s, window, 0, tail_from);
            // anything still >= 0x80 means the input is not 7-bit text and the marker scheme does not apply (span 0 is inflated
            // with its true history, so the whole of it is screened; later spans: the bytes that were marked)
            const uint8_t* o = s.out.data();
            const size_t n_check = s.index == 0 ? s.out.size() : (size_t)s.marked_until;
            uint64_t any = 0;
            size_t i = 0;
            for (; i + 8 <= n_check; i += 8) { uint64_t w; memcpy(&w, o + i, 8); any |= w; }
            for (; i < n_check; ++i) any |= (uint64_t)o[i] << 0;
            if (any & 0x8080808080808080ull) { s.ok = false; s.why = "the file holds bytes >= 0x80 (not 7-bit text)"; }
        }
        const uint64_t tc = now_ns();
        if (s.ok) s.crc = (uint32_t)crc32_z(0, s.out.data(), s.out.size());
        t_crc_ += now_ns() - tc;
    }

    void work() {
        try {
            work_loop();
        } catch (...) {                                              // (out of memory in a worker: the consumer falls back or fails)
            {
                std::lock_guard<std::mutex> g(mu_);
                failed_ = true;
                for (Span* c : spans_)
                    if (!c->done) { c->ok = false; c->why = "a worker thread failed"; c->done = true; }
            }
            cv_.notify_all();
            tcv_.notify_all();
            bcv_.notify_all();
        }
    }
    void work_loop() {
        for (;;) {
            Span* s = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return abort_ || failed_ || (next_span_ < n_spans_ && in_flight_ < in_flight_max_); });
                if (abort_ || failed_ || next_span_ >= n_spans_) return;
                if (!free_spans_.empty()) { s = free_spans_.back(); free_spans_.pop_back(); }   // (buffers keep their capacity)
                else s = new Span();
                s->index = next_span_++;
                s->ok = s->done = false;
                s->why.clear();
                ++in_flight_;
                spans_.push_back(s);
            }
            s->final_span = s->index + 1 == n_spans_;
            if (s->index == 0) s->start_bit = 0;
            else {
                s->start_bit = boundary(s->index);
                if (s->start_bit == UINT64_MAX) s->why = "no block start found for span " + std::to_string(s->index);
            }
            if (s->start_bit == AT_END) {                           // behind the stream's last block start: nothing to do
                s->out.clear(); s->lowbits.clear(); s->marked_until = 0;
                s->end_bit = AT_END;
                s->ok = true;
            } else if (s->start_bit != UINT64_MAX) decode(*s);
            finish(*s);
            {
                std::lock_guard<std::mutex> g(mu_);
                s->done = true;
            }
            cv_.notify_all();
        }
    }

    // the next span in order, resolved and checked; nullptr at the end (or after a fallback was set up)
    Span* next_ready() {
        if (finished_) return nullptr;
        Span* s = nullptr;
        const uint64_t tcw = now_ns();
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] {
                for (Span* c : spans_)
                    if (c->index == deliver_ && c->done) { s = c; return true; }
                return false;
            });
        }
        t_consumer_wait_ += now_ns() - tcw;
        // a span must begin where the one in front ended (both found the same block start by themselves)
        if (!s->ok || (deliver_ > 0 && s->start_bit != prev_end_bit_)) {
            give_up(s->ok ? "span " + std::to_string(deliver_) + " does not begin where the one in front of it ended" : s->why);
            return nullptr;
        }
        crc_ = (uint32_t)crc32_combine(crc_, s->crc, (z_off_t)s->out.size());
        total_out_ += s->out.size();
        prev_end_bit_ = s->end_bit;
        ++deliver_;
        if (s->final_span) {
            finished_ = true;
            if (crc_ != want_crc_ || (uint32_t)(total_out_ & 0xffffffffu) != want_isize_)
                throw std::runtime_error("gzip checksum mismatch in " + path_ + " (parallel inflate of a member that is not 7-bit text?); "
                                         "set SMG_GUNZIP_SEQUENTIAL=1");
        }
        return s;
    }
    void recycle(Span* s) {
        {
            std::lock_guard<std::mutex> g(mu_);
            spans_.erase(std::find(spans_.begin(), spans_.end(), s));
            --in_flight_;
            free_spans_.push_back(s);
        }
        cv_.notify_all();
    }
    // abandon the parallel form: everything delivered so far was correct (spans are verified in order), so the
    // sequential stream skips that many bytes and carries on
    void give_up(const std::string& why) {
        reason_ = why;
        if (getenv("SMG_GUNZIP_TRACE")) fprintf(stderr, "[gunzip] %s: back to the sequential stream: %s\n", path_.c_str(), why.c_str());
        {
            std::lock_guard<std::mutex> g(mu_);
            failed_ = true;
        }
        cv_.notify_all();
        tcv_.notify_all();
        for (auto& t : workers_) t.join();
        workers_.clear();
        for (Span* c : spans_) delete c;
        spans_.clear();
        fell_back_ = true;
        finished_ = true;
        seq_skip_ = total_out_;
    }

    std::string path_, reason_;
    size_t span_;
    int fd_ = -1;
    uint64_t size_ = 0;
    unsigned n_threads_ = 1;
    bool parallel_ = false, fell_back_ = false, finished_ = false;
    const uint8_t* map_ = nullptr;
    const uint8_t* deflate_ = nullptr;
    uint64_t deflate_len_ = 0;
    uint32_t want_crc_ = 0, want_isize_ = 0, crc_ = 0;
    uint64_t total_out_ = 0, seq_skip_ = 0;
    gzFile gzf_ = nullptr;
    std::vector<uint8_t> dict_hi_, dict_lo_, dict_zero_, window_;
    uint64_t n_spans_ = 0, next_span_ = 0, deliver_ = 0, prev_end_bit_ = 0;
    unsigned in_flight_ = 0, in_flight_max_ = 4;
    std::vector<Span*> spans_, free_spans_;
    std::vector<std::thread> workers_;
    std::mutex mu_, bmu_, tmu_;
    std::condition_variable cv_, bcv_, tcv_;
    std::vector<std::vector<uint8_t>> tails_;   // final last 32 KB of every finished span (the next span's window)
    std::vector<uint8_t> tail_state_;           // 0 not yet, 1 there, 2 the span failed
    std::vector<uint64_t> bounds_;
    std::vector<uint8_t> bstate_;               // 0 unknown, 1 being searched, 2 known
    std::atomic<bool> abort_{false}, failed_{false};
    std::atomic<uint64_t> t_search_{0}, t_pass1_{0}, t_scan_{0}, t_pass2_{0}, t_pass3_{0}, t_wait_tail_{0}, t_crc_{0}, t_consumer_wait_{0}, t_copy_{0};
    static uint64_t now_ns() {
        return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    Span* cur_ = nullptr;
    size_t cur_off_ = 0;
};

}  // namespace smg
