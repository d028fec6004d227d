// capi.cpp -- the extern "C" surface of libsourmash_amd.so (include/sourmash_amd.h).
//
// Part 1 re-implements the hot-path subset of the reference's FFI shims
// (src/core/src/ffi/{utils,mod,minhash,signature,cmd/compute}.rs) on top of the
// host containers (minhash_host.hpp, signature_host.hpp) and the HIP kernels
// (DeviceCtx / device_api.hpp).  Part 2 are the smgpu_* batch extensions.
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
#include <cmath>
#include <fstream>
#include <memory>
#include <sstream>
#include <thread>
#include "../../include/sourmash_amd.h"
#include "collection.hpp"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include "device_ctx.hpp"
#include "hostxfer.hpp"
#include "pargz.hpp"
#include "ingest.hpp"
#include "sigload.hpp"
#include "murmur3.hpp"
#include "residues.hpp"
#include "signature_host.hpp"

using namespace smg;

// ---------------------------------------------------------------------------------------------
// thread-local last error + landing pad (ffi/utils.rs:17-19,58-83,195-207)
// ---------------------------------------------------------------------------------------------
namespace {

thread_local uint32_t g_err_code = 0;
thread_local std::string g_err_msg;

void set_error(uint32_t code, const std::string& msg) { g_err_code = code; g_err_msg = msg; }

template <class R, class F>
R landing(F&& f) {
    try {
        return f();
    } catch (const Error& e) {
        set_error(e.code, e.what());
    } catch (const std::bad_alloc&) {
        set_error(E_PANIC, "sourmash panicked: out of memory");
    } catch (const std::exception& e) {
        set_error(E_PANIC, std::string("sourmash panicked: ") + e.what());
    } catch (...) {
        set_error(E_PANIC, "sourmash panicked: unknown exception");
    }
    return R{};
}
template <class F>
void landing_void(F&& f) {
    landing<int>([&]() { f(); return 0; });
}

SourmashStr make_str(const std::string& s) {
    SourmashStr out;
    out.data = (char*)malloc(s.size() + 1);
    memcpy(out.data, s.data(), s.size());
    out.data[s.size()] = 0;
    out.len = s.size();
    out.owned = true;
    return out;
}

// ---- deferred sketching behind the per-record API -------------------------------------------------------------
// The reference's loop is one add_sequence per record (src/sourmash/command_sketch.py:746-768, signature.rs:38-58).
// A kernel launch per 150-base read would cost ~40 us of copy + launch + synchronise each, so add_sequence only
// validates and appends the record to the sketch's `pending` buffer; the buffer goes through the sketch kernel in one
// launch when it is large enough or when anything looks at the sketch (every accessor below goes through MH / SIG,
// which settle first).  Results are those of immediate hashing: adding hashes to a sketch commutes (set union, counts
// add, bottom-k keeps the smallest), and everything that does not commute with it settles first.
constexpr size_t PENDING_FLUSH_BYTES = (size_t)32 << 20;
void settle(KmerMinHash& mh, bool streaming = false);

inline KmerMinHash* RAW(SourmashKmerMinHash* p) { return reinterpret_cast<KmerMinHash*>(p); }
inline const KmerMinHash* RAW(const SourmashKmerMinHash* p) { return reinterpret_cast<const KmerMinHash*>(p); }
inline KmerMinHash* settled(KmerMinHash* m) {
    if (m && !m->pending.empty()) {
        // accessors without a landing pad cannot throw across the C boundary: the failure is left in the thread's error slot
        try { settle(*m); }
        catch (const Error& e) { set_error(e.code, e.what()); }
        catch (const std::exception& e) { set_error(E_PANIC, std::string("sourmash panicked: ") + e.what()); }
    }
    return m;
}
inline KmerMinHash* MH(SourmashKmerMinHash* p) { return settled(reinterpret_cast<KmerMinHash*>(p)); }
inline const KmerMinHash* MH(const SourmashKmerMinHash* p) { return settled(const_cast<KmerMinHash*>(reinterpret_cast<const KmerMinHash*>(p))); }
inline Signature* SIGRAW(SourmashSignature* p) { return reinterpret_cast<Signature*>(p); }
inline Signature* SIG(SourmashSignature* p) {
    Signature* s = reinterpret_cast<Signature*>(p);
    if (s) for (auto& mh : s->sketches) settled(&mh);
    return s;
}
inline const Signature* SIG(const SourmashSignature* p) { return SIG(const_cast<SourmashSignature*>(p)); }
inline ComputeParameters* CP(SourmashComputeParameters* p) { return reinterpret_cast<ComputeParameters*>(p); }
inline const ComputeParameters* CP(const SourmashComputeParameters* p) { return reinterpret_cast<const ComputeParameters*>(p); }

uint64_t* slice_out(const std::vector<uint64_t>& v, uintptr_t* size) {
    *size = v.size();
    uint64_t* p = (uint64_t*)malloc((v.size() ? v.size() : 1) * sizeof(uint64_t));
    if (v.size()) memcpy(p, v.data(), v.size() * sizeof(uint64_t));
    return p;
}

std::string upper_ascii(const uint8_t* p, size_t n) {
    std::string s((const char*)p, n);
    for (auto& c : s) if (c >= 'a' && c <= 'z') c = (char)(c - 32);
    return s;
}

inline uint64_t keep_threshold(const KmerMinHash& mh) { return mh.max_hash ? mh.max_hash : ~0ull; }

// The DNA add_sequence path (signature.rs:38-58 + :246-306) on the GPU.
// force == false: the walk is streaming in the reference, so the hashes of every
// k-mer before the first offending one are added before InvalidDNA is raised.
// protein / dayhoff / hp sketches: residues (is_protein) or DNA translated in six frames; no validity test in
// either mode (signature.rs:307-393), so `force` plays no role
void add_residue_kmers(KmerMinHash& mh, const uint8_t* seq, size_t len, bool is_protein) {
    if (mh.num == 0 && mh.max_hash == 0) return;
    if (mh.ksize / 3 == 0 || len < mh.ksize / 3) return;
    if (mh.is_dna())                                                // signature.rs:367-384: no alphabet to map to
        throw Error(E_INVALID_HASH_FUNCTION, "Invalid hash function: \"DNA\"");
    DeviceCtx& ctx = DeviceCtx::get();
    std::lock_guard<std::recursive_mutex> g(ctx.mutex());
    std::vector<uint64_t> hs, cs;
    ctx.protein_sketch_host(seq, len, mh.ksize, mh.hash_function, mh.seed, is_protein, keep_threshold(mh),
                            mh.track_abundance, mh.num, hs, cs);
    mh.add_sorted_batch(hs.data(), mh.track_abundance ? cs.data() : nullptr, hs.size());
}

// first byte outside ACGTacgt (what kills a k-mer, encodings.rs:370-377 after signature.rs:214's upper-casing), or SIZE_MAX.
// Only decides whether add_sequence(force = false) must raise and where the valid prefix ends; the k-mers themselves
// are validated again, hashed and filtered by the kernel.
size_t first_invalid_byte(const uint8_t* seq, size_t len) {
    static const struct Table {
        uint8_t bad[256];
        Table() { memset(bad, 1, sizeof(bad)); for (const char* c = "ACGTacgt"; *c; ++c) bad[(uint8_t)*c] = 0; }
    } t;
    size_t i = 0;
#if defined(__SSE2__)
    // sixteen bytes at a time (SSE2, part of every x86-64): fold the case bit away, then a byte is fine iff it equals one
    // of A C G T (a byte >= 0x80 keeps its top bit and equals none of them)
    const __m128i fold = _mm_set1_epi8((char)0xdf), cA = _mm_set1_epi8('A'), cC = _mm_set1_epi8('C'), cG = _mm_set1_epi8('G'),
                  cT = _mm_set1_epi8('T');
    for (; i + 16 <= len; i += 16) {
        const __m128i x = _mm_and_si128(_mm_loadu_si128(reinterpret_cast<const __m128i*>(seq + i)), fold);
        const __m128i ok = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(x, cA), _mm_cmpeq_epi8(x, cC)),
                                        _mm_or_si128(_mm_cmpeq_epi8(x, cG), _mm_cmpeq_epi8(x, cT)));
        const unsigned m = (unsigned)_mm_movemask_epi8(ok);
        if (m != 0xffffu) return i + (size_t)__builtin_ctz(~m & 0xffffu);
    }
#endif
    for (; i < len; ++i)
        if (t.bad[seq[i]]) return i;
    return SIZE_MAX;
}

void settle(KmerMinHash& mh, bool streaming) {
    // Accessors that only read (get_mins, md5sum, similarity ...) come through here too, possibly from two threads on
    // the same sketch (ctypes releases the GIL): one of them hashes the queue, the other finds it empty afterwards.
    static std::recursive_mutex settle_mu;
    std::lock_guard<std::recursive_mutex> sg(settle_mu);
    if (mh.pending.empty()) return;
    // whatever happens below, the records are consumed once.  A flush from inside the add_sequence loop (`streaming`)
    // keeps the queue's storage (a fresh 32 MiB string per flush costs 8,192 page faults, which was a fifth of the
    // per-record time of a loop over 150-bp reads); a flush because somebody looks at the sketch gives storage beyond
    // 1 MiB back, or every sketch of a many-ksize signature would hold a copy of its longest record for life.
    struct Consume {
        std::string& q;
        bool keep;
        ~Consume() {
            if (keep || q.capacity() <= ((size_t)1 << 20)) q.clear();
            else std::string().swap(q);
        }
    } consume{mh.pending, streaming};
    DeviceCtx& ctx = DeviceCtx::get();
    std::lock_guard<std::recursive_mutex> g(ctx.mutex());
    std::vector<uint64_t> hs, cs;
    ctx.sketch_host((const uint8_t*)mh.pending.data(), mh.pending.size(), mh.ksize, mh.seed, keep_threshold(mh), mh.track_abundance,
                    mh.num, hs, cs);
    mh.add_sorted_batch(hs.data(), mh.track_abundance ? cs.data() : nullptr, hs.size());
}

// The DNA add_sequence path (signature.rs:38-58 + :246-306).
// force == false: the walk is streaming in the reference, so the hashes of every k-mer before the first offending
// one are added before InvalidDNA is raised: the valid prefix is queued, then the error is raised from this call.
// `scanned`: the caller has run first_invalid_byte over seq[0, len) already and passes its result in `first_bad`
void add_sequence_dna(KmerMinHash& mh, const uint8_t* seq, size_t len, bool force, bool scanned = false, size_t first_bad = SIZE_MAX) {
    if (!mh.is_dna()) { add_residue_kmers(mh, seq, len, false); return; }
    const uint32_t k = mh.ksize;
    if (len < k || k == 0) return;                                  // signature.rs:206-210
    if (mh.num == 0 && mh.max_hash == 0) return;                    // sketch that can never hold anything
    (void)DeviceCtx::get();                                         // no device: fail now, not at the first accessor
    check_dna_ksize(k);
    size_t use_len = len;
    bool raise = false;
    size_t bad_kmer = 0;
    if (!force) {
        const size_t p = scanned ? first_bad : first_invalid_byte(seq, len);
        if (p != SIZE_MAX) {
            bad_kmer = p + 1 >= k ? p + 1 - k : 0;                  // first k-mer whose window covers byte p
            if (bad_kmer < len - k + 1) {
                raise = true;
                use_len = bad_kmer + k - 1;                         // k-mers 0 .. bad_kmer-1 only
            }
        }
    }
    if (use_len >= k) {
        mh.pending.append((const char*)seq, use_len);
        mh.pending.push_back('\n');                                 // records never share a k-mer
        if (mh.pending.size() >= PENDING_FLUSH_BYTES) settle(mh, true);
    }
    if (raise) throw err_invalid_dna(upper_ascii(seq + bad_kmer, k));   // errors.rs:49-50
}

struct Downsampled {
    const KmerMinHash* a;
    const KmerMinHash* b;
    KmerMinHash tmp;
};
// minhash.rs:540-548 / 688-696: the finer sketch is downsampled to the coarser scaled
void align_scaled(const KmerMinHash& x, const KmerMinHash& y, bool downsample, Downsampled& d) {
    d.a = &x; d.b = &y;
    if (downsample && x.scaled() != y.scaled()) {
        if (x.scaled() > y.scaled()) { d.tmp = y.downsample_scaled(x.scaled()); d.b = &d.tmp; }
        else { d.tmp = x.downsample_scaled(y.scaled()); d.a = &d.tmp; }
    }
}

PairStats device_pair(const KmerMinHash& a, const KmerMinHash& b, bool want_abund, bool want_list, uint64_t num,
                      std::vector<uint64_t>* list) {
    DeviceCtx& ctx = DeviceCtx::get();
    std::lock_guard<std::recursive_mutex> g(ctx.mutex());
    return ctx.pair(a, b, want_abund, want_list, num, list);
}

// minhash.rs:593-621 -> (common, union)
std::pair<uint64_t, uint64_t> intersection_size(const KmerMinHash& a, const KmerMinHash& b) {
    a.check_compatible(b);
    if (a.num != 0) {
        const PairStats st = device_pair(a, b, false, false, a.num, nullptr);
        const uint64_t uni_all = a.size() + b.size() - st.common;
        return {st.common_num, uni_all < a.num ? uni_all : a.num};
    }
    const PairStats st = device_pair(a, b, false, false, 0, nullptr);
    return {st.common, a.size() + b.size() - st.common};
}

double jaccard(const KmerMinHash& a, const KmerMinHash& b) {      // minhash.rs:624-631
    const auto cu = intersection_size(a, b);
    return (double)cu.first / (double)(cu.second > 1 ? cu.second : 1);
}

double angular(const KmerMinHash& a, const KmerMinHash& b) {      // minhash.rs:635-680
    a.check_compatible(b);
    if (!a.track_abundance || !b.track_abundance) throw err_needs_abundance();
    const PairStats st = device_pair(a, b, true, false, 0, nullptr);
    const double na = std::sqrt((double)st.a_sq), nb = std::sqrt((double)st.b_sq);
    if (na == 0.0 || nb == 0.0) return 0.0;
    double p = (double)st.prod / (na * nb);
    if (p > 1.0) p = 1.0;
    return 1.0 - 2.0 * std::acos(p) / 3.14159265358979323846264338327950288;
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------------------------------------
// library / errors
// ---------------------------------------------------------------------------------------------
void sourmash_init(void) {}
void sourmash_err_clear(void) { g_err_code = 0; g_err_msg.clear(); }
SourmashErrorCode sourmash_err_get_last_code(void) { return g_err_code; }
SourmashStr sourmash_err_get_last_message(void) {
    if (g_err_code == 0) { SourmashStr s = {nullptr, 0, false}; return s; }
    return make_str(g_err_msg);
}
SourmashStr sourmash_err_get_backtrace(void) { SourmashStr s = {nullptr, 0, false}; return s; }
void sourmash_str_free(SourmashStr* s) {
    if (s && s->owned && s->data) { free(s->data); s->data = nullptr; s->len = 0; s->owned = false; }
}
// module-level residue helpers (ffi/mod.rs; encodings.rs:299-347)
char sourmash_aa_to_dayhoff(char aa) { return (char)aa_to_dayhoff((uint8_t)aa); }
char sourmash_aa_to_hp(char aa) { return (char)aa_to_hp((uint8_t)aa); }
char sourmash_translate_codon(const char* codon) {
    return landing<char>([&]() -> char {
        const size_t n = codon ? strlen(codon) : 0;
        if (n == 1) return 'X';                                                 // encodings.rs:309-311
        if (n == 2) return (char)translate_codon(ascii_upper((uint8_t)codon[0]), ascii_upper((uint8_t)codon[1]), 'N');
        if (n == 3) return (char)translate_codon(ascii_upper((uint8_t)codon[0]), ascii_upper((uint8_t)codon[1]),
                                                 ascii_upper((uint8_t)codon[2]));
        throw Error(E_INVALID_CODON_LENGTH, std::to_string(n));
    });
}
SourmashStr sourmash_str_from_cstr(const char* s) { return make_str(s ? s : ""); }

uint64_t hash_murmur(const char* kmer, uint64_t seed) {           // ffi/mod.rs:22-31
    if (!kmer) return 0;
    return mmh3_h1_bytes((const uint8_t*)kmer, strlen(kmer), seed);
}

// ---------------------------------------------------------------------------------------------
// compute parameters (ffi/cmd/compute.rs)
// ---------------------------------------------------------------------------------------------
SourmashComputeParameters* computeparams_new(void) { return reinterpret_cast<SourmashComputeParameters*>(new ComputeParameters()); }
void computeparams_free(SourmashComputeParameters* p) { delete CP(p); }
bool computeparams_dayhoff(const SourmashComputeParameters* p) { return CP(p)->dayhoff; }
bool computeparams_dna(const SourmashComputeParameters* p) { return CP(p)->dna; }
bool computeparams_hp(const SourmashComputeParameters* p) { return CP(p)->hp; }
bool computeparams_protein(const SourmashComputeParameters* p) { return CP(p)->protein; }
bool computeparams_track_abundance(const SourmashComputeParameters* p) { return CP(p)->track_abundance; }
const uint32_t* computeparams_ksizes(const SourmashComputeParameters* p, uintptr_t* size) {
    const auto& k = CP(p)->ksizes;
    *size = k.size();
    uint32_t* out = (uint32_t*)malloc((k.size() ? k.size() : 1) * sizeof(uint32_t));
    if (k.size()) memcpy(out, k.data(), k.size() * sizeof(uint32_t));
    return out;
}
void computeparams_ksizes_free(uint32_t* ptr, uintptr_t) { free(ptr); }
uint32_t computeparams_num_hashes(const SourmashComputeParameters* p) { return CP(p)->num_hashes; }
uint64_t computeparams_scaled(const SourmashComputeParameters* p) { return CP(p)->scaled; }
uint64_t computeparams_seed(const SourmashComputeParameters* p) { return CP(p)->seed; }
void computeparams_set_dayhoff(SourmashComputeParameters* p, bool v) { CP(p)->dayhoff = v; }
void computeparams_set_dna(SourmashComputeParameters* p, bool v) { CP(p)->dna = v; }
void computeparams_set_hp(SourmashComputeParameters* p, bool v) { CP(p)->hp = v; }
void computeparams_set_protein(SourmashComputeParameters* p, bool v) { CP(p)->protein = v; }
void computeparams_set_track_abundance(SourmashComputeParameters* p, bool v) { CP(p)->track_abundance = v; }
void computeparams_set_ksizes(SourmashComputeParameters* p, const uint32_t* ks, uintptr_t n) {
    landing_void([&] { CP(p)->ksizes.assign(ks, ks + n); });
}
void computeparams_set_num_hashes(SourmashComputeParameters* p, uint32_t n) { CP(p)->num_hashes = n; }
void computeparams_set_scaled(SourmashComputeParameters* p, uint64_t s) { CP(p)->scaled = s; }
void computeparams_set_seed(SourmashComputeParameters* p, uint64_t s) { CP(p)->seed = s; }

// ---------------------------------------------------------------------------------------------
// sketch object (ffi/minhash.rs)
// ---------------------------------------------------------------------------------------------
SourmashKmerMinHash* kmerminhash_new(uint64_t scaled, uint32_t k, HashFunctions hf, uint64_t seed, bool track,
                                     uint32_t n) {
    return landing<SourmashKmerMinHash*>([&]() -> SourmashKmerMinHash* {
        return reinterpret_cast<SourmashKmerMinHash*>(new KmerMinHash(scaled, k, hf, seed, track, n));
    });
}
void kmerminhash_free(SourmashKmerMinHash* p) {
    KmerMinHash* m = RAW(p);
    if (m) {
        if (DeviceCtx* ctx = DeviceCtx::peek()) {                       // its device mirror goes with it
            if (m->mirrored_gen) {
                std::lock_guard<std::recursive_mutex> g(ctx->mutex());
                ctx->forget(*m);
            }
        }
    }
    delete m;
}
void kmerminhash_slice_free(uint64_t* ptr, uintptr_t) { free(ptr); }

void kmerminhash_add_sequence(SourmashKmerMinHash* p, const char* sequence, bool force) {
    landing_void([&] {
        if (!sequence) throw err_internal("null sequence");
        add_sequence_dna(*RAW(p), (const uint8_t*)sequence, strlen(sequence), force);   // CStr: stops at NUL (ffi/minhash.rs:53-59)
    });
}

const uint64_t* kmerminhash_seq_to_hashes(SourmashKmerMinHash* p, const char* sequence, uintptr_t insize, bool force,
                                          bool bad_kmers_as_zeroes, bool is_protein, uintptr_t* size) {
    return landing<const uint64_t*>([&]() -> const uint64_t* {
        KmerMinHash& mh = *MH(p);
        const uint8_t* seq = (const uint8_t*)sequence;
        std::vector<uint64_t> out;
        if (is_protein || !mh.is_dna()) {
            const uint32_t kr = mh.ksize / 3;
            if (kr == 0 || insize < kr || (!is_protein && insize < (uintptr_t)kr * 3)) return slice_out(out, size);
            if (mh.is_dna()) throw Error(E_INVALID_HASH_FUNCTION, "Invalid hash function: \"DNA\"");
            std::vector<uint64_t> hs;
            {
                DeviceCtx& ctx = DeviceCtx::get();
                std::lock_guard<std::recursive_mutex> g(ctx.mutex());
                ctx.protein_hashes_host(seq, insize, mh.ksize, mh.hash_function, mh.seed, is_protein, hs);
            }
            const bool zeros = force && bad_kmers_as_zeroes;
            // the translate iterator brackets its buffer with two Ok(0) markers (signature.rs:330-348), which only
            // the keep-zeroes mode lets through (ffi/minhash.rs:76-84)
            if (zeros && !is_protein) out.push_back(0);
            for (uint64_t h : hs) if (h != 0 || zeros) out.push_back(h);
            if (zeros && !is_protein) out.push_back(0);
            return slice_out(out, size);
        }
        const uint32_t k = mh.ksize;
        if (insize >= k && k != 0) {
            DeviceCtx& ctx = DeviceCtx::get();
            std::lock_guard<std::recursive_mutex> g(ctx.mutex());
            ctx.kmer_hashes_host(seq, insize, k, mh.seed, out);       // 0 where a k-mer covers an invalid byte
            if (!force) {
                // ffi/minhash.rs:76-96: the first bad k-mer aborts with InvalidDNA
                const size_t pos = ctx.first_invalid_host(seq, insize);
                if (pos != SIZE_MAX) {
                    const size_t bad = pos + 1 >= k ? pos + 1 - k : 0;
                    if (bad < out.size()) throw err_invalid_dna(upper_ascii(seq + bad, k));
                }
            }
            if (!(force && bad_kmers_as_zeroes)) {
                size_t w = 0;
                for (uint64_t h : out) if (h != 0) out[w++] = h;
                out.resize(w);
            }
        }
        return slice_out(out, size);
    });
}

void kmerminhash_add_hash(SourmashKmerMinHash* p, uint64_t h) { RAW(p)->add_hash(h); }   // commutes with the queued records
void kmerminhash_add_hash_with_abundance(SourmashKmerMinHash* p, uint64_t h, uint64_t a) {
    landing_void([&] { MH(p)->add_hash_with_abundance(h, a); });   // abundance 0 removes: settles the queued records first
}
void kmerminhash_add_many(SourmashKmerMinHash* p, const uint64_t* hs, uintptr_t n) {
    landing_void([&] {
        if (!hs && n) throw err_internal("null hashes pointer");
        for (uintptr_t i = 0; i < n; ++i) MH(p)->add_hash(hs[i]);
    });
}
void kmerminhash_add_from(SourmashKmerMinHash* p, const SourmashKmerMinHash* o) {
    landing_void([&] { for (uint64_t h : MH(o)->mins) MH(p)->add_hash(h); });   // minhash.rs:518-523
}
void kmerminhash_add_word(SourmashKmerMinHash* p, const char* word) {           // minhash.rs:401-404
    if (!word) return;
    landing_void([&] { RAW(p)->add_hash(mmh3_h1_bytes((const uint8_t*)word, strlen(word), RAW(p)->seed)); });
}
void kmerminhash_add_protein(SourmashKmerMinHash* p, const char* sequence) {
    landing_void([&] {
        if (!sequence) throw err_internal("null sequence");
        add_residue_kmers(*MH(p), (const uint8_t*)sequence, strlen(sequence), true);
    });
}
void kmerminhash_remove_hash(SourmashKmerMinHash* p, uint64_t h) { landing_void([&] { MH(p)->remove_hash(h); }); }
void kmerminhash_remove_many(SourmashKmerMinHash* p, const uint64_t* hs, uintptr_t n) {
    landing_void([&] {
        if (n < 16) { for (uintptr_t i = 0; i < n; ++i) MH(p)->remove_hash(hs[i]); return; }
        std::vector<uint64_t> sorted(hs, hs + n);
        std::sort(sorted.begin(), sorted.end());
        MH(p)->remove_sorted(sorted.data(), sorted.size());
    });
}
void kmerminhash_remove_from(SourmashKmerMinHash* p, const SourmashKmerMinHash* o) {
    landing_void([&] { MH(p)->remove_sorted(MH(o)->mins.data(), MH(o)->mins.size()); });
}
void kmerminhash_clear(SourmashKmerMinHash* p) { RAW(p)->clear(); }   // drops the queued records too

const uint64_t* kmerminhash_get_mins(const SourmashKmerMinHash* p, uintptr_t* size) {
    return landing<const uint64_t*>([&]() -> const uint64_t* { return slice_out(MH(p)->mins, size); });
}
uintptr_t kmerminhash_get_mins_size(const SourmashKmerMinHash* p) { return MH(p)->size(); }
const uint64_t* kmerminhash_get_abunds(SourmashKmerMinHash* p, uintptr_t* size) {
    return landing<const uint64_t*>([&]() -> const uint64_t* {
        if (!MH(p)->track_abundance) throw Error(E_PANIC, "sourmash panicked: not implemented");   // ffi/minhash.rs:248-260
        return slice_out(MH(p)->abunds, size);
    });
}
void kmerminhash_set_abundances(SourmashKmerMinHash* p, const uint64_t* hs, const uint64_t* as, uintptr_t n, bool clear) {
    landing_void([&] {                                                          // ffi/minhash.rs:269-300
        if ((!hs || !as) && n) throw err_internal("null pointer");
        std::vector<std::pair<uint64_t, uint64_t>> pairs(n);
        for (uintptr_t i = 0; i < n; ++i) pairs[i] = {hs[i], as[i]};
        std::sort(pairs.begin(), pairs.end());
        if (clear) MH(p)->clear();
        for (auto& pr : pairs) MH(p)->add_hash_with_abundance(pr.first, pr.second);
    });
}
SourmashStr kmerminhash_md5sum(const SourmashKmerMinHash* p) {
    return landing<SourmashStr>([&] { return make_str(MH(p)->md5sum()); });
}
void kmerminhash_merge(SourmashKmerMinHash* p, const SourmashKmerMinHash* o) {
    landing_void([&] { MH(p)->merge(*MH(o)); });
}
bool kmerminhash_is_compatible(const SourmashKmerMinHash* p, const SourmashKmerMinHash* o) {
    try { MH(p)->check_compatible(*MH(o)); return true; } catch (const Error&) { return false; }
}

uint64_t kmerminhash_count_common(const SourmashKmerMinHash* p, const SourmashKmerMinHash* o, bool downsample) {
    return landing<uint64_t>([&]() -> uint64_t {                                // minhash.rs:539-558
        Downsampled d;
        align_scaled(*MH(p), *MH(o), downsample, d);
        d.a->check_compatible(*d.b);
        return device_pair(*d.a, *d.b, false, false, 0, nullptr).common;
    });
}

SourmashKmerMinHash* kmerminhash_intersection(const SourmashKmerMinHash* p, const SourmashKmerMinHash* o) {
    return landing<SourmashKmerMinHash*>([&]() -> SourmashKmerMinHash* {        // ffi/minhash.rs:428-441, minhash.rs:560-589
        const KmerMinHash& a = *MH(p);
        const KmerMinHash& b = *MH(o);
        a.check_compatible(b);
        std::vector<uint64_t> list;
        device_pair(a, b, false, true, 0, &list);
        if (a.num != 0) {
            // bottom-k: keep only hashes that survive the merged-and-truncated union (minhash.rs:563-585)
            const uint64_t uni = a.size() + b.size() - list.size();
            if (uni > a.num) {
                // the num-th smallest of the union bounds the kept intersection
                KmerMinHash u = a;
                u.disable_abundance();
                KmerMinHash bb = b; bb.disable_abundance();
                u.merge(bb);
                const uint64_t cutoff = u.mins.empty() ? 0 : u.mins.back();
                while (!list.empty() && list.back() > cutoff) list.pop_back();
            }
        }
        KmerMinHash* out = new KmerMinHash(a);
        out->clear();
        for (uint64_t h : list) out->add_hash(h);
        return reinterpret_cast<SourmashKmerMinHash*>(out);
    });
}

uint64_t kmerminhash_intersection_union_size(const SourmashKmerMinHash* p, const SourmashKmerMinHash* o, uint64_t* usize) {
    return landing<uint64_t>([&]() -> uint64_t {                                // ffi/minhash.rs:443-457
        try {
            const auto cu = intersection_size(*MH(p), *MH(o));
            *usize = cu.second;
            return cu.first;
        } catch (const Error& e) {
            if (e.code >= 101 && e.code <= 104) { *usize = 0; return 0; }      // incompatible -> (0, 0), no error
            throw;
        }
    });
}

double kmerminhash_jaccard(const SourmashKmerMinHash* p, const SourmashKmerMinHash* o) {
    return landing<double>([&] { return jaccard(*MH(p), *MH(o)); });
}

double kmerminhash_similarity(const SourmashKmerMinHash* p, const SourmashKmerMinHash* o, bool ignore_abundance,
                              bool downsample) {
    return landing<double>([&]() -> double {                                    // minhash.rs:682-702
        Downsampled d;
        align_scaled(*MH(p), *MH(o), downsample, d);
        if (ignore_abundance || !d.a->track_abundance || !d.b->track_abundance) return jaccard(*d.a, *d.b);
        return angular(*d.a, *d.b);
    });
}

double kmerminhash_angular_similarity(const SourmashKmerMinHash* p, const SourmashKmerMinHash* o) {
    return landing<double>([&] { return angular(*MH(p), *MH(o)); });
}

uint32_t kmerminhash_num(const SourmashKmerMinHash* p) { return RAW(p)->num; }
uint32_t kmerminhash_ksize(const SourmashKmerMinHash* p) { return RAW(p)->ksize; }
uint64_t kmerminhash_seed(const SourmashKmerMinHash* p) { return RAW(p)->seed; }
uint64_t kmerminhash_max_hash(const SourmashKmerMinHash* p) { return RAW(p)->max_hash; }
HashFunctions kmerminhash_hash_function(const SourmashKmerMinHash* p) { return RAW(p)->hash_function; }
void kmerminhash_hash_function_set(SourmashKmerMinHash* p, HashFunctions hf) {
    landing_void([&] {
        if (hf < 1 || hf > 4) throw Error(E_INVALID_HASH_FUNCTION, "Invalid hash function: \"" + std::to_string(hf) + "\"");
        MH(p)->set_hash_function(hf);
    });
}
bool kmerminhash_is_protein(const SourmashKmerMinHash* p) { return RAW(p)->hash_function == HF_PROTEIN; }
bool kmerminhash_dayhoff(const SourmashKmerMinHash* p) { return RAW(p)->hash_function == HF_DAYHOFF; }
bool kmerminhash_hp(const SourmashKmerMinHash* p) { return RAW(p)->hash_function == HF_HP; }
bool kmerminhash_track_abundance(const SourmashKmerMinHash* p) { return RAW(p)->track_abundance; }
void kmerminhash_enable_abundance(SourmashKmerMinHash* p) { landing_void([&] { MH(p)->enable_abundance(); }); }
void kmerminhash_disable_abundance(SourmashKmerMinHash* p) { MH(p)->disable_abundance(); }

// ---------------------------------------------------------------------------------------------
// signature container (ffi/signature.rs)
// ---------------------------------------------------------------------------------------------
SourmashSignature* signature_new(void) {
    return landing<SourmashSignature*>([&]() -> SourmashSignature* { return reinterpret_cast<SourmashSignature*>(new Signature()); });
}
void signature_free(SourmashSignature* p) { delete SIGRAW(p); }
SourmashSignature* signature_from_params(const SourmashComputeParameters* p) {
    return landing<SourmashSignature*>([&]() -> SourmashSignature* {
        return reinterpret_cast<SourmashSignature*>(new Signature(Signature::from_params(*CP(p))));
    });
}
uintptr_t signature_len(const SourmashSignature* p) { return SIGRAW(const_cast<SourmashSignature*>(p))->sketches.size(); }

void signature_add_sequence(SourmashSignature* p, const char* sequence, bool force) {
    landing_void([&] {                                                          // signature.rs:661-677
        if (!sequence) throw err_internal("null sequence");
        const size_t len = strlen(sequence);
        for (auto& mh : SIGRAW(p)->sketches) add_sequence_dna(mh, (const uint8_t*)sequence, len, force);
    });
}
void signature_add_protein(SourmashSignature* p, const char* sequence) {
    landing_void([&] {                                                          // signature.rs:679-697
        if (!sequence) throw err_internal("null sequence");
        const size_t len = strlen(sequence);
        for (auto& mh : SIG(p)->sketches) add_residue_kmers(mh, (const uint8_t*)sequence, len, true);
    });
}
void signature_set_name(SourmashSignature* p, const char* name) { landing_void([&] { if (name) SIGRAW(p)->name = std::string(name); }); }
void signature_set_filename(SourmashSignature* p, const char* name) { landing_void([&] { if (name) SIGRAW(p)->filename = std::string(name); }); }
SourmashStr signature_get_name(const SourmashSignature* p) {
    return landing<SourmashStr>([&] { return make_str(SIG(p)->name ? *SIG(p)->name : std::string()); });
}
SourmashStr signature_get_filename(const SourmashSignature* p) {
    return landing<SourmashStr>([&] { return make_str(SIG(p)->filename ? *SIG(p)->filename : std::string()); });
}
SourmashStr signature_get_license(const SourmashSignature* p) {
    return landing<SourmashStr>([&] { return make_str(SIG(p)->license); });
}
SourmashKmerMinHash* signature_first_mh(const SourmashSignature* p) {
    return landing<SourmashKmerMinHash*>([&]() -> SourmashKmerMinHash* {        // ffi/signature.rs:167-182: a fresh clone
        if (SIG(p)->sketches.empty()) throw err_internal("found unsupported sketch type");
        return reinterpret_cast<SourmashKmerMinHash*>(new KmerMinHash(SIG(p)->sketches[0]));
    });
}
SourmashKmerMinHash** signature_get_mhs(const SourmashSignature* p, uintptr_t* size) {
    return landing<SourmashKmerMinHash**>([&]() -> SourmashKmerMinHash** {
        const auto& sk = SIG(p)->sketches;
        *size = sk.size();
        SourmashKmerMinHash** out = (SourmashKmerMinHash**)malloc((sk.size() ? sk.size() : 1) * sizeof(void*));
        for (size_t i = 0; i < sk.size(); ++i) out[i] = reinterpret_cast<SourmashKmerMinHash*>(new KmerMinHash(sk[i]));
        return out;
    });
}
void signature_set_mh(SourmashSignature* p, const SourmashKmerMinHash* o) {
    landing_void([&] { SIG(p)->sketches.clear(); SIG(p)->sketches.push_back(*MH(o)); });
}
void signature_push_mh(SourmashSignature* p, const SourmashKmerMinHash* o) {
    landing_void([&] { SIG(p)->sketches.push_back(*MH(o)); });
}
bool signature_eq(const SourmashSignature* p, const SourmashSignature* o) {
    return landing<bool>([&] { return SIG(p)->equals(*SIG(o)); });
}
SourmashStr signature_save_json(const SourmashSignature* p) {
    return landing<SourmashStr>([&] { std::string s; SIG(p)->to_json(s); return make_str(s); });
}

static SourmashSignature** sigs_out(std::vector<Signature>&& sigs, uintptr_t* size) {
    *size = sigs.size();
    SourmashSignature** out = (SourmashSignature**)malloc((sigs.size() ? sigs.size() : 1) * sizeof(void*));
    for (size_t i = 0; i < sigs.size(); ++i) out[i] = reinterpret_cast<SourmashSignature*>(new Signature(std::move(sigs[i])));
    return out;
}

SourmashSignature** signatures_load_buffer(const char* ptr, uintptr_t insize, bool, uintptr_t ksize,
                                           const char* select_moltype, uintptr_t* size) {
    return landing<SourmashSignature**>([&]() -> SourmashSignature** {
        if (!ptr) throw err_internal("null buffer");
        uint32_t mol = 0;
        if (select_moltype) mol = molecule_from_name(select_moltype);
        return sigs_out(load_signatures(ptr, insize, ksize, select_moltype ? &mol : nullptr), size);
    });
}
SourmashSignature** signatures_load_path(const char* path, bool, uintptr_t ksize, const char* select_moltype,
                                         uintptr_t* size) {
    return landing<SourmashSignature**>([&]() -> SourmashSignature** {
        if (!path) throw err_internal("null path");
        std::ifstream in(path, std::ios::binary);
        if (!in) throw Error(E_IO, std::string("No such file or directory: ") + path);
        std::stringstream ss;
        ss << in.rdbuf();
        const std::string data = ss.str();
        uint32_t mol = 0;
        if (select_moltype) mol = molecule_from_name(select_moltype);
        return sigs_out(load_signatures(data.data(), data.size(), ksize, select_moltype ? &mol : nullptr), size);
    });
}
const uint8_t* signatures_save_buffer(const SourmashSignature* const* ptr, uintptr_t n, uint8_t compression,
                                      uintptr_t* osize) {
    return landing<const uint8_t*>([&]() -> const uint8_t* {                    // ffi/signature.rs:220-259
        if (!ptr && n) throw err_internal("null pointer");
        std::string s = "[";
        for (uintptr_t i = 0; i < n; ++i) { if (i) s += ','; SIG(ptr[i])->to_json(s); }
        s += ']';
        if (compression > 0) s = gzip_bytes(s, compression > 9 ? 9 : compression);
        *osize = s.size();
        uint8_t* out = (uint8_t*)malloc(s.size() ? s.size() : 1);
        memcpy(out, s.data(), s.size());
        return out;
    });
}
void nodegraph_buffer_free(uint8_t* ptr, uintptr_t) { free(ptr); }

// =============================================================================================
// PART 2: batch extensions
// =============================================================================================
// host float helper of the compare layer: out[i] = pow(x[i], y[ny == 1 ? 0 : i]) with this process's libm -- the
// function CPython's float ** ends in (floatobject.c float_pow), which is what the reference's containment / ANI
// formulas evaluate (minhash.py:832-834, distance_utils.py:283).  No device involved; threads split the array.
void smgpu_host_pow_f64(const double* x, const double* y, uintptr_t ny, double* out, uintptr_t n, uint32_t n_threads) {
    landing_void([&] {
        if (n == 0) return;
        if (!x || !y || !out) throw err_internal("null pointer");
        size_t t = n_threads ? n_threads : std::thread::hardware_concurrency();
        if (t > 64) t = 64;
        if (t < 1 || n < 65536) t = 1;
        auto work = [&](size_t lo, size_t hi) {
            if (ny == 1) { const double e = y[0]; for (size_t i = lo; i < hi; ++i) out[i] = std::pow(x[i], e); }
            else for (size_t i = lo; i < hi; ++i) out[i] = std::pow(x[i], y[i]);
        };
        if (t == 1) { work(0, n); return; }
        std::vector<std::thread> pool;
        for (size_t k = 0; k < t; ++k) pool.emplace_back(work, n * k / t, n * (k + 1) / t);
        for (auto& th : pool) th.join();
    });
}

uintptr_t smgpu_first_invalid_dna_byte(const char* seq, uintptr_t len) {
    if (!seq || !len) return UINTPTR_MAX;
    const size_t p = first_invalid_byte((const uint8_t*)seq, len);
    return p == SIZE_MAX ? UINTPTR_MAX : (uintptr_t)p;
}

int32_t smgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
bool smgpu_available(void) { return smgpu_device_count() > 0; }

void smgpu_minhash_add_buffer(SourmashKmerMinHash* p, const char* buf, uintptr_t len, bool force) {
    landing_void([&] {
        if (!buf && len) throw err_internal("null buffer");
        if (!MH(p)->is_dna())     // records joined in one buffer would shift the reading frames of the later records
            throw err_internal("smgpu_minhash_add_buffer takes DNA sketches; feed protein sketches record by record");
        add_sequence_dna(*RAW(p), (const uint8_t*)buf, len, force);
    });
}

// add_sequence with the error code as the return value (0 = fine): one call per record for bindings whose call overhead
// matters (ctypes); the message is left in the thread's error slot as usual.  len bytes, no NUL needed.
uint32_t smgpu_minhash_add_sequence_rc(SourmashKmerMinHash* p, const char* sequence, uintptr_t len, bool force) {
    g_err_code = 0;
    landing_void([&] {
        if (!sequence && len) throw err_internal("null sequence");
        // C-string semantics of kmerminhash_add_sequence: a NUL ends the record.  force = false scans the record anyway, and
        // a NUL is an invalid byte like any other: one pass finds whichever comes first.
        KmerMinHash& mh = *RAW(p);
        const uint8_t* seq = (const uint8_t*)sequence;
        if (!force && mh.is_dna()) {
            const size_t bad = first_invalid_byte(seq, len);
            if (bad == SIZE_MAX) { add_sequence_dna(mh, seq, len, false, true, SIZE_MAX); return; }
            if (seq[bad] == 0) { add_sequence_dna(mh, seq, bad, false, true, SIZE_MAX); return; }   // clean up to the NUL
            const void* nul = memchr(seq + bad, 0, len - bad);
            add_sequence_dna(mh, seq, nul ? (size_t)((const uint8_t*)nul - seq) : len, false, true, bad);
            return;
        }
        const void* nul = len ? memchr(sequence, 0, len) : nullptr;
        add_sequence_dna(mh, seq, nul ? (size_t)((const char*)nul - sequence) : len, force);
    });
    return g_err_code;
}
// settle the records queued by add_sequence now (the accessors do it on their own; this is for timing and tests)
void smgpu_minhash_flush(SourmashKmerMinHash* p) { landing_void([&] { settle(*RAW(p)); }); }
uint64_t smgpu_minhash_pending_bytes(const SourmashKmerMinHash* p) { return RAW(p)->pending.size(); }

uint64_t smgpu_signature_add_file(SourmashSignature* p, const char* path, uint64_t* n_records) {
    return landing<uint64_t>([&]() -> uint64_t {
        if (!path) throw err_internal("null path");
        std::vector<KmerMinHash*> mhs;
        for (auto& mh : SIG(p)->sketches) mhs.push_back(&mh);
        uint64_t recs = 0, bases = 0;
        sketch_file_into(mhs, path, &recs, &bases);
        if (n_records) *n_records = recs;
        return bases;
    });
}
SourmashSignature** smgpu_sketch_files(const char* const* paths, uintptr_t n, const SourmashComputeParameters* params,
                                       uint32_t n_threads, uint64_t* total_bases) {
    return landing<SourmashSignature**>([&]() -> SourmashSignature** {
        if (!paths && n) throw err_internal("null paths");
        if (!CP(params)->dna || CP(params)->protein || CP(params)->dayhoff || CP(params)->hp)
            throw err_internal("smgpu_sketch_files takes DNA parameters; protein / dayhoff / hp sketches are fed record by record");
        std::vector<std::string> files(paths, paths + n);
        std::vector<Signature> sigs;
        uint64_t bases = 0;
        sketch_files_parallel(files, *CP(params), n_threads, sigs, &bases);
        if (total_bases) *total_bases = bases;
        uintptr_t size = 0;
        return sigs_out(std::move(sigs), &size);
    });
}
uint64_t smgpu_gunzip_files(const char* const* paths, uintptr_t n, uint8_t* out, uint64_t capacity, uint64_t* lens, double* stats) {
    return landing<uint64_t>([&]() -> uint64_t {
        if ((!paths || !lens) && n) throw err_internal("null paths");
        DeviceCtx& ctx = DeviceCtx::get();
        std::lock_guard<std::recursive_mutex> g(ctx.mutex());
        hipStream_t st = ctx.stream();
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<GunzipMember> ms(n);
        std::vector<std::vector<uint8_t>> blobs(n);
        uint64_t total = 0;
        for (uintptr_t i = 0; i < n; ++i) {
            FILE* f = fopen(paths[i], "rb");
            if (!f) throw Error(E_IO, std::string("No such file or directory: ") + paths[i]);
            fseek(f, 0, SEEK_END);
            const long sz = ftell(f);
            fseek(f, 0, SEEK_SET);
            blobs[i].resize((size_t)std::max(0L, sz));
            const size_t got = sz > 0 ? fread(blobs[i].data(), 1, (size_t)sz, f) : 0;
            fclose(f);
            if (got != (size_t)std::max(0L, sz)) throw Error(E_IO, std::string("short read on ") + paths[i]);
            ms[i].file_off = total;
            ms[i].file_len = (uint64_t)sz;
            total += ((uint64_t)sz + 7) & ~7ull;
        }
        PinnedBuf host;
        host.reserve((size_t)total + GUNZIP_PAD);
        memset(host.p, 0, (size_t)total + GUNZIP_PAD);
        for (uintptr_t i = 0; i < n; ++i) memcpy(host.p + ms[i].file_off, blobs[i].data(), blobs[i].size());
        blobs.clear();
        AsyncBuf dev((size_t)total + GUNZIP_PAD, st);
        hip_check(hipMemcpyAsync(dev.p, host.p, (size_t)total + GUNZIP_PAD, hipMemcpyHostToDevice, st), "H2D");
        hip_check(hipStreamSynchronize(st), "sync");
        const double io_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        void* d_out = nullptr;
        GunzipStats gs;
        gunzip_device(host.p, dev.as<uint8_t>(), total, ms, &d_out, st, &gs);
        struct FreeOut { void* p; hipStream_t st; ~FreeOut() { if (p) arena_free(p, st); } } free_out{d_out, st};
        uint64_t at = 0;
        for (uintptr_t i = 0; i < n; ++i) {
            if (!ms[i].ok) { lens[i] = ~0ull; continue; }
            if (at + ms[i].out_len > capacity) throw err_internal("smgpu_gunzip_files: output capacity too small");
            if (ms[i].out_len)
                hip_check(hipMemcpyAsync(out + at, (const uint8_t*)d_out + ms[i].out_off, (size_t)ms[i].out_len, hipMemcpyDeviceToHost, st), "D2H");
            lens[i] = ms[i].out_len;
            at += ms[i].out_len;
        }
        hip_check(hipStreamSynchronize(st), "sync");
        if (stats) {
            const double v[10] = {(double)gs.survivors, (double)gs.candidates, (double)gs.runs, gs.scan_ms, gs.pass1_ms, gs.link_ms, gs.pass2_ms,
                                  gs.finish_ms, gs.total_ms, io_ms};
            for (int k = 0; k < 16; ++k) stats[k] = k < 10 ? v[k] : 0.0;
        }
        return at;
    });
}
void smgpu_sigload_counters(uint64_t* out) {
    if (!out) return;
    out[0] = sigload_counters().on_device.load();
    out[1] = sigload_counters().on_host.load();
}
void smgpu_gunzip_counters(uint64_t* out) {
    if (!out) return;
    out[0] = gunzip_counters().on_device.load();
    out[1] = gunzip_counters().refused.load();
}
uint64_t smgpu_minhash_add_file(SourmashKmerMinHash* p, const char* path, uint64_t* n_records) {
    return landing<uint64_t>([&]() -> uint64_t {
        if (!path) throw err_internal("null path");
        std::vector<KmerMinHash*> mhs{MH(p)};
        uint64_t recs = 0, bases = 0;
        sketch_file_into(mhs, path, &recs, &bases);
        if (n_records) *n_records = recs;
        return bases;
    });
}

uint64_t smgpu_sketch_workspace_bytes(uint64_t out_capacity) {
    return (uint64_t)out_capacity * 8 + sort_unique_temp_bytes(out_capacity) + 512;
}

uint64_t smgpu_sketch_dna_raw(const uint8_t* d_seq, uint64_t len, uint32_t ksize, uint64_t seed, uint64_t max_hash,
                              uint64_t* d_out, uint64_t cap, uint64_t* d_result, void* d_ws, uint64_t ws_bytes,
                              void* stream) {
    uint64_t ret = ~0ull;
    landing_void([&] {
        hipStream_t st = (hipStream_t)stream;
        if (ws_bytes < smgpu_sketch_workspace_bytes(cap)) throw err_internal("workspace too small (smgpu_sketch_workspace_bytes)");
        uint64_t* d_raw = (uint64_t*)d_ws;                                   // unordered kept hashes
        void* d_tmp = (char*)d_ws + ((cap * 8 + 255) / 256) * 256;
        const size_t tmp_bytes = (size_t)(ws_bytes - ((cap * 8 + 255) / 256) * 256);
        const uint64_t thr = max_hash ? max_hash : ~0ull;
        check_dna_ksize(ksize);
        hip_check(hipMemsetAsync(d_result, 0, 16, st), "memset");
        hip_check(sketch_dna_launch(d_seq, len, ksize, seed, thr, d_raw, (unsigned long long*)d_result, cap, st), "sketch_dna");
        unsigned long long kept = 0;
        hip_check(hipMemcpyAsync(&kept, d_result, 8, hipMemcpyDeviceToHost, st), "D2H");
        hip_check(hipStreamSynchronize(st), "sync");
        if (kept > cap)
            throw err_internal("output capacity too small: " + std::to_string(kept) + " kept hashes > capacity " + std::to_string(cap));
        int bits = 64;
        if (thr != ~0ull) { bits = 1; while (bits < 64 && (thr >> bits)) ++bits; }
        hip_check(sort_unique(d_raw, kept, d_out, nullptr, d_result + 1, d_tmp, tmp_bytes, bits, st), "sort_unique");
        unsigned long long nu = 0;
        hip_check(hipMemcpyAsync(&nu, d_result + 1, 8, hipMemcpyDeviceToHost, st), "D2H");
        hip_check(hipStreamSynchronize(st), "sync");
        ret = nu;
    });
    return ret;
}

uint64_t smgpu_sort_unique_raw(uint64_t* d_keys, uint64_t n, uint64_t* d_out, uint64_t* d_n_out, void* d_ws, uint64_t ws_bytes, void* stream) {
    uint64_t ret = ~0ull;
    landing_void([&] {
        hipStream_t st = (hipStream_t)stream;
        if (ws_bytes < sort_unique_temp_bytes(n)) throw err_internal("workspace too small (smgpu_sketch_workspace_bytes)");
        hip_check(sort_unique(d_keys, n, d_out, nullptr, d_n_out, d_ws, (size_t)ws_bytes, 64, st), "sort_unique");
        unsigned long long nu = 0;
        hip_check(hipMemcpyAsync(&nu, d_n_out, 8, hipMemcpyDeviceToHost, st), "D2H");
        hip_check(hipStreamSynchronize(st), "sync");
        ret = nu;
    });
    return ret;
}

void smgpu_sketch_dna_kernel_raw(const uint8_t* d_seq, uint64_t len, uint32_t ksize, uint64_t seed, uint64_t max_hash,
                                 uint64_t* d_out, uint64_t cap, uint64_t* d_count, void* stream) {
    landing_void([&] {
        check_dna_ksize(ksize);
        hip_check(sketch_dna_launch(d_seq, len, ksize, seed, max_hash ? max_hash : ~0ull, d_out,
                                    (unsigned long long*)d_count, cap, (hipStream_t)stream), "sketch_dna");
    });
}

// The kernels behind protein / dayhoff / hp sketches on device-resident input (benchmarks; the object API goes through
// DeviceCtx::protein_sketch_host): residues of a protein sequence, or the six-frame translation of DNA (signature.rs:307-393), into
// d_aa, then every window of k_aa residues hashed and the hashes 1 <= h <= max_hash appended to d_out.  -> residues written.
uint64_t smgpu_sketch_residues_kernels_raw(const uint8_t* d_seq, uint64_t len, uint32_t k_aa, uint32_t hash_function, uint64_t seed,
                                           uint64_t max_hash, bool translate, uint8_t* d_aa, uint64_t aa_capacity, uint64_t* d_out,
                                           uint64_t cap, uint64_t* d_count, void* stream) {
    return landing<uint64_t>([&]() -> uint64_t {
        if (hash_function < HF_PROTEIN || hash_function > HF_HP) throw err_internal("hash_function must be protein, dayhoff or hp");
        const uint64_t n_aa = translate ? translated_bytes(len) : len;
        // (the window kernel reads d_aa in aligned 8-byte words: the capacity has to cover the last word it touches)
        if (((n_aa + 7) & ~7ull) > aa_capacity) throw err_internal("d_aa is too small: " + std::to_string(n_aa) + " residues need " + std::to_string((n_aa + 7) & ~7ull) + " bytes");
        hipStream_t st = (hipStream_t)stream;
        if (translate) hip_check(translate_launch(d_seq, len, hash_function, d_aa, st), "translate");
        else hip_check(residues_launch(d_seq, len, hash_function, d_aa, st), "residues");
        hip_check(residue_windows_launch(d_aa, n_aa, k_aa, seed, max_hash ? max_hash : ~0ull, d_out, (unsigned long long*)d_count, cap,
                                         false, st), "residue windows");
        return n_aa;
    });
}

void smgpu_synth_dna_raw(uint8_t* d_out, uint64_t start, uint64_t n, uint64_t seed, uint64_t record_len, void* stream) {
    landing_void([&] { hip_check(synth_dna_launch(d_out, start, n, seed, record_len, (hipStream_t)stream), "synth_dna"); });
}

void smgpu_compare_raw(const uint64_t* d_hashes, const uint64_t* d_offsets, uint32_t n, uint32_t row_lo, uint32_t row_hi,
                       uint32_t* d_common, double* d_jaccard, void* stream) {
    landing_void([&] {
        if (!d_common) throw err_internal("d_common is required");
        hip_check(compare_counts_launch(d_hashes, d_offsets, n, row_lo, row_hi, d_common, (hipStream_t)stream), "compare");
        if (d_jaccard)
            hip_check(jaccard_from_counts_launch(d_common, d_offsets, n, row_lo, row_hi, d_jaccard, (hipStream_t)stream), "jaccard");
    });
}

void smgpu_compare_blocks_raw(const uint64_t* d_hashes, const uint64_t* d_offsets, uint32_t n, uint32_t rb_first,
                              uint32_t rb_stride, uint32_t rb_count, uint32_t* d_common, void* stream) {
    landing_void([&] {
        hip_check(compare_blocks_launch(d_hashes, d_offsets, n, rb_first, rb_stride, rb_count, d_common, (hipStream_t)stream),
                  "compare_blocks");
    });
}
void smgpu_symmetrize_raw(uint32_t* d_common, uint32_t n, void* stream) {
    landing_void([&] { hip_check(symmetrize_launch(d_common, n, (hipStream_t)stream), "symmetrize"); });
}
void smgpu_jaccard_raw(const uint32_t* d_common, const uint64_t* d_offsets, uint32_t n, uint32_t row_lo, uint32_t row_hi,
                       double* d_jaccard, void* stream) {
    landing_void([&] {
        hip_check(jaccard_from_counts_launch(d_common, d_offsets, n, row_lo, row_hi, d_jaccard, (hipStream_t)stream), "jaccard");
    });
}

// ---- compare indexes: bit rows over the collection's dictionary, or bit rows for the frequent hashes plus an
//      inverted list of the rare ones (bitindex.hip / sparse_pairs.hip) -----------------------------------------
struct BitIndex {
    uint32_t n = 0, words_per_row = 0;
    uint64_t universe = 0, total = 0;
    uint32_t* bits = nullptr;              // [n][words_per_row]; null when no hash is frequent
    // inverted part (null for the pure bit-row index)
    uint32_t* rows_sorted = nullptr;       // row of every (hash, row) element, ordered by hash
    uint32_t* run_end = nullptr;           // end of the element's run if its hash is rare, else 0
    uint64_t inv_total = 0;                // entries of rows_sorted / run_end (every element after the sort; the rare ones only
                                           // from the sort-free builder)
    uint64_t frequent = 0, rare_pairs = 0;
    uint32_t threshold = 0;
    uint32_t builder = 0;                  // 1: sort-free dictionary builder (dictindex.hip), 2: radix sort of all (hash, row) pairs
    hipStream_t stream = nullptr;          // the arrays come from this stream's pool and go back to it, in order
    ~BitIndex() {
        if (bits) arena_free(bits, stream);
        if (rows_sorted) arena_free(rows_sorted, stream);
        if (run_end) arena_free(run_end, stream);
    }
};

// measured rates behind the cost model (1 x MI355X)
constexpr double RATE_MERGE_STEPS = 6.0e12;   // merge-step equivalents / s of compare_hash_kernel (a pair of sketches = n_i + n_j steps): 5.7e12 at 1,000 x 5,000
                                              // hashes, 7.0e12 at 2,000, 9.5e12 at C4 (profiles/r06_compare_small.jsonl) -- the smaller one decides small problems
constexpr double MERGE_ROUND_FLOOR = 6.0e-8;  // seconds per hash of the mean sketch: the latency floor of the general kernel's rounds
constexpr double RATE_BIT_WORDS = 9.5e12;     // 32-bit AND+popcount / s (bitmatrix_kernel at C4: the VALU roof)
constexpr double RATE_PAIR_ATOMICS = 4.0e9;   // matrix increments / s (rare_pairs_kernel)
// the all-pairs callers compute the triangle and mirror it: a pair costs ONE visit of its tile / ONE increment, like the
// n * total / 2 ... steps the merge rate is calibrated on.  (A bit column costs n^2/2 pairs x 1/32 word, a rare hash held by
// m sketches m^2/2 increments: they meet at m = n * sqrt(RATE_PAIR_ATOMICS / (32 * 2 * RATE_BIT_WORDS)), the threshold below.)
constexpr double TRIANGLE = 0.5;

// Build the cheapest exact index for the collection, or return nullptr (no error) when the merge kernel is.
static BitIndex* bitindex_build(const uint64_t* d_hashes, const uint64_t* d_offsets, uint32_t n, hipStream_t st,
                                uint32_t forced_threshold = 0, bool one_shot = false, uint64_t total = 0) {
    if (n == 0) return nullptr;
    if (total == 0) {                                               // the caller did not say how many hashes there are
        hip_check(hipMemcpyAsync(&total, d_offsets + n, 8, hipMemcpyDeviceToHost, st), "D2H");
        hip_check(hipStreamSynchronize(st), "sync");
    }
    if (total == 0 || total > 0xffffffffull) return nullptr;
    // (the general kernel walks a tile's sketches in lock-step rounds of <= 64 hashes, ~4 us a round however few tiles there are:
    //  sixteen 5,000-hash sketches take 0.32 ms, 256 of them 0.43 -- tools/bench_compare_small.py, profiles/r06_compare_small.jsonl)
    const double t_merge = std::max((double)n * (double)total / RATE_MERGE_STEPS, (double)total / (double)n * MERGE_ROUND_FLOOR);
    // a hash held by m sketches costs m^2 increments as a rare hash, or one bit column (n^2/2 pairs x 1/32 word) as
    // a frequent one: the two meet at m ~ n * sqrt(RATE_PAIR_ATOMICS / (64 * RATE_BIT_WORDS))
    uint32_t threshold = (uint32_t)((double)n * std::sqrt(RATE_PAIR_ATOMICS / (64.0 * RATE_BIT_WORDS)));
    if (threshold < 1) threshold = 1;
    if (forced_threshold) threshold = forced_threshold;
    const double pairs = 0.5 * (double)n * (double)n;
    // The sort-free builder first (dictindex.hip): it serves collections whose distinct hashes fit its per-bucket tables
    // (any collection with heavy sharing: C3 / C4 have 55,000 distinct hashes for 5e6 / 5e7 elements); a bucket that
    // overflows sends the build to the sort below.  SMG_COMPARE_INDEX=sort skips it (tests run both).
    static const bool sort_only = [] { const char* e = getenv("SMG_COMPARE_INDEX"); return e && !strcmp(e, "sort"); }();
    // an index that serves ONE compare must also pay for its own build (~0.1 ms of launches + three passes over the elements)
    // (its slice bounds take 2 KB per sketch: past two million sketches the sort's scratch is the smaller one)
    if (!sort_only && n <= (2u << 20) && !(one_shot && !forced_threshold && t_merge < 0.1e-3 + (double)total / 4.0e10)) {
        std::unique_ptr<BitIndex> bi(new BitIndex());
        bi->n = n; bi->total = total; bi->stream = st;
        AsyncBuf scratch(dict_scratch_bytes(n), st);
        unsigned long long* d_out = reinterpret_cast<unsigned long long*>(scratch.as<char>() + 64);   // inside the zeroed header
        hip_check(hipMemsetAsync(scratch.p, 0, 256, st), "memset");
        hip_check(dict_count_launch(d_hashes, d_offsets, n, threshold, scratch.p, d_out, st), "dictionary pass 1");
        unsigned long long out[5] = {0, 0, 0, 0, 0};
        hip_check(hipMemcpyAsync(out, d_out, sizeof(out), hipMemcpyDeviceToHost, st), "D2H");
        hip_check(hipStreamSynchronize(st), "sync");
        if (out[4] == 0) {
            const uint64_t U = out[0], n_freq = out[1], rare_pairs = out[2], rare_elems = out[3];
            bi->universe = U;
            const uint32_t words = n_freq ? (uint32_t)(((n_freq + 31) / 32 + 31) / 32 * 32) : 0;     // whole 32-word k-steps
            const double t_index = TRIANGLE * (pairs * (double)words / RATE_BIT_WORDS + 2.0 * (double)rare_pairs / RATE_PAIR_ATOMICS) +
                                   (double)total / 2.0e10;
            if ((t_index > t_merge && !forced_threshold) || (double)n * words * 4.0 > 8.0 * (1ull << 30)) return nullptr;
            bi->frequent = n_freq; bi->rare_pairs = rare_pairs; bi->threshold = threshold; bi->words_per_row = words;
            if (words) {
                hip_check(arena_alloc((void**)&bi->bits, (size_t)n * words * 4, st), "arena_alloc");
                hip_check(hipMemsetAsync(bi->bits, 0, (size_t)n * words * 4, st), "memset");
            }
            if (rare_elems) {
                hip_check(arena_alloc((void**)&bi->rows_sorted, rare_elems * 4 + 16, st), "arena_alloc");
                hip_check(arena_alloc((void**)&bi->run_end, rare_elems * 4 + 16, st), "arena_alloc");
                bi->inv_total = rare_elems;
            }
            hip_check(dict_emit_launch(d_hashes, d_offsets, n, scratch.p, bi->bits, words, bi->rows_sorted, bi->run_end, st),
                      "dictionary pass 2");
            bi->builder = 1;
            return bi.release();
        }
    }
    // an index that serves ONE compare must also pay for its own sort: ~0.6 ms of fixed cost + total / 1.4e9 s (round 6: a collection
    // of mostly private hashes, 1,000 / 2,000 sketches of 5,000: 3.8 / 8.2 ms through this builder against 0.88 / 2.8 ms for the general
    // kernel -- the earlier total / 5e9 sent such collections here, tools/bench_compare_small.py)
    if (one_shot && !forced_threshold && t_merge < 0.6e-3 + (double)total / 1.4e9) return nullptr;
    // (hash, row) of the whole collection sorted by hash; runs = distinct hashes with their number of holders.
    // Scratch comes from the stream-ordered pool and the host reads back once: the kernels of a small build take
    // less time than one hipMalloc / hipFree pair or one extra synchronisation.
    std::unique_ptr<BitIndex> bi(new BitIndex());
    bi->n = n; bi->total = total; bi->stream = st;
    const size_t tb = inverted_temp_bytes(total);
    AsyncBuf keys_a(total * 8, st), keys_b(total * 8, st), rows_tmp(total * 4, st), counts((total + 2) * 4, st), tmp(tb, st),
        scal(64, st), flags((total + 2) * 4, st), run_off((total + 2) * 8, st), freq_rank((total + 2) * 8, st);
    hip_check(arena_alloc((void**)&bi->rows_sorted, total * 4 + 16, st), "arena_alloc");
    hip_check(hipMemsetAsync(scal.p, 0, 64, st), "memset");
    uint64_t* d_n_runs = scal.as<uint64_t>();
    unsigned long long* d_out = scal.as<unsigned long long>() + 1;      // [runs, frequent, rare pair increments]
    hip_check(inverted_sort_launch(d_hashes, d_offsets, n, total, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(),
                                   rows_tmp.as<uint32_t>(), bi->rows_sorted, counts.as<uint32_t>(), d_n_runs, tmp.p, tb, st),
              "inverted sort");
    hip_check(inverted_classify_launch(counts.as<uint32_t>(), d_n_runs, total, threshold, flags.as<uint32_t>(),
                                       run_off.as<uint64_t>(), freq_rank.as<uint64_t>(), d_out, tmp.p, tb, st), "classify");
    unsigned long long out[3] = {0, 0, 0};
    hip_check(hipMemcpyAsync(out, d_out, 24, hipMemcpyDeviceToHost, st), "D2H");
    hip_check(hipStreamSynchronize(st), "sync");
    const uint64_t U = out[0], n_freq = out[1], rare_pairs = out[2];
    bi->universe = U;
    bi->inv_total = total;
    const uint32_t words = n_freq ? (uint32_t)(((n_freq + 31) / 32 + 31) / 32 * 32) : 0;     // whole 32-word k-steps
    const double t_index = TRIANGLE * (pairs * (double)words / RATE_BIT_WORDS + 2.0 * (double)rare_pairs / RATE_PAIR_ATOMICS) +
                           (double)total / 2.0e10;                                         // + one pass over the elements
    if ((t_index > t_merge && !forced_threshold) || (double)n * words * 4.0 > 8.0 * (1ull << 30))
        return nullptr;                                             // merge kernel wins / bitmap cap (~BitIndex frees)
    bi->frequent = n_freq; bi->rare_pairs = rare_pairs; bi->threshold = threshold; bi->words_per_row = words;
    if (words) {
        hip_check(arena_alloc((void**)&bi->bits, (size_t)n * words * 4, st), "arena_alloc");
        hip_check(hipMemsetAsync(bi->bits, 0, (size_t)n * words * 4, st), "memset");
    }
    hip_check(arena_alloc((void**)&bi->run_end, total * 4 + 16, st), "arena_alloc");
    hip_check(inverted_apply_launch(run_off.as<uint64_t>(), flags.as<uint32_t>(), freq_rank.as<uint64_t>(), U, total, bi->rows_sorted,
                                    bi->run_end, bi->bits, words, st), "inverted apply");
    if (rare_pairs == 0 && n_freq == U) {             // nothing is rare: plain bit rows, drop the inverted part
        arena_free(bi->rows_sorted, st); arena_free(bi->run_end, st);
        bi->rows_sorted = bi->run_end = nullptr;
    }
    bi->builder = 2;
    return bi.release();                              // stream-ordered: usable by later work on `st` without a sync
}

static void bitindex_compare(const BitIndex* bi, uint32_t rb_first, uint32_t rb_stride, uint32_t rb_count, uint32_t* d_common,
                             hipStream_t st, bool upper_only = false) {
    if (bi->bits)
        hip_check(bitmatrix_launch(bi->bits, bi->words_per_row, bi->n, rb_first, rb_stride, rb_count, d_common, st, upper_only),
                  "bitmatrix");
    else
        hip_check(hipMemsetAsync(d_common, 0, (size_t)rb_count * 16 * bi->n * 4, st), "memset");
    if (bi->run_end)
        hip_check(rare_pairs_launch(bi->rows_sorted, bi->run_end, bi->inv_total, bi->n, rb_first, rb_stride, rb_count, d_common, st,
                                    upper_only),
                  "rare pairs");
}

SmgpuBitIndex* smgpu_bitindex_new(const uint64_t* d_hashes, const uint64_t* d_offsets, uint32_t n, void* stream) {
    return landing<SmgpuBitIndex*>([&]() -> SmgpuBitIndex* {
        return reinterpret_cast<SmgpuBitIndex*>(bitindex_build(d_hashes, d_offsets, n, (hipStream_t)stream));
    });
}
SmgpuBitIndex* smgpu_bitindex_new_ex(const uint64_t* d_hashes, const uint64_t* d_offsets, uint32_t n, uint64_t total_hashes,
                                     uint32_t threshold, bool one_shot, void* stream) {
    return landing<SmgpuBitIndex*>([&]() -> SmgpuBitIndex* {
        return reinterpret_cast<SmgpuBitIndex*>(bitindex_build(d_hashes, d_offsets, n, (hipStream_t)stream, threshold, one_shot,
                                                               total_hashes));
    });
}
void smgpu_bitindex_free(SmgpuBitIndex* p) { delete reinterpret_cast<BitIndex*>(p); }
uint64_t smgpu_bitindex_universe(const SmgpuBitIndex* p) { return reinterpret_cast<const BitIndex*>(p)->universe; }
uint32_t smgpu_bitindex_builder(const SmgpuBitIndex* p) { return reinterpret_cast<const BitIndex*>(p)->builder; }
void smgpu_bitindex_stats(const SmgpuBitIndex* p, uint64_t* frequent_hashes, uint64_t* rare_pairs, uint32_t* threshold) {
    const BitIndex* bi = reinterpret_cast<const BitIndex*>(p);
    *frequent_hashes = bi->frequent ? bi->frequent : (bi->run_end ? 0 : bi->universe);
    *rare_pairs = bi->rare_pairs;
    *threshold = bi->threshold;
}
void smgpu_bitindex_compare_raw(const SmgpuBitIndex* p, uint32_t rb_first, uint32_t rb_stride, uint32_t rb_count,
                                uint32_t* d_common, void* stream) {
    landing_void([&] {
        bitindex_compare(reinterpret_cast<const BitIndex*>(p), rb_first, rb_stride, rb_count, d_common, (hipStream_t)stream);
    });
}
void smgpu_bitindex_compare_upper_raw(const SmgpuBitIndex* p, uint32_t rb_first, uint32_t rb_stride, uint32_t rb_count,
                                      uint32_t* d_common, void* stream) {
    landing_void([&] {
        bitindex_compare(reinterpret_cast<const BitIndex*>(p), rb_first, rb_stride, rb_count, d_common, (hipStream_t)stream, true);
    });
}

// n x n counts of a device-resident CSR into dc (device; ceil(n / 16) * 16 rows): dense collections take the bit-row path
static void compare_counts_device(const uint64_t* d_hashes, const uint64_t* d_offsets, uint64_t n, uint64_t total, AsyncBuf& dc,
                                  hipStream_t st) {
    // result matrices from the arena (kept between calls): hipMalloc / hipFree of a gigabyte per call cost more than the
    // comparison on hosts with a slow driver path
    dc.reset((size_t)((n + 15) / 16 * 16) * n * 4, st);
    std::unique_ptr<BitIndex> bi(bitindex_build(d_hashes, d_offsets, (uint32_t)n, st, 0, true, total));
    if (bi) { // bit rows for the frequent hashes + inverted lists for the rare ones: the triangle, then its mirror image
        bitindex_compare(bi.get(), 0, 1, (uint32_t)((n + 15) / 16), dc.as<uint32_t>(), st, true);
        hip_check(symmetrize_launch(dc.as<uint32_t>(), (uint32_t)n, st), "symmetrize");
    } else    // LDS-tiled merge walk
        hip_check(compare_counts_launch(d_hashes, d_offsets, (uint32_t)n, 0, (uint32_t)n, dc.as<uint32_t>(), st), "compare");
}

// ... (+ Jaccard) into host matrices.  The matrices leave through the pinned ring (hostxfer.hpp): the f64 matrix of config C4
// is 800 MB, and a pageable hipMemcpy of that size was most of the call.
static void compare_device_csr(const uint64_t* d_hashes, const uint64_t* d_offsets, uint64_t n, uint64_t total,
                               uint32_t* common_out, double* jaccard_out, hipStream_t st) {
    AsyncBuf dc;
    compare_counts_device(d_hashes, d_offsets, n, total, dc, st);
    std::unique_ptr<AsyncBuf> dj;
    if (jaccard_out) {
        dj.reset(new AsyncBuf((size_t)n * n * 8, st));
        hip_check(jaccard_from_counts_launch(dc.as<uint32_t>(), d_offsets, (uint32_t)n, 0, (uint32_t)n, dj->as<double>(), st), "jaccard");
    }
    if (common_out) HostXfer::get().device_to_host(common_out, dc.p, (size_t)n * n * 4, st);     // (travels while the Jaccard kernel runs)
    if (jaccard_out) HostXfer::get().device_to_host(jaccard_out, dj->p, (size_t)n * n * 8, st);
    hip_check(hipStreamSynchronize(st), "sync");
}

// the hash vectors (or abundance vectors) of n sketches back to back into d_dst: worker threads pack pinned chunks, ONE H2D copy
// per 32 MiB chunk (round 4: one pageable copy per sketch)
static void upload_rows(const SourmashKmerMinHash* const* mhs, uintptr_t n, const std::vector<uint64_t>& offsets, bool abunds,
                        void* d_dst, hipStream_t st) {
    std::vector<HostPiece> pieces(n);
    for (uintptr_t i = 0; i < n; ++i) {
        const KmerMinHash* m = MH(mhs[i]);
        pieces[i] = HostPiece{abunds ? (const void*)m->abunds.data() : (const void*)m->mins.data(), (size_t)offsets[i] * 8, m->size() * 8};
    }
    HostXfer::get().gather_to_device(d_dst, pieces, (size_t)offsets[n] * 8, st);
}

void smgpu_compare_all_pairs(const SourmashKmerMinHash* const* mhs, uintptr_t n, uint32_t* common_out, double* jaccard_out) {
    landing_void([&] {
        if (n == 0) return;
        if (n > 0xffffffffu) throw err_internal("too many sketches");
        for (uintptr_t i = 1; i < n; ++i) MH(mhs[0])->check_compatible(*MH(mhs[i]));
        if (MH(mhs[0])->num != 0)
            throw err_internal("smgpu_compare_all_pairs handles scaled sketches; use kmerminhash_similarity for num sketches");
        std::vector<uint64_t> offsets(n + 1, 0);
        for (uintptr_t i = 0; i < n; ++i) offsets[i + 1] = offsets[i] + MH(mhs[i])->size();
        const uint64_t total = offsets[n];
        DeviceCtx& ctx = DeviceCtx::get();
        std::lock_guard<std::recursive_mutex> g(ctx.mutex());
        hipStream_t st = ctx.stream();
        AsyncBuf dh(total * 8 + 16, st), doff((n + 1) * 8, st);
        upload_rows(mhs, n, offsets, false, dh.p, st);
        hip_check(hipMemcpyAsync(doff.p, offsets.data(), (n + 1) * 8, hipMemcpyHostToDevice, st), "H2D");
        compare_device_csr(dh.as<uint64_t>(), doff.as<uint64_t>(), n, total, common_out, jaccard_out, st);
    });
}

// A list with SEVERAL scaled values, downsample = True: common_out[i][j] = |A ∩ B| with both sketches downsampled to the pair's
// coarser scaled (compare.py:14-64 -> minhash.rs:682-702, 777-798), the diagonal = the row's size as given.  class_of[i]: index of
// sketch i's own scaled value in the ascending list of distinct values; class_max_hash[c]: max_hash of value c.  sizes_out
// [n_classes][n]: the size of sketch i downsampled to value c (0 where c is finer than the sketch).  One upload of the
// sketches as given, the prefixes of every class gathered on the device; the host builds no downsampled sketch objects.
void smgpu_compare_all_pairs_mixed(const SourmashKmerMinHash* const* mhs, uintptr_t n, const uint32_t* class_of,
                                   const uint64_t* class_max_hash, uintptr_t n_classes, uint32_t* common_out, uint64_t* sizes_out) {
    landing_void([&] {
        if (n == 0 || n_classes == 0) return;
        if (n > 0xffffffffu) throw err_internal("too many sketches");
        if (!class_of || !class_max_hash || !common_out || !sizes_out) throw err_internal("null pointer");
        for (uintptr_t i = 1; i < n; ++i) {                          // as check_compatible with downsample = true: everything but the threshold
            const KmerMinHash *a = MH(mhs[0]), *b = MH(mhs[i]);
            if (a->ksize != b->ksize) throw Error(E_MISMATCH_KSIZES, "different ksizes cannot be compared");
            if (a->hash_function != b->hash_function) throw Error(E_MISMATCH_DNA_PROT, "DNA/prot minhashes cannot be compared");
            if (a->seed != b->seed) throw Error(E_MISMATCH_SEED, "mismatch in seed; comparison fail");
        }
        for (uintptr_t i = 0; i < n; ++i) {
            if (MH(mhs[i])->num != 0 || MH(mhs[i])->max_hash == 0) throw err_internal("smgpu_compare_all_pairs_mixed takes scaled sketches");
            if (class_of[i] >= n_classes) throw err_internal("class index out of range");
        }
        std::vector<uint64_t> offsets(n + 1, 0);
        for (uintptr_t i = 0; i < n; ++i) offsets[i + 1] = offsets[i] + MH(mhs[i])->size();
        const uint64_t total = offsets[n];
        // prefix lengths: hashes <= max_hash of every class at least as coarse as the sketch's own (minhash.rs:777-798)
        for (uintptr_t c = 0; c < n_classes; ++c)
            for (uintptr_t i = 0; i < n; ++i) {
                const KmerMinHash* m = MH(mhs[i]);
                uint64_t len = 0;
                if (class_of[i] == c) len = m->size();
                else if (class_of[i] < c) len = (uint64_t)(std::upper_bound(m->mins.begin(), m->mins.end(), class_max_hash[c]) - m->mins.begin());
                sizes_out[c * n + i] = len;
            }
        DeviceCtx& ctx = DeviceCtx::get();
        std::lock_guard<std::recursive_mutex> g(ctx.mutex());
        hipStream_t st = ctx.stream();
        AsyncBuf dh(total * 8 + 16, st), dcls(n * 4 + 16, st), dout((size_t)n * n * 4 + 16, st);
        upload_rows(mhs, n, offsets, false, dh.p, st);
        hip_check(hipMemcpyAsync(dcls.p, class_of, n * 4, hipMemcpyHostToDevice, st), "H2D");
        hip_check(hipMemsetAsync(dout.p, 0, (size_t)n * n * 4, st), "memset");
        std::vector<uint64_t> src, noff;
        std::vector<uint32_t> rows;
        for (uintptr_t c = 0; c < n_classes; ++c) {
            rows.clear(); src.clear(); noff.assign(1, 0);
            bool any_own = false;
            for (uintptr_t i = 0; i < n; ++i)
                if (class_of[i] <= c) {
                    rows.push_back((uint32_t)i);
                    src.push_back(offsets[i]);
                    noff.push_back(noff.back() + sizes_out[c * n + i]);
                    any_own = any_own || class_of[i] == c;
                }
            const uint32_t m = (uint32_t)rows.size();
            if (!any_own || m == 0) continue;
            AsyncBuf dsrc(m * 8 + 16, st), dnoff((m + 1) * 8 + 16, st), drows(m * 4 + 16, st), dsub_h(noff.back() * 8 + 16, st), dsub;
            hip_check(hipMemcpyAsync(dsrc.p, src.data(), m * 8, hipMemcpyHostToDevice, st), "H2D");
            hip_check(hipMemcpyAsync(dnoff.p, noff.data(), (m + 1) * 8, hipMemcpyHostToDevice, st), "H2D");
            hip_check(hipMemcpyAsync(drows.p, rows.data(), m * 4, hipMemcpyHostToDevice, st), "H2D");
            hip_check(csr_prefix_gather_launch(dh.as<uint64_t>(), dsrc.as<uint64_t>(), dnoff.as<uint64_t>(), m, dsub_h.as<uint64_t>(), st), "gather rows");
            compare_counts_device(dsub_h.as<uint64_t>(), dnoff.as<uint64_t>(), m, noff.back(), dsub, st);
            hip_check(class_scatter_launch(dsub.as<uint32_t>(), m, drows.as<uint32_t>(), dcls.as<uint32_t>(), (uint32_t)c, (uint32_t)n,
                                           dout.as<uint32_t>(), st), "scatter class");
            hip_check(hipStreamSynchronize(st), "sync");             // (the host vectors above are reused by the next class)
        }
        HostXfer::get().device_to_host(common_out, dout.p, (size_t)n * n * 4, st);
        hip_check(hipStreamSynchronize(st), "sync");
    });
}

// The first sketch of each signature WITHOUT the clone signature_first_mh makes (ffi/signature.rs:167-182 clones; a 10,000-signature
// compare cloned 400 MB before it started), and every sketch's parameters in one call instead of a dozen FFI calls per sketch:
// params[i] = {ksize as stored, hash_function, seed, max_hash, num, track_abundance, size, 0}.  The handles are BORROWED: valid while
// the signature lives and is not modified, never to be freed.
static void sketch_params(const KmerMinHash* m, uint64_t* out) {
    out[0] = m->ksize; out[1] = m->hash_function; out[2] = m->seed; out[3] = m->max_hash; out[4] = m->num;
    out[5] = m->track_abundance ? 1 : 0; out[6] = m->size(); out[7] = 0;
}
void smgpu_signatures_sketch_views(const SourmashSignature* const* sigs, uintptr_t n, const SourmashKmerMinHash** out_mhs, uint64_t* params) {
    landing_void([&] {
        for (uintptr_t i = 0; i < n; ++i) {
            if (SIG(sigs[i])->sketches.empty()) throw err_internal("found unsupported sketch type");
            const KmerMinHash* m = &SIG(sigs[i])->sketches[0];
            out_mhs[i] = reinterpret_cast<const SourmashKmerMinHash*>(m);
            sketch_params(MH(out_mhs[i]), params + 8 * i);             // (MH: queued records are hashed first, like every reader)
        }
    });
}
void smgpu_minhashes_params(const SourmashKmerMinHash* const* mhs, uintptr_t n, uint64_t* params) {
    landing_void([&] { for (uintptr_t i = 0; i < n; ++i) sketch_params(MH(mhs[i]), params + 8 * i); });
}

// Page-locked host memory for large results (cached by the library's arena on release): a device-to-host copy into it runs at the
// link's rate with no staging and no page faults -- sourmash_amd/compare.py puts the n x n matrices of large collections there.
void* smgpu_host_alloc(uintptr_t bytes) {
    return landing<void*>([&]() -> void* {
        void* p = nullptr;
        hip_check(arena_pinned_alloc(&p, bytes ? bytes : 1), "pinned host allocation");
        return p;
    });
}
void smgpu_host_free(void* p) { arena_pinned_free(p); }

void smgpu_xfer_roundtrip(const void* const* pieces, const uint64_t* lens, uintptr_t n, void* out, uint64_t out_bytes) {
    landing_void([&] {
        DeviceCtx& ctx = DeviceCtx::get();
        std::lock_guard<std::recursive_mutex> g(ctx.mutex());
        hipStream_t st = ctx.stream();
        std::vector<HostPiece> ps(n);
        size_t total = 0;
        for (uintptr_t i = 0; i < n; ++i) { ps[i] = HostPiece{pieces[i], total, (size_t)lens[i]}; total += (size_t)lens[i]; }
        if (total != out_bytes) throw err_internal("smgpu_xfer_roundtrip: the pieces do not add up to out_bytes");
        AsyncBuf dev(total + 16, st);
        HostXfer::get().gather_to_device(dev.p, ps, total, st);
        HostXfer::get().device_to_host(out, dev.p, total, st);
    });
}
void smgpu_xfer_stats(uint64_t* out5, bool reset) {
    const HostXfer::Stats x = HostXfer::get().stats();
    if (out5) { out5[0] = x.h2d_bytes; out5[1] = x.d2h_bytes; out5[2] = x.h2d_ns; out5[3] = x.d2h_ns; out5[4] = x.calls; }
    if (reset) HostXfer::get().reset_stats();
}

// ---- all pairs of bottom-k / abundance-tracking sketches (compare_ext.hip) ------------------------------------------
// The reference's compare loop calls similarity() per pair (compare.py:36-54); for num sketches that is the merged-and-
// truncated union rule (minhash.rs:593-621), for sketches that track abundance the angular similarity (minhash.rs:635-702).
static void pack_collection(const SourmashKmerMinHash* const* mhs, uintptr_t n, bool with_abund, std::vector<uint64_t>& offsets,
                            AsyncBuf& dh, AsyncBuf& da, AsyncBuf& doff, bool* narrow, hipStream_t st) {
    offsets.assign(n + 1, 0);
    for (uintptr_t i = 0; i < n; ++i) offsets[i + 1] = offsets[i] + MH(mhs[i])->size();
    const uint64_t total = offsets[n];
    dh.reset(total * 8 + 16, st);
    doff.reset((n + 1) * 8, st);
    if (with_abund) da.reset(total * 8 + 16, st);
    bool small = true;
    upload_rows(mhs, n, offsets, false, dh.p, st);
    if (with_abund) {
        upload_rows(mhs, n, offsets, true, da.p, st);
        for (uintptr_t i = 0; i < n && small; ++i)
            for (uint64_t a : MH(mhs[i])->abunds) small = small && a <= 0xffffffffull;
    }
    hip_check(hipMemcpyAsync(doff.p, offsets.data(), (n + 1) * 8, hipMemcpyHostToDevice, st), "H2D");
    if (narrow) *narrow = small;
}

void smgpu_compare_num_all_pairs(const SourmashKmerMinHash* const* mhs, uintptr_t n, uint32_t* common_out, uint32_t* union_out,
                                 double* jaccard_out) {
    landing_void([&] {
        if (n == 0) return;
        if (n > 0xffffffffu) throw err_internal("too many sketches");
        for (uintptr_t i = 1; i < n; ++i) MH(mhs[0])->check_compatible(*MH(mhs[i]));
        std::vector<uint32_t> nums(n);
        for (uintptr_t i = 0; i < n; ++i) {
            nums[i] = MH(mhs[i])->num;
            if (nums[i] == 0) throw err_internal("smgpu_compare_num_all_pairs takes bottom-k (num) sketches; scaled sketches go to smgpu_compare_all_pairs");
        }
        DeviceCtx& ctx = DeviceCtx::get();
        std::lock_guard<std::recursive_mutex> g(ctx.mutex());
        hipStream_t st = ctx.stream();
        std::vector<uint64_t> offsets;
        AsyncBuf dh, da, doff;
        pack_collection(mhs, n, false, offsets, dh, da, doff, nullptr, st);
        AsyncBuf dn(n * 4, st), dc((size_t)n * n * 4, st), du((size_t)n * n * 4, st), dj((size_t)n * n * 8, st);
        hip_check(hipMemcpyAsync(dn.p, nums.data(), n * 4, hipMemcpyHostToDevice, st), "H2D");
        hip_check(hipMemsetAsync(dc.p, 0, (size_t)n * n * 4, st), "memset");
        hip_check(compare_num_launch(dh.as<uint64_t>(), doff.as<uint64_t>(), dn.as<uint32_t>(), (uint32_t)n, dc.as<uint32_t>(),
                                     du.as<uint32_t>(), dj.as<double>(), st), "compare (num)");
        if (common_out) HostXfer::get().device_to_host(common_out, dc.p, (size_t)n * n * 4, st);
        if (union_out) HostXfer::get().device_to_host(union_out, du.p, (size_t)n * n * 4, st);
        if (jaccard_out) HostXfer::get().device_to_host(jaccard_out, dj.p, (size_t)n * n * 8, st);
        hip_check(hipStreamSynchronize(st), "sync");
    });
}

// out[i][j] = 1 - 2 acos(min(prod / (sqrt(sq_i) sqrt(sq_j)), 1)) / pi, 0 when a norm is 0 (minhash.rs:662-679); the host's libm
// like the reference's f64::sqrt / f64::acos; the diagonal is 1.0 (compare.py:33)
void smgpu_host_angular_f64(const uint64_t* prod, const uint64_t* sumsq, uintptr_t n, double* out, uint32_t n_threads) {
    landing_void([&] {
        if (n == 0) return;
        if (!prod || !sumsq || !out) throw err_internal("null pointer");
        size_t t = n_threads ? n_threads : std::thread::hardware_concurrency();
        if (t > 64) t = 64;
        if (t < 1 || n < 256) t = 1;
        std::vector<double> norm(n);
        for (uintptr_t i = 0; i < n; ++i) norm[i] = std::sqrt((double)sumsq[i]);
        auto work = [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i)
                for (size_t j = 0; j < n; ++j) {
                    double v;
                    if (i == j) v = 1.0;
                    else if (norm[i] == 0.0 || norm[j] == 0.0) v = 0.0;
                    else {
                        // the pair is evaluated as similarity(lower index, higher index): norm_a belongs to the lower one
                        const double na = i < j ? norm[i] : norm[j], nb = i < j ? norm[j] : norm[i];
                        double p = (double)prod[i * n + j] / (na * nb);
                        if (p > 1.0) p = 1.0;
                        v = 1.0 - 2.0 * std::acos(p) / 3.14159265358979323846264338327950288;
                    }
                    out[i * n + j] = v;
                }
        };
        if (t == 1) { work(0, n); return; }
        std::vector<std::thread> pool;
        for (size_t k = 0; k < t; ++k) pool.emplace_back(work, n * k / t, n * (k + 1) / t);
        for (auto& th : pool) th.join();
    });
}

void smgpu_compare_angular_all_pairs(const SourmashKmerMinHash* const* mhs, uintptr_t n, double* sims_out, uint64_t* prod_out,
                                     uint64_t* sumsq_out) {
    landing_void([&] {
        if (n == 0) return;
        if (n > 0xffffffffu) throw err_internal("too many sketches");
        for (uintptr_t i = 1; i < n; ++i) MH(mhs[0])->check_compatible(*MH(mhs[i]));
        for (uintptr_t i = 0; i < n; ++i)
            if (!MH(mhs[i])->track_abundance) throw Error(E_NEEDS_ABUNDANCE_TRACKING, "sketch needs abundance for this operation");
        std::vector<uint64_t> prod((size_t)n * n), sq(n);
        {
            DeviceCtx& ctx = DeviceCtx::get();
            std::lock_guard<std::recursive_mutex> g(ctx.mutex());
            hipStream_t st = ctx.stream();
            std::vector<uint64_t> offsets;
            AsyncBuf dh, da, doff;
            bool narrow = true;
            pack_collection(mhs, n, true, offsets, dh, da, doff, &narrow, st);
            AsyncBuf dc((size_t)n * n * 4, st), dp((size_t)n * n * 8, st), ds(n * 8, st);
            hip_check(hipMemsetAsync(dc.p, 0, (size_t)n * n * 4, st), "memset");
            hip_check(hipMemsetAsync(dp.p, 0, (size_t)n * n * 8, st), "memset");
            hip_check(compare_abund_launch(dh.as<uint64_t>(), da.as<uint64_t>(), doff.as<uint64_t>(), (uint32_t)n, narrow,
                                           dc.as<uint32_t>(), dp.as<unsigned long long>(), ds.as<unsigned long long>(), st, offsets.empty() ? 0 : offsets.back()),
                      "compare (abundance)");
            HostXfer::get().device_to_host(prod.data(), dp.p, (size_t)n * n * 8, st);
            hip_check(hipMemcpyAsync(sq.data(), ds.p, n * 8, hipMemcpyDeviceToHost, st), "D2H");
            hip_check(hipStreamSynchronize(st), "sync");
        }
        if (prod_out) memcpy(prod_out, prod.data(), (size_t)n * n * 8);
        if (sumsq_out) memcpy(sumsq_out, sq.data(), n * 8);
        if (sims_out) smgpu_host_angular_f64(prod.data(), sq.data(), n, sims_out, 0);
    });
}

// the same kernels on caller-owned device buffers (torch tensors): benchmarks, and callers that keep collections resident
void smgpu_compare_num_raw(const uint64_t* d_hashes, const uint64_t* d_offsets, const uint32_t* d_nums, uint32_t n, uint32_t* d_common,
                           uint32_t* d_union, double* d_jaccard, void* stream) {
    landing_void([&] { hip_check(compare_num_launch(d_hashes, d_offsets, d_nums, n, d_common, d_union, d_jaccard, (hipStream_t)stream), "compare (num)"); });
}
void smgpu_compare_abund_raw(const uint64_t* d_hashes, const uint64_t* d_abunds, const uint64_t* d_offsets, uint32_t n, bool narrow,
                             uint32_t* d_common, uint64_t* d_prod, uint64_t* d_sumsq, void* stream) {
    landing_void([&] {
        hip_check(compare_abund_launch(d_hashes, d_abunds, d_offsets, n, narrow, d_common, (unsigned long long*)d_prod,
                                       (unsigned long long*)d_sumsq, (hipStream_t)stream), "compare (abundance)");
    });
}

void smgpu_compare_abund_raw_n(const uint64_t* d_hashes, const uint64_t* d_abunds, const uint64_t* d_offsets, uint32_t n, uint64_t total_hashes,
                               bool narrow, uint32_t* d_common, uint64_t* d_prod, uint64_t* d_sumsq, void* stream) {
    landing_void([&] {
        hip_check(compare_abund_launch(d_hashes, d_abunds, d_offsets, n, narrow, d_common, (unsigned long long*)d_prod,
                                       (unsigned long long*)d_sumsq, (hipStream_t)stream, total_hashes), "compare (abundance)");
    });
}

// ---- device-resident sketch collections and gather counters ------------------------------------
struct SketchSet {
    DevBuf hashes, offsets;
    uint64_t n = 0, total = 0;
    // filled by smgpu_sketchset_load (empty for sets built from sketch handles)
    std::vector<uint64_t> host_offsets;
    std::vector<ManifestRow> rows;
    uint32_t ksize = 0, hash_function = HF_DNA;
    uint64_t seed = 42, max_hash = 0, num = 0, skipped = 0;
    // (the device blocks are arena blocks and go back to the arena with the DevBufs: no hipFree, which would synchronise the device)
};
struct GatherCounter {
    const SketchSet* set = nullptr;
    DevBuf q, list, scal;
    uint64_t nq = 0;
    GatherDev g;
    ~GatherCounter() { gather_destroy(g); }          // (q / list / scal: arena blocks, released by their DevBufs)
};
// raw variant: query and database are caller-owned device buffers (torch tensors)
struct GatherRaw {
    GatherDev g;
    ~GatherRaw() { gather_destroy(g); }
};

// run the armed loop to exhaustion; -> number of results (host copies if out_* given).
// Two device loops with identical results: "scan" (default) picks the arg-max over all counters every round; "replay"
// runs the multi-GPU protocol with one shard (the 16 best rows exported, rounds replayed among those candidates while
// the best stays above the best key kept back).  Measured on one GPU (profiles/r02_gather_loops.txt): replay wins only
// when the leading counters are well separated; with clustered counters (config C5: every round lowers all of them by
// a few hashes, so the kept-back key overtakes after ~3 rounds) the extra exchanges make it slower, 69 vs 29 us / round.
static bool gather_use_replay() {
    static const int mode = [] {
        const char* e = getenv("SMG_GATHER_LOOP");
        return e && !strcmp(e, "replay") ? 1 : 0;
    }();
    return mode == 1;
}
// Persistent-loop exit codes that mean "gave up waiting" rather than "broken" (gather.hip: the gate 10, the sweeps 11-13, a peer
// rank's gate 14).  Only the GATE codes are a decision of the grid as a whole: one word decides, nothing has been touched, and
// the rounds can carry on in place from the same state.  11-13 are per-workgroup spin limits inside a round: workgroup X can
// leave at epoch e while workgroup Z, whose sweep began later, still sees the late record and applies round e -- the written-back
// counters and uncovered set then belong to different round counts.  After those the state is void: single-rank callers get the
// error, gather_distributed builds a fresh index and runs the record protocol (parallel.py).
static bool gather_loop_gave_up(unsigned long long code) { return code >= 10 && code <= 14; }
static bool gather_loop_state_untouched(unsigned long long code) { return code == 10 || code == 14; }
static uint64_t gather_drain(GatherDev& g, uint64_t* out_idx, uint64_t* out_isect, uint64_t cap, hipStream_t st) {
    unsigned long long head[GS_SLOTS];
    const bool replay = gather_use_replay() && g.ndb > 0 && g.nq > 0;
    static const bool trace = getenv("SMG_GATHER_TRACE") != nullptr;   // per batch: time to enqueue, time until the GPU is through
    // SMG_GATHER_GRAPH: 0 eager launches only, 1 graph replays only, unset: eager until a batch shows the host cannot keep
    // ahead (the GPU was through almost as soon as the last launch was issued), then graph replays of 64 rounds
    static const int graph_mode = [] { const char* e = getenv("SMG_GATHER_GRAPH"); return e ? atoi(e) : -1; }();
    bool use_graph = graph_mode == 1;
    unsigned batch = 32;
    const auto t_all = std::chrono::steady_clock::now();
    hipEvent_t ev0 = nullptr, ev1 = nullptr;                           // GPU span of the rounds (smgpu_gather_stats)
    struct EvGuard { hipEvent_t &a, &b; ~EvGuard() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } } ev_guard{ev0, ev1};
    hip_check(hipEventCreate(&ev0), "event");
    hip_check(hipEventCreate(&ev1), "event");
    hip_check(hipEventRecord(ev0, st), "event");
    bool graph_synced = false;
    double gpu_ms_graph = 0.0;
    // the whole loop as one resident kernel when the index allows it (gather.hip: gather_loop_kernel)
    bool persistent = false;
    if (!replay && graph_mode != 1) hip_check(gather_run_persistent(g, st, &persistent), "gather loop (persistent)");
    if (persistent) {
        hip_check(hipMemcpyAsync(g.pinned + 16, g.state, sizeof(head), hipMemcpyDeviceToHost, st), "D2H");
        hip_check(hipStreamSynchronize(st), "sync");
        memcpy(head, g.pinned + 16, sizeof(head));
        if (head[GS_ERR]) {
            // The grid did not become resident as a whole (another kernel or another process holds CUs): the kernel gave up at
            // its gate with nothing touched (code 10) and the two-kernel rounds below carry on from exactly that state -- a
            // library inside somebody's process cannot ask for an idle device.  Anything else -- a spin limit inside a round
            // (11-13: not a grid-wide decision, see above), a staged row that never arrived (2-4) -- is an error.
            if (!gather_loop_state_untouched(head[GS_ERR]))
                throw err_internal("gather loop failed (code " + std::to_string(head[GS_ERR]) + ", epoch " + std::to_string(head[13]) +
                                   ", workgroup " + std::to_string(head[14]) + "): the index's counters are void, build it again");
            if (trace)
                fprintf(stderr, "[gather] persistent loop gave up (code %llu, workgroup %llu) after %llu rounds, %.1f us: two-kernel rounds take over\n",
                        head[GS_ERR], head[14], head[GS_ROUNDS], std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_all).count());
            hip_check(hipMemsetAsync(g.state + GS_ERR, 0, 8, st), "memset");
            g.loop_fallbacks++;
            persistent = false;
        } else if (trace)
            fprintf(stderr, "[gather] persistent loop: %llu rounds, %.1f us\n", head[GS_ROUNDS],
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_all).count());
    }
    for (; !persistent;) {
        const auto t0 = std::chrono::steady_clock::now();
        hipStream_t bs = st;                                           // the stream this batch runs on
        if (use_graph && !graph_synced) {                              // the graph runs on the index's own stream: everything
            hip_check(hipStreamSynchronize(st), "sync");               // enqueued on the caller's stream (begin, earlier batches) first
            graph_synced = true;
        }
        if (replay) hip_check(gather_enqueue_replay(g, (batch + GATHER_TOPK_MAX - 1) / GATHER_TOPK_MAX, st), "gather rounds");
        else if (use_graph) hip_check(gather_enqueue_rounds_graph(g, batch, &bs), "gather rounds (graph)");
        else hip_check(gather_enqueue_rounds(g, batch, st), "gather rounds");
        const auto t1 = std::chrono::steady_clock::now();
        hip_check(hipMemcpyAsync(g.pinned + 16, g.state, sizeof(head), hipMemcpyDeviceToHost, bs), "D2H");
        hip_check(hipStreamSynchronize(bs), "sync");
        memcpy(head, g.pinned + 16, sizeof(head));
        if (bs != st) gpu_ms_graph += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        const double enqueue_us = std::chrono::duration<double, std::micro>(t1 - t0).count();
        const double wait_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
        if (trace)
            fprintf(stderr, "[gather] batch of %u rounds (%s): enqueue %.1f us, then %.1f us until done (rounds so far %llu; %.1f us since the loop began)\n",
                    batch, replay ? "replay" : use_graph ? "graph" : "eager", enqueue_us, wait_us, head[GS_ROUNDS],
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_all).count());
        if (head[GS_DONE]) break;
        if (!replay && !use_graph && graph_mode == -1 && batch >= 64 && wait_us < 0.2 * enqueue_us) use_graph = true;
        if (batch < 512) batch *= 2;
    }
    const uint64_t n = head[GS_ROUNDS];
    const uint64_t m = n < cap ? n : cap;
    hip_check(hipEventRecord(ev1, st), "event");
    if (m && out_idx) hip_check(hipMemcpyAsync(out_idx, g.out_idx, m * 8, hipMemcpyDeviceToHost, st), "D2H");
    if (m && out_isect) hip_check(hipMemcpyAsync(out_isect, g.out_isect, m * 8, hipMemcpyDeviceToHost, st), "D2H");
    hip_check(hipStreamSynchronize(st), "sync");
    float span = 0.f;
    (void)hipEventElapsedTime(&span, ev0, ev1);
    g.loop_gpu_ms = span + gpu_ms_graph;                               // graph batches run on another stream: their wall clock
    g.loop_host_ns = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_all).count();
    if (trace)
        fprintf(stderr, "[gather] %llu rounds read back; %.1f us since the loop began\n", (unsigned long long)n,
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_all).count());
    return n;
}

SmgpuSketchSet* smgpu_sketchset_new(const SourmashKmerMinHash* const* mhs, uintptr_t n) {
    return landing<SmgpuSketchSet*>([&]() -> SmgpuSketchSet* {
        for (uintptr_t i = 0; i < n; ++i) (void)MH(mhs[i]);          // queued records are hashed before the context is taken
        DeviceCtx& ctx = DeviceCtx::get();
        std::lock_guard<std::recursive_mutex> g(ctx.mutex());
        std::unique_ptr<SketchSet> s(new SketchSet());
        std::vector<uint64_t> off(n + 1, 0);
        for (uintptr_t i = 0; i < n; ++i) off[i + 1] = off[i] + MH(mhs[i])->size();
        s->n = n; s->total = off[n];
        hipStream_t st = ctx.stream();
        s->hashes.reserve(s->total * 8 + 16, st);
        s->offsets.reserve((n + 1) * 8, st);
        upload_rows(mhs, n, off, false, s->hashes.p, st);               // pinned chunks packed by worker threads, one H2D copy each
        hip_check(hipMemcpyAsync(s->offsets.p, off.data(), (n + 1) * 8, hipMemcpyHostToDevice, st), "H2D");
        hip_check(hipStreamSynchronize(st), "sync");
        return reinterpret_cast<SmgpuSketchSet*>(s.release());
    });
}
void smgpu_sketchset_free(SmgpuSketchSet* p) { delete reinterpret_cast<SketchSet*>(p); }
uintptr_t smgpu_sketchset_len(const SmgpuSketchSet* p) { return reinterpret_cast<const SketchSet*>(p)->n; }

// files -> host CSR (collection.hpp); no device needed
static LoadedCollection load_collection(const char* const* paths, uintptr_t n_paths, uint32_t ksize, const char* moltype,
                                        uint64_t scaled, uint32_t n_threads) {
    LoadSelect sel;
    sel.ksize = ksize;
    sel.hash_function = (moltype && *moltype) ? (int)molecule_from_name(moltype) : -1;
    sel.scaled = scaled;
    CollectionLoader loader(sel, n_threads);
    for (uintptr_t i = 0; i < n_paths; ++i) loader.add_path(paths[i]);
    return loader.run();
}
SmgpuCollection* smgpu_collection_load(const char* const* paths, uintptr_t n_paths, uint32_t ksize, const char* moltype,
                                       uint64_t scaled, uint32_t n_threads) {
    return landing<SmgpuCollection*>([&]() -> SmgpuCollection* {
        return reinterpret_cast<SmgpuCollection*>(new LoadedCollection(load_collection(paths, n_paths, ksize, moltype, scaled, n_threads)));
    });
}
void smgpu_collection_free(SmgpuCollection* p) { delete reinterpret_cast<LoadedCollection*>(p); }
uintptr_t smgpu_collection_len(const SmgpuCollection* p) { return reinterpret_cast<const LoadedCollection*>(p)->rows.size(); }
uint64_t smgpu_collection_total_hashes(const SmgpuCollection* p) { return reinterpret_cast<const LoadedCollection*>(p)->hashes.size(); }
uint64_t smgpu_collection_skipped(const SmgpuCollection* p) { return reinterpret_cast<const LoadedCollection*>(p)->skipped; }
const uint64_t* smgpu_collection_hashes(const SmgpuCollection* p) { return reinterpret_cast<const LoadedCollection*>(p)->hashes.data(); }
const uint64_t* smgpu_collection_offsets(const SmgpuCollection* p) { return reinterpret_cast<const LoadedCollection*>(p)->offsets.data(); }
SourmashStr smgpu_collection_manifest(const SmgpuCollection* p) {
    return landing<SourmashStr>([&]() -> SourmashStr {
        return make_str(manifest_to_csv(reinterpret_cast<const LoadedCollection*>(p)->rows));
    });
}
void smgpu_collection_params(const SmgpuCollection* p, uint32_t* ksize, uint32_t* hash_function, uint64_t* seed,
                             uint64_t* max_hash, uint64_t* num) {
    const LoadedCollection* c = reinterpret_cast<const LoadedCollection*>(p);
    *ksize = c->ksize; *hash_function = c->hash_function; *seed = c->seed; *max_hash = c->max_hash; *num = c->num;
}

static SketchSet* upload_collection(LoadedCollection&& col) {
    DeviceCtx& ctx = DeviceCtx::get();
    std::lock_guard<std::recursive_mutex> g(ctx.mutex());
    hipStream_t st = ctx.stream();
    std::unique_ptr<SketchSet> s(new SketchSet());
    s->n = col.rows.size();
    s->total = col.hashes.size();
    s->hashes.reserve(s->total * 8 + 16, st);
    s->offsets.reserve((s->n + 1) * 8, st);
    if (s->total) hip_check(hipMemcpyAsync(s->hashes.p, col.hashes.data(), s->total * 8, hipMemcpyHostToDevice, st), "H2D");
    hip_check(hipMemcpyAsync(s->offsets.p, col.offsets.data(), (s->n + 1) * 8, hipMemcpyHostToDevice, st), "H2D");
    hip_check(hipStreamSynchronize(st), "sync");
    s->host_offsets = std::move(col.offsets);
    s->rows = std::move(col.rows);
    s->ksize = col.ksize; s->hash_function = col.hash_function; s->seed = col.seed;
    s->max_hash = col.max_hash; s->num = col.num; s->skipped = col.skipped;
    return s.release();
}
// files -> CSR in HBM, no per-sketch objects
SmgpuSketchSet* smgpu_sketchset_load(const char* const* paths, uintptr_t n_paths, uint32_t ksize, const char* moltype,
                                     uint64_t scaled, uint32_t n_threads) {
    return landing<SmgpuSketchSet*>([&]() -> SmgpuSketchSet* {
        DeviceCtx& ctx = DeviceCtx::get();                          // fail before parsing when there is no GPU
        static const bool host_only = [] { const char* e = getenv("SMG_SIGLOAD_DEVICE"); return e && e[0] == '0'; }();
        if (host_only)
            return reinterpret_cast<SmgpuSketchSet*>(upload_collection(load_collection(paths, n_paths, ksize, moltype, scaled, n_threads)));
        // inflate + number parsing on the device, the CSR never leaves HBM (sigload.hpp); what the device does not take is
        // parsed by the host path document by document
        LoadSelect sel;
        sel.ksize = ksize;
        sel.hash_function = (moltype && *moltype) ? (int)molecule_from_name(moltype) : -1;
        sel.scaled = scaled;
        CollectionLoader loader(sel, n_threads);
        for (uintptr_t i = 0; i < n_paths; ++i) loader.add_path(paths[i]);
        std::lock_guard<std::recursive_mutex> g(ctx.mutex());
        hipStream_t st = ctx.stream();
        CollectionLoader::DeviceResult res;
        loader.run_device(st, res);
        std::unique_ptr<SketchSet> s(new SketchSet());
        s->hashes.p = res.d_hashes;
        s->hashes.cap = (size_t)res.total * 8 + 16;
        s->hashes.st = st;
        s->n = res.rows.size();
        s->total = res.total;
        s->offsets.reserve((s->n + 1) * 8, st);
        hip_check(hipMemcpyAsync(s->offsets.p, res.offsets.data(), (s->n + 1) * 8, hipMemcpyHostToDevice, st), "H2D");
        hip_check(hipStreamSynchronize(st), "sync");
        s->host_offsets = std::move(res.offsets);
        s->rows = std::move(res.rows);
        s->ksize = res.ksize; s->hash_function = res.hash_function; s->seed = res.seed;
        s->max_hash = res.max_hash; s->num = res.num; s->skipped = res.skipped;
        sigload_counters().on_device += res.on_device;
        sigload_counters().on_host += res.on_host;
        return reinterpret_cast<SmgpuSketchSet*>(s.release());
    });
}
SmgpuSketchSet* smgpu_sketchset_from_collection(const SmgpuCollection* p) {
    return landing<SmgpuSketchSet*>([&]() -> SmgpuSketchSet* {
        LoadedCollection copy = *reinterpret_cast<const LoadedCollection*>(p);
        return reinterpret_cast<SmgpuSketchSet*>(upload_collection(std::move(copy)));
    });
}
// rows[0..n) of a loaded set as a new set (device-to-device row gather; manifest rows follow)
SmgpuSketchSet* smgpu_sketchset_subset(const SmgpuSketchSet* p, const uint64_t* rows, uintptr_t n) {
    return landing<SmgpuSketchSet*>([&]() -> SmgpuSketchSet* {
        const SketchSet* s = reinterpret_cast<const SketchSet*>(p);
        if (s->host_offsets.size() != s->n + 1) throw err_internal("smgpu_sketchset_subset needs a set made by smgpu_sketchset_load");
        std::unique_ptr<SketchSet> out(new SketchSet());
        out->n = n;
        out->ksize = s->ksize; out->hash_function = s->hash_function; out->seed = s->seed;
        out->max_hash = s->max_hash; out->num = s->num;
        out->host_offsets.assign(n + 1, 0);
        out->rows.reserve(n);
        for (uintptr_t i = 0; i < n; ++i) {
            if (rows[i] >= s->n) throw err_internal("sketch index out of range");
            out->host_offsets[i + 1] = out->host_offsets[i] + (s->host_offsets[rows[i] + 1] - s->host_offsets[rows[i]]);
            if (rows[i] < s->rows.size()) out->rows.push_back(s->rows[rows[i]]);
        }
        out->total = out->host_offsets[n];
        DeviceCtx& ctx = DeviceCtx::get();
        std::lock_guard<std::recursive_mutex> g(ctx.mutex());
        hipStream_t st = ctx.stream();
        out->hashes.reserve((out->total + 1) * 8, st);
        out->offsets.reserve((n + 1) * 8, st);
        DevBuf d_rows;                                              // (an arena block: released on the stream, no device-wide synchronisation)
        d_rows.reserve((n + 1) * 8, st);
        hip_check(hipMemcpyAsync(out->offsets.p, out->host_offsets.data(), (n + 1) * 8, hipMemcpyHostToDevice, st), "H2D");
        if (n) hip_check(hipMemcpyAsync(d_rows.p, rows, n * 8, hipMemcpyHostToDevice, st), "H2D");
        hip_check(copy_rows_launch(s->hashes.as<uint64_t>(), s->offsets.as<uint64_t>(), d_rows.as<uint64_t>(), n,
                                   out->offsets.as<uint64_t>(), out->hashes.as<uint64_t>(), st), "copy_rows");
        hip_check(hipStreamSynchronize(st), "sync");
        return reinterpret_cast<SmgpuSketchSet*>(out.release());
    });
}
uint64_t smgpu_sketchset_total_hashes(const SmgpuSketchSet* p) { return reinterpret_cast<const SketchSet*>(p)->total; }
uint64_t smgpu_sketchset_skipped(const SmgpuSketchSet* p) { return reinterpret_cast<const SketchSet*>(p)->skipped; }
SourmashStr smgpu_sketchset_manifest(const SmgpuSketchSet* p) {
    return landing<SourmashStr>([&]() -> SourmashStr {
        return make_str(manifest_to_csv(reinterpret_cast<const SketchSet*>(p)->rows));
    });
}
void smgpu_sketchset_params(const SmgpuSketchSet* p, uint32_t* ksize, uint32_t* hash_function, uint64_t* seed,
                            uint64_t* max_hash, uint64_t* num) {
    const SketchSet* s = reinterpret_cast<const SketchSet*>(p);
    *ksize = s->ksize; *hash_function = s->hash_function; *seed = s->seed; *max_hash = s->max_hash; *num = s->num;
}
void smgpu_sketchset_sizes(const SmgpuSketchSet* p, uint64_t* out) {
    landing_void([&] {
        const SketchSet* s = reinterpret_cast<const SketchSet*>(p);
        if (s->host_offsets.size() == s->n + 1) {
            for (uint64_t i = 0; i < s->n; ++i) out[i] = s->host_offsets[i + 1] - s->host_offsets[i];
            return;
        }
        std::vector<uint64_t> off(s->n + 1);
        DeviceCtx& ctx = DeviceCtx::get();
        std::lock_guard<std::recursive_mutex> g(ctx.mutex());
        hip_check(hipMemcpyAsync(off.data(), s->offsets.p, (s->n + 1) * 8, hipMemcpyDeviceToHost, ctx.stream()), "D2H");
        hip_check(hipStreamSynchronize(ctx.stream()), "sync");
        for (uint64_t i = 0; i < s->n; ++i) out[i] = off[i + 1] - off[i];
    });
}
// the sketch of row `index` as a new handle (for result rows that need a full object)
SourmashKmerMinHash* smgpu_sketchset_get(const SmgpuSketchSet* p, uint64_t index) {
    return landing<SourmashKmerMinHash*>([&]() -> SourmashKmerMinHash* {
        const SketchSet* s = reinterpret_cast<const SketchSet*>(p);
        if (index >= s->n) throw err_internal("sketch index out of range");
        if (s->host_offsets.size() != s->n + 1) throw err_internal("smgpu_sketchset_get needs a set made by smgpu_sketchset_load");
        const uint64_t lo = s->host_offsets[index], len = s->host_offsets[index + 1] - lo;
        std::unique_ptr<KmerMinHash> mh(new KmerMinHash(0, s->ksize * (s->hash_function == HF_DNA ? 1 : 3), s->hash_function,
                                                        s->seed, false, (uint32_t)s->num));
        mh->max_hash = s->max_hash;
        mh->mins.resize(len);
        mh->touch();
        DeviceCtx& ctx = DeviceCtx::get();
        std::lock_guard<std::recursive_mutex> g(ctx.mutex());
        if (len) hip_check(hipMemcpyAsync(mh->mins.data(), s->hashes.as<uint64_t>() + lo, len * 8, hipMemcpyDeviceToHost, ctx.stream()), "D2H");
        hip_check(hipStreamSynchronize(ctx.stream()), "sync");
        return reinterpret_cast<SourmashKmerMinHash*>(mh.release());
    });
}
void smgpu_sketchset_device_csr(const SmgpuSketchSet* p, const uint64_t** d_hashes, const uint64_t** d_offsets) {
    const SketchSet* s = reinterpret_cast<const SketchSet*>(p);
    *d_hashes = s->hashes.as<uint64_t>();
    *d_offsets = s->offsets.as<uint64_t>();
}
// |query ∩ row| for every row: one streaming pass, no gather index (search / prefetch over a loaded collection)
void smgpu_sketchset_overlaps(const SmgpuSketchSet* p, const SourmashKmerMinHash* query, uint64_t* out) {
    landing_void([&] {
        const SketchSet* s = reinterpret_cast<const SketchSet*>(p);
        if (s->n == 0) return;
        const uint64_t nq = MH(query)->size();
        DeviceCtx& ctx = DeviceCtx::get();
        std::lock_guard<std::recursive_mutex> g(ctx.mutex());
        hipStream_t st = ctx.stream();
        AsyncBuf dq(nq * 8 + 16, st), dc(s->n * 8 + 16, st);
        if (nq) hip_check(hipMemcpyAsync(dq.p, MH(query)->mins.data(), nq * 8, hipMemcpyHostToDevice, st), "H2D");
        hip_check(hipMemsetAsync(dc.p, 0, s->n * 8, st), "memset");
        hip_check(overlap_vector_launch(dq.as<uint64_t>(), nq, s->hashes.as<uint64_t>(), s->offsets.as<uint64_t>(), s->n,
                                        dc.as<unsigned long long>(), 0, st), "overlap");
        hip_check(hipMemcpyAsync(out, dc.p, s->n * 8, hipMemcpyDeviceToHost, st), "D2H");
        hip_check(hipStreamSynchronize(st), "sync");
    });
}
void smgpu_sketchset_compare(const SmgpuSketchSet* p, uint32_t* common_out, double* jaccard_out) {
    landing_void([&] {
        const SketchSet* s = reinterpret_cast<const SketchSet*>(p);
        if (s->n == 0) return;
        if (s->num != 0) throw err_internal("smgpu_sketchset_compare handles scaled sketches");
        DeviceCtx& ctx = DeviceCtx::get();
        std::lock_guard<std::recursive_mutex> g(ctx.mutex());
        compare_device_csr(s->hashes.as<uint64_t>(), s->offsets.as<uint64_t>(), s->n, s->total, common_out, jaccard_out, ctx.stream());
    });
}

SmgpuCounter* smgpu_counter_new(const SmgpuSketchSet* set, const SourmashKmerMinHash* query) {
    return landing<SmgpuCounter*>([&]() -> SmgpuCounter* {
        const SketchSet* s = reinterpret_cast<const SketchSet*>(set);
        (void)MH(query);                                              // queued records are hashed before the context is taken
        DeviceCtx& ctx = DeviceCtx::get();
        std::lock_guard<std::recursive_mutex> g(ctx.mutex());
        hipStream_t st = ctx.stream();
        std::unique_ptr<GatherCounter> c(new GatherCounter());
        c->set = s;
        c->nq = MH(query)->size();
        c->q.reserve(c->nq * 8 + 16, st);
        c->scal.reserve(64, st);
        if (c->nq) hip_check(hipMemcpyAsync(c->q.p, MH(query)->mins.data(), c->nq * 8, hipMemcpyHostToDevice, st), "H2D");
        c->g.Q = c->q.as<uint64_t>();
        c->g.nq = c->nq;
        c->g.hashes = s->hashes.as<uint64_t>();
        c->g.offsets = s->offsets.as<uint64_t>();
        c->g.ndb = s->n;
        c->g.index_base = 0;
        hip_check(gather_build(c->g, st), "gather index");        // postings + counters = |Q ∩ D_d| for every d
        hip_check(gather_begin(c->g, 0, s->n ? s->n : 1, st), "gather arm");
        return reinterpret_cast<SmgpuCounter*>(c.release());
    });
}
void smgpu_counter_free(SmgpuCounter* p) { delete reinterpret_cast<GatherCounter*>(p); }
void smgpu_counter_get(const SmgpuCounter* p, uint64_t* out) {
    landing_void([&] {
        const GatherCounter* c = reinterpret_cast<const GatherCounter*>(p);
        DeviceCtx& ctx = DeviceCtx::get();
        std::lock_guard<std::recursive_mutex> g(ctx.mutex());
        if (c->set->n) hip_check(hipMemcpyAsync(out, c->g.counters, c->set->n * 8, hipMemcpyDeviceToHost, ctx.stream()), "D2H");
        hip_check(hipStreamSynchronize(ctx.stream()), "sync");
    });
}
void smgpu_counter_set(SmgpuCounter* p, uint64_t index, uint64_t value) {
    landing_void([&] {
        GatherCounter* c = reinterpret_cast<GatherCounter*>(p);
        if (index >= c->set->n) throw err_internal("counter index out of range");
        DeviceCtx& ctx = DeviceCtx::get();
        std::lock_guard<std::recursive_mutex> g(ctx.mutex());
        c->g.counters_touched = true;
        hip_check(hipMemcpyAsync(c->g.counters + index, &value, 8, hipMemcpyHostToDevice, ctx.stream()), "H2D");
        hip_check(hipStreamSynchronize(ctx.stream()), "sync");
    });
}
bool smgpu_counter_best(const SmgpuCounter* p, uint64_t* index, uint64_t* count) {
    return landing<bool>([&]() -> bool {
        GatherCounter* c = const_cast<GatherCounter*>(reinterpret_cast<const GatherCounter*>(p));
        DeviceCtx& ctx = DeviceCtx::get();
        std::lock_guard<std::recursive_mutex> g(ctx.mutex());
        hipStream_t st = ctx.stream();
        unsigned long long* d_best = c->scal.as<unsigned long long>();
        hip_check(hipMemsetAsync(d_best, 0, 8, st), "memset");
        hip_check(argmax_launch(c->g.counters, c->set->n, 0, d_best, st), "argmax");
        unsigned long long key = 0;
        hip_check(hipMemcpyAsync(&key, d_best, 8, hipMemcpyDeviceToHost, st), "D2H");
        hip_check(hipStreamSynchronize(st), "sync");
        if (key == 0) return false;
        *count = key >> 32;
        *index = (uint64_t)(0xffffffffull & ~key);
        return true;
    });
}
void smgpu_counter_consume(SmgpuCounter* p, const SourmashKmerMinHash* intersect) {
    landing_void([&] {
        GatherCounter* c = reinterpret_cast<GatherCounter*>(p);
        const uint64_t ni = MH(intersect)->size();
        if (ni == 0) return;
        DeviceCtx& ctx = DeviceCtx::get();
        std::lock_guard<std::recursive_mutex> g(ctx.mutex());
        hipStream_t st = ctx.stream();
        c->list.reserve((ni + 1) * 8 + 16, st);
        uint64_t* d_list = c->list.as<uint64_t>();
        hip_check(hipMemcpyAsync(d_list, &ni, 8, hipMemcpyHostToDevice, st), "H2D");
        hip_check(hipMemcpyAsync(d_list + 1, MH(intersect)->mins.data(), ni * 8, hipMemcpyHostToDevice, st), "H2D");
        // the postings cover hashes of the original query only; an intersect holding anything else (a caller
        // outside the peek/consume protocol) takes the streaming kernel over the whole database instead
        unsigned long long* d_hits = c->scal.as<unsigned long long>() + 1;
        hip_check(hipMemsetAsync(d_hits, 0, 32, st), "memset");
        hip_check(pair_match_launch(d_list + 1, ni, c->g.Q, c->nq, nullptr, nullptr, nullptr, d_hits, 0, st), "pair_match");
        unsigned long long hits = 0;
        hip_check(hipMemcpyAsync(&hits, d_hits, 8, hipMemcpyDeviceToHost, st), "D2H");
        hip_check(hipStreamSynchronize(st), "sync");
        if (hits == ni)
            hip_check(gather_consume_list(c->g, d_list, st), "consume");
        else
            hip_check(overlap_vector_launch(d_list + 1, ni, c->set->hashes.as<uint64_t>(), c->set->offsets.as<uint64_t>(),
                                            c->set->n, c->g.counters, 1, st), "consume");
        hip_check(hipStreamSynchronize(st), "sync");
    });
}
uint64_t smgpu_counter_gather(SmgpuCounter* p, uint64_t threshold_hashes, uint64_t* out_index, uint64_t* out_isect,
                              uint64_t cap) {
    return landing<uint64_t>([&]() -> uint64_t {
        GatherCounter* c = reinterpret_cast<GatherCounter*>(p);
        DeviceCtx& ctx = DeviceCtx::get();
        std::lock_guard<std::recursive_mutex> g(ctx.mutex());
        hipStream_t st = ctx.stream();
        hip_check(gather_begin(c->g, threshold_hashes, c->set->n ? c->set->n : 1, st), "gather arm");
        return gather_drain(c->g, out_index, out_isect, cap, st);
    });
}

// ---- raw gather over caller-owned device buffers ----------------------------------------------------------------
SmgpuGather* smgpu_gather_new_raw(const uint64_t* d_query, uint64_t nq, const uint64_t* d_hashes, const uint64_t* d_offsets,
                                  uint64_t ndb, uint64_t index_base, void* stream) {
    return landing<SmgpuGather*>([&]() -> SmgpuGather* {
        std::unique_ptr<GatherRaw> r(new GatherRaw());
        r->g.Q = d_query;
        r->g.nq = nq;
        r->g.hashes = d_hashes;
        r->g.offsets = d_offsets;
        r->g.ndb = ndb;
        r->g.index_base = index_base;
        hip_check(gather_build(r->g, (hipStream_t)stream), "gather index");
        return reinterpret_cast<SmgpuGather*>(r.release());
    });
}
void smgpu_gather_free(SmgpuGather* p) { delete reinterpret_cast<GatherRaw*>(p); }
uint64_t smgpu_gather_postings(const SmgpuGather* p) { return reinterpret_cast<const GatherRaw*>(p)->g.npairs; }
void smgpu_gather_stats(SmgpuGather* p, double* out) {
    landing_void([&] {
        GatherDev& g = reinterpret_cast<GatherRaw*>(p)->g;
        float ms = 0.f;
        hip_check(gather_build_kernel_ms(g, &ms), "build events");
        out[0] = ms;
        out[1] = g.build_host_ns * 1e-6;
        out[2] = g.build_driver_ns * 1e-6;
        out[3] = (double)g.build_driver_allocs;
        out[4] = (double)g.build_syncs;
        out[5] = g.build_sync_wait_ns * 1e-6;
        out[6] = g.loop_gpu_ms;
        out[7] = g.loop_host_ns * 1e-6;
        out[8] = (double)g.loop_fallbacks;
    });
}
void smgpu_arena_stats(uint64_t* out) {
    const ArenaStats a = arena_stats();
    out[0] = a.driver_allocs; out[1] = a.driver_frees; out[2] = a.driver_ns; out[3] = a.reuse_hits;
    out[4] = a.live_bytes; out[5] = a.cached_bytes; out[6] = a.peak_bytes; out[7] = a.cross_stream_waits;
}
void smgpu_arena_trim(uint64_t keep_bytes) { arena_trim(keep_bytes); }
uint64_t smgpu_gunzip_file(const char* path, uint32_t threads, uint64_t span_bytes, uint8_t* out, uint64_t cap, uint32_t* crc_out,
                           bool* parallel_used) {
    return landing<uint64_t>([&]() -> uint64_t {
        try {
            ParallelGunzip pg(path, threads, span_bytes ? (size_t)span_bytes : ((size_t)1 << 20));
            std::vector<uint8_t> buf((size_t)8 << 20);
            uint64_t total = 0;
            uint32_t crc = 0;
            for (;;) {
                const size_t n = pg.read(buf.data(), buf.size());
                if (n == 0) break;
                if (crc_out) crc = (uint32_t)crc32_z(crc, buf.data(), n);
                if (out && total < cap) memcpy(out + total, buf.data(), (size_t)std::min<uint64_t>(n, cap - total));
                total += n;
            }
            if (crc_out) *crc_out = crc;
            if (parallel_used) *parallel_used = pg.parallel();
            return total;
        } catch (const std::runtime_error& e) {
            throw Error(E_NIFFLER, e.what());
        }
    });
}
uint32_t smgpu_gunzip_position_code(uint32_t p) { return ParallelGunzip::position_code(p & 32767u); }
void smgpu_gather_counters_get(const SmgpuGather* p, uint64_t* out, void* stream) {
    landing_void([&] {
        const GatherDev& g = reinterpret_cast<const GatherRaw*>(p)->g;
        if (g.ndb) hip_check(hipMemcpyAsync(out, g.counters, g.ndb * 8, hipMemcpyDeviceToHost, (hipStream_t)stream), "D2H");
        hip_check(hipStreamSynchronize((hipStream_t)stream), "sync");
    });
}
void smgpu_gather_begin(SmgpuGather* p, uint64_t threshold_hashes, uint64_t max_rounds, void* stream) {
    landing_void([&] {
        hip_check(gather_begin(reinterpret_cast<GatherRaw*>(p)->g, threshold_hashes, max_rounds, (hipStream_t)stream),
                  "gather arm");
    });
}
uint64_t smgpu_gather_run(SmgpuGather* p, uint64_t* out_index, uint64_t* out_isect, uint64_t cap, void* stream) {
    return landing<uint64_t>([&]() -> uint64_t {
        return gather_drain(reinterpret_cast<GatherRaw*>(p)->g, out_index, out_isect, cap, (hipStream_t)stream);
    });
}
// ---- several ranks running the same gather rounds through shared host memory (gather.hip: gather_launch_loop) ----
struct GatherXchg {
    unsigned long long* host = nullptr;      // the mapping in this process
    unsigned long long* dev = nullptr;       // its device-visible address
    size_t bytes = 0;
    uint32_t world = 0;
    uint64_t rowcap = 0;
    std::string shm_name;                    // "" : private pinned memory (one process drives every rank)
    bool creator = false, registered = false;
    // device kind: this rank's area lives in its own device memory, the peers' areas are mapped in through hipIpc handles
    bool device = false;
    uint32_t rank = 0;
    unsigned long long* own = nullptr;
    std::vector<unsigned long long*> peers;  // [world] device-visible address of every rank's area (own included)
    std::vector<void*> opened;               // mappings to close
    ~GatherXchg() {
        if (device) {
            for (void* p : opened) (void)hipIpcCloseMemHandle(p);
            if (own) (void)hipFree(own);     // (not an arena block: IPC handles are per allocation; freed once per process lifetime)
            return;
        }
        if (!host) return;
        if (shm_name.empty()) (void)hipHostFree(host);
        else {
            if (registered) (void)hipHostUnregister(host);
            munmap(host, bytes);
            if (creator) shm_unlink(shm_name.c_str());
        }
    }
};
SmgpuGatherXchg* smgpu_gather_xchg_new(const char* shm_name, uint32_t world, uint64_t rowcap, bool create) {
    return landing<SmgpuGatherXchg*>([&]() -> SmgpuGatherXchg* {
        if (world == 0 || world > 1024 || rowcap == 0) throw err_internal("gather exchange: bad geometry");
        std::unique_ptr<GatherXchg> x(new GatherXchg());
        x->world = world; x->rowcap = rowcap;
        x->bytes = ((size_t)2 * world * 4 + (size_t)2 * world * rowcap) * 8;
        x->bytes = (x->bytes + 4095) & ~(size_t)4095;
        if (!shm_name || !*shm_name) {
            hip_check(hipHostMalloc((void**)&x->host, x->bytes, hipHostMallocDefault), "hipHostMalloc");
            memset(x->host, 0, x->bytes);
            x->dev = x->host;
        } else {
            x->shm_name = shm_name;
            x->creator = create;
            const int fd = shm_open(shm_name, create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
            if (fd < 0) throw err_internal(std::string("gather exchange: shm_open(") + shm_name + ") failed");
            if (create && ftruncate(fd, (off_t)x->bytes) != 0) { ::close(fd); shm_unlink(shm_name); throw err_internal("gather exchange: ftruncate failed"); }
            void* m = mmap(nullptr, x->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            ::close(fd);
            if (m == MAP_FAILED) { if (create) shm_unlink(shm_name); throw err_internal("gather exchange: mmap failed"); }
            x->host = (unsigned long long*)m;
            if (create) memset(x->host, 0, x->bytes);
            hip_check(hipHostRegister(x->host, x->bytes, hipHostRegisterPortable | hipHostRegisterMapped), "hipHostRegister");
            x->registered = true;
            void* d = nullptr;
            hip_check(hipHostGetDevicePointer(&d, x->host, 0), "hipHostGetDevicePointer");
            x->dev = (unsigned long long*)d;
        }
        return reinterpret_cast<SmgpuGatherXchg*>(x.release());
    });
}
// The exchange in DEVICE memory (north_star: "over xGMI"): every rank owns one area -- [2][4] record granules + [2][rowcap] row
// granules -- in its own HBM, fine-grained so that a peer's system-scope loads see the owner's write-through stores, exports it as
// an IPC handle, and maps its peers' areas.  A rank writes only its own area (local stores) and polls the others' (reads over
// xGMI between the GPUs of a node; plain device memory between two processes on one GPU).  The tags of the granules tell runs
// apart, so the area is zeroed once, here.
SmgpuGatherXchg* smgpu_gather_xchg_new_device(uint32_t world, uint32_t rank, uint64_t rowcap) {
    return landing<SmgpuGatherXchg*>([&]() -> SmgpuGatherXchg* {
        if (world == 0 || world > GATHER_PEERS_MAX || rank >= world || rowcap == 0) throw err_internal("gather exchange (device): bad geometry");
        std::unique_ptr<GatherXchg> x(new GatherXchg());
        x->device = true;
        x->world = world; x->rank = rank; x->rowcap = rowcap;
        x->bytes = (((size_t)8 + (size_t)2 * rowcap) * 8 + 4095) & ~(size_t)4095;
        void* p = nullptr;
        hipError_t e = hipExtMallocWithFlags(&p, x->bytes, hipDeviceMallocFinegrained);
        if (e != hipSuccess) { (void)hipGetLastError(); throw err_internal(std::string("gather exchange (device): fine-grained allocation failed: ") + hipGetErrorString(e)); }
        x->own = (unsigned long long*)p;
        hip_check(hipMemset(x->own, 0, x->bytes), "memset");
        hip_check(hipDeviceSynchronize(), "sync");
        x->peers.assign(world, nullptr);
        x->peers[rank] = x->own;
        return reinterpret_cast<SmgpuGatherXchg*>(x.release());
    });
}
uintptr_t smgpu_gather_xchg_ipc_handle_size(void) { return sizeof(hipIpcMemHandle_t); }
void smgpu_gather_xchg_ipc_export(const SmgpuGatherXchg* xp, uint8_t* handle_out) {
    landing_void([&] {
        const GatherXchg* x = reinterpret_cast<const GatherXchg*>(xp);
        if (!x->device) throw err_internal("gather exchange: not a device-memory exchange");
        hipIpcMemHandle_t h;
        hip_check(hipIpcGetMemHandle(&h, x->own), "hipIpcGetMemHandle");
        memcpy(handle_out, &h, sizeof(h));
    });
}
void smgpu_gather_xchg_ipc_open(SmgpuGatherXchg* xp, uint32_t peer_rank, const uint8_t* handle) {
    landing_void([&] {
        GatherXchg* x = reinterpret_cast<GatherXchg*>(xp);
        if (!x->device || peer_rank >= x->world) throw err_internal("gather exchange: bad peer");
        if (peer_rank == x->rank) return;
        hipIpcMemHandle_t h;
        memcpy(&h, handle, sizeof(h));
        void* p = nullptr;
        hip_check(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle");
        x->opened.push_back(p);
        x->peers[peer_rank] = (unsigned long long*)p;
    });
}
bool smgpu_gather_xchg_is_device(const SmgpuGatherXchg* xp) { return reinterpret_cast<const GatherXchg*>(xp)->device; }
void smgpu_gather_xchg_free(SmgpuGatherXchg* p) { delete reinterpret_cast<GatherXchg*>(p); }
bool smgpu_gather_loop_eligible(const SmgpuGather* p, uint32_t n_wg) {
    return gather_loop_eligible(reinterpret_cast<const GatherRaw*>(p)->g, n_wg);
}
void smgpu_gather_loop_reserve(SmgpuGather* p, uint32_t n_wg, uint64_t rowcap, void* stream) {
    landing_void([&] { hip_check(gather_loop_reserve(reinterpret_cast<GatherRaw*>(p)->g, (hipStream_t)stream, n_wg, rowcap), "loop memory"); });
}
// enqueue the armed loop of this rank's shard (nothing is waited for: smgpu_gather_results reads it back)
bool smgpu_gather_launch_shared(SmgpuGather* p, SmgpuGatherXchg* xp, uint32_t rank, uint32_t run_id, uint32_t n_wg, void* stream) {
    return landing<bool>([&]() -> bool {
        GatherDev& g = reinterpret_cast<GatherRaw*>(p)->g;
        GatherXchg* x = reinterpret_cast<GatherXchg*>(xp);
        GatherShared sh;
        sh.rec = x->dev;
        sh.rows = x->dev ? x->dev + (size_t)2 * x->world * 4 : nullptr;
        sh.W = x->world; sh.rank = rank; sh.rowcap = x->rowcap; sh.run_id = run_id;
        if (x->device) {
            if (rank != x->rank) throw err_internal("gather exchange (device): this area belongs to another rank");
            for (uint32_t r = 0; r < x->world; ++r)
                if (!x->peers[r]) throw err_internal("gather exchange (device): the area of rank " + std::to_string(r) + " was never opened");
            sh.peers = x->peers.data();
        }
        if (g.longest_row > x->rowcap) throw err_internal("gather exchange: a row of this shard is longer than the exchange's slots");
        bool ran = false;
        hip_check(gather_launch_loop(g, (hipStream_t)stream, n_wg, &sh, &ran), "gather loop (shared)");
        return ran;
    });
}
void smgpu_debug_hold_cus(uint32_t n_wg, uint32_t lds_bytes, uint64_t micros, void* stream) {
    landing_void([&] { hip_check(debug_hold_cus(n_wg, lds_bytes, micros, (hipStream_t)stream), "hold CUs"); });
}
uint64_t smgpu_gather_longest_row(const SmgpuGather* p) { return reinterpret_cast<const GatherRaw*>(p)->g.longest_row; }
void smgpu_gather_topk_export_raw(SmgpuGather* p, uint64_t* d_records, uint32_t k, uint64_t stride, void* stream) {
    landing_void([&] {
        hip_check(gather_topk_export(reinterpret_cast<GatherRaw*>(p)->g, d_records, k, stride, (hipStream_t)stream), "top-k export");
    });
}
void smgpu_gather_cands_load_raw(SmgpuGather* p, const uint64_t* d_records, uint32_t n_records, uint64_t stride, void* stream) {
    landing_void([&] {
        hip_check(gather_cands_load(reinterpret_cast<GatherRaw*>(p)->g, d_records, n_records, stride, (hipStream_t)stream),
                  "candidate load");
    });
}
void smgpu_gather_replay_raw(SmgpuGather* p, uint32_t rounds, void* stream) {
    landing_void([&] {
        hip_check(gather_replay_rounds(reinterpret_cast<GatherRaw*>(p)->g, rounds, (hipStream_t)stream), "replay");
    });
}
uint64_t smgpu_gather_poll(SmgpuGather* p, bool* done, void* stream) {
    return landing<uint64_t>([&]() -> uint64_t {
        GatherDev& g = reinterpret_cast<GatherRaw*>(p)->g;
        unsigned long long head[GS_SLOTS];
        hip_check(hipMemcpyAsync(head, g.state, sizeof(head), hipMemcpyDeviceToHost, (hipStream_t)stream), "D2H");
        hip_check(hipStreamSynchronize((hipStream_t)stream), "sync");
        if (done) *done = head[GS_DONE] != 0;
        return head[GS_ROUNDS];
    });
}
uint64_t smgpu_gather_results(SmgpuGather* p, uint64_t* out_index, uint64_t* out_isect, uint64_t cap, void* stream) {
    return landing<uint64_t>([&]() -> uint64_t {
        GatherDev& g = reinterpret_cast<GatherRaw*>(p)->g;
        hipStream_t st = (hipStream_t)stream;
        unsigned long long head[GS_SLOTS];
        hip_check(hipMemcpyAsync(head, g.state, sizeof(head), hipMemcpyDeviceToHost, st), "D2H");
        hip_check(hipStreamSynchronize(st), "sync");
        if (head[GS_ERR]) {
            // codes 10-14: the loop gave up between two rounds (at its gate, or waiting for a workgroup or a rank) and the index is
            // intact -- the caller agrees with its peers on what to do next (parallel.gather_distributed: the record protocol)
            const unsigned long long code = head[GS_ERR];
            if (gather_loop_gave_up(code)) { hip_check(hipMemsetAsync(g.state + GS_ERR, 0, 8, st), "memset"); g.loop_fallbacks++; }
            throw err_internal("gather loop: a workgroup or rank waited too long for its peers (code " + std::to_string(code) + ", epoch " +
                               std::to_string(head[13]) + ", workgroup " + std::to_string(head[14]) + ")");
        }
        const uint64_t n = head[GS_ROUNDS], m = n < cap ? n : cap;
        if (m) {
            hip_check(hipMemcpyAsync(out_index, g.out_idx, m * 8, hipMemcpyDeviceToHost, st), "D2H");
            hip_check(hipMemcpyAsync(out_isect, g.out_isect, m * 8, hipMemcpyDeviceToHost, st), "D2H");
            hip_check(hipStreamSynchronize(st), "sync");
        }
        return n;
    });
}

void smgpu_overlap_raw(const uint64_t* d_query, uint64_t nq, const uint64_t* d_hashes, const uint64_t* d_offsets,
                       uint64_t ndb, uint64_t* d_overlap, int32_t op, void* stream) {
    landing_void([&] {
        hip_check(overlap_vector_launch(d_query, nq, d_hashes, d_offsets, ndb, (unsigned long long*)d_overlap, op,
                                        (hipStream_t)stream), "overlap");
    });
}
void smgpu_argmax_raw(const uint64_t* d_overlap, uint64_t ndb, uint64_t index_base, uint64_t* d_best, void* stream) {
    landing_void([&] {
        hip_check(argmax_launch((const unsigned long long*)d_overlap, ndb, index_base, (unsigned long long*)d_best,
                                (hipStream_t)stream), "argmax");
    });
}
uint64_t smgpu_intersect_workspace_bytes(uint64_t n) { return (uint64_t)select_temp_bytes(n) + ((n + 255) / 256) * 256 + 256; }

static void select_pair(const uint64_t* d_a, uint64_t na, const uint64_t* d_b, uint64_t nb, uint64_t* d_out, uint64_t* d_n,
                        void* d_ws, uint64_t ws_bytes, int invert, void* stream) {
    landing_void([&] {
        if (ws_bytes < smgpu_intersect_workspace_bytes(na)) throw err_internal("workspace too small (smgpu_intersect_workspace_bytes)");
        uint8_t* flags = (uint8_t*)d_ws;
        void* tmp = (char*)d_ws + ((na + 255) / 256) * 256;
        const size_t tmp_bytes = (size_t)(ws_bytes - ((na + 255) / 256) * 256);
        hipStream_t st = (hipStream_t)stream;
        hip_check(pair_match_launch(d_a, na, d_b, nb, nullptr, nullptr, flags, nullptr, invert, st), "pair_match");
        hip_check(select_flagged(d_a, flags, na, d_out, d_n, tmp, tmp_bytes, st), "select");
    });
}
void smgpu_intersect_raw(const uint64_t* d_a, uint64_t na, const uint64_t* d_b, uint64_t nb, uint64_t* d_out, uint64_t* d_n,
                         void* d_ws, uint64_t ws_bytes, void* stream) {
    select_pair(d_a, na, d_b, nb, d_out, d_n, d_ws, ws_bytes, 0, stream);
}
void smgpu_subtract_raw(const uint64_t* d_a, uint64_t na, const uint64_t* d_b, uint64_t nb, uint64_t* d_out, uint64_t* d_n,
                        void* d_ws, uint64_t ws_bytes, void* stream) {
    select_pair(d_a, na, d_b, nb, d_out, d_n, d_ws, ws_bytes, 1, stream);
}

}  // extern "C"
