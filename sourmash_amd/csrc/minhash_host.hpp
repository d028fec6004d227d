// Host-side sketch object behind the opaque `SourmashKmerMinHash*` handle.
//
// Mirrors the observable behaviour of src/core/src/sketch/minhash.rs:36-913
// (KmerMinHash, the Vec-backed sketch the C-ABI exposes): a sorted unique
// vector of u64 hashes with optional per-hash abundances.  Container
// bookkeeping (insert / remove / merge / downsample / md5) is host code; every
// operation that walks k-mers or intersects two sketches (add_sequence,
// seq_to_hashes, count_common, intersection, jaccard, similarity,
// angular_similarity) is executed by the HIP kernels through DeviceCtx
// (device_ctx.hpp) and fails with an Internal error if no GPU is present.
#pragma once
#include <stdint.h>
#include <algorithm>
#include <atomic>
#include <string>
#include <vector>
#include "md5.hpp"
#include "smg_errors.hpp"

namespace smg {

enum HashFn : uint32_t { HF_DNA = 1, HF_PROTEIN = 2, HF_DAYHOFF = 3, HF_HP = 4 };   // include/sourmash.h:11-17

// src/core/src/sketch/minhash.rs:21-27 (u64::MAX as f64 == 2^64; truncating cast)
inline uint64_t max_hash_for_scaled(uint64_t scaled) {
    if (scaled == 0) return 0;
    if (scaled == 1) return UINT64_MAX;
    return (uint64_t)(18446744073709551616.0 / (double)scaled);
}
// src/core/src/sketch/minhash.rs:29-34 (Rust float->int casts saturate)
inline uint64_t scaled_for_max_hash(uint64_t max_hash) {
    if (max_hash == 0) return 0;
    const double q = 18446744073709551616.0 / (double)max_hash;
    return q >= 18446744073709551616.0 ? UINT64_MAX : (uint64_t)q;
}

struct KmerMinHash {
    uint32_t num = 0;
    uint32_t ksize = 0;
    uint32_t hash_function = HF_DNA;
    uint64_t seed = 42;
    uint64_t max_hash = 0;
    bool track_abundance = false;
    std::vector<uint64_t> mins;
    std::vector<uint64_t> abunds;   // parallel to mins iff track_abundance
    // DNA records handed to add_sequence and not hashed yet (each followed by a '\n', which no k-mer may span): the C-ABI
    // layer runs them through the sketch kernel in one launch when the state is next looked at (capi.cpp: settle)
    mutable std::string pending;
    // Content generation: a process-wide unique number taken at construction and again by every mutator.  Two objects with
    // the same generation hold the same hashes (a copy keeps it until either side changes), which is what lets the device
    // keep a mirror of a sketch's hashes between per-pair calls (device_ctx.hpp: mirror_of) and drop it the moment the
    // sketch changes.  Code that writes mins / abunds directly calls touch().
    uint64_t gen = next_gen();
    mutable uint64_t mirrored_gen = 0;   // the generation this object last had a device mirror made for (0: none): lets the device
                                         // drop the mirror of the content this object no longer holds instead of waiting for LRU
    static uint64_t next_gen() {
        static std::atomic<uint64_t> g{1};
        return g.fetch_add(1, std::memory_order_relaxed);
    }
    void touch() { gen = next_gen(); }

    KmerMinHash() = default;
    // a copy shares the content generation (same hashes -> same device mirror) but not the bookkeeping of who made that mirror:
    // freeing or changing the copy must not take the original's mirror away
    KmerMinHash(const KmerMinHash& o)
        : num(o.num), ksize(o.ksize), hash_function(o.hash_function), seed(o.seed), max_hash(o.max_hash),
          track_abundance(o.track_abundance), mins(o.mins), abunds(o.abunds), pending(o.pending), gen(o.gen), mirrored_gen(0) {}
    KmerMinHash& operator=(const KmerMinHash& o) {
        if (this != &o) {
            num = o.num; ksize = o.ksize; hash_function = o.hash_function; seed = o.seed; max_hash = o.max_hash;
            track_abundance = o.track_abundance; mins = o.mins; abunds = o.abunds; pending = o.pending; gen = o.gen;
        }                                   // (mirrored_gen stays: the next device use drops the mirror of the old content)
        return *this;
    }
    KmerMinHash(KmerMinHash&&) = default;
    KmerMinHash& operator=(KmerMinHash&&) = default;
    // minhash.rs:186-221
    KmerMinHash(uint64_t scaled, uint32_t k, uint32_t hf, uint64_t seed_, bool track, uint32_t n)
        : num(n), ksize(k), hash_function(hf), seed(seed_), max_hash(max_hash_for_scaled(scaled)),
          track_abundance(track) {
        mins.reserve(n > 0 ? (n < (1u << 20) ? n : (1u << 20)) : 1000);   // a caller-chosen n must not decide an allocation
    }

    uint64_t scaled() const { return scaled_for_max_hash(max_hash); }
    size_t size() const { return mins.size(); }
    bool is_dna() const { return hash_function == HF_DNA; }

    void clear() { touch(); mins.clear(); abunds.clear(); pending.clear(); }   // minhash.rs:239-244

    // minhash.rs:406-416
    void remove_hash(uint64_t h) {
        touch();
        auto it = std::lower_bound(mins.begin(), mins.end(), h);
        if (it != mins.end() && *it == h) {
            const size_t pos = (size_t)(it - mins.begin());
            mins.erase(it);
            if (track_abundance) abunds.erase(abunds.begin() + (long)pos);
        }
    }

    // minhash.rs:418-430 remove_many / remove_from: same result as remove_hash per element, done as
    // one sorted set difference (the reference's Vec::remove per hash is quadratic on big queries).
    void remove_sorted(const uint64_t* hs, size_t n) {
        touch();
        size_t i = 0, j = 0, w = 0;
        while (i < mins.size()) {
            while (j < n && hs[j] < mins[i]) ++j;
            if (j < n && hs[j] == mins[i]) { ++i; continue; }
            mins[w] = mins[i];
            if (track_abundance) abunds[w] = abunds[i];
            ++w; ++i;
        }
        mins.resize(w);
        if (track_abundance) abunds.resize(w);
    }

    // minhash.rs:313-383
    void add_hash_with_abundance(uint64_t h, uint64_t abundance) {
        touch();
        const uint64_t current_max = mins.empty() ? UINT64_MAX : mins.back();
        if (h > max_hash && max_hash != 0) return;           // keep rule is inclusive (:319)
        if (num == 0 && max_hash == 0) return;
        if (abundance == 0) { remove_hash(h); return; }
        if (mins.empty()) {
            mins.push_back(h);
            if (track_abundance) abunds.push_back(abundance);
            return;
        }
        if (h <= max_hash || h <= current_max || mins.size() < (size_t)num) {
            auto it = std::lower_bound(mins.begin(), mins.end(), h);
            const size_t pos = (size_t)(it - mins.begin());
            if (it == mins.end()) {
                mins.push_back(h);
                if (track_abundance) abunds.push_back(abundance);
            } else if (*it != h) {
                mins.insert(it, h);
                if (track_abundance) abunds.insert(abunds.begin() + (long)pos, abundance);
                if (num != 0 && mins.size() > (size_t)num) {
                    mins.pop_back();
                    if (track_abundance) abunds.pop_back();
                }
            } else if (track_abundance) {
                abunds[pos] += abundance;
            }
        }
    }
    void add_hash(uint64_t h) { add_hash_with_abundance(h, 1); }

    // Bulk insert of an already sorted, unique batch (what the sketch kernels
    // return) with optional multiplicities: same result as calling
    // add_hash_with_abundance per element, in one linear merge.
    void add_sorted_batch(const uint64_t* hs, const uint64_t* counts, size_t n) {
        if (n == 0) return;
        touch();
        if (num == 0 && max_hash == 0) return;
        if (mins.empty() && num == 0) {                        // first batch of a scaled sketch: it IS the sketch
            const size_t keep = (size_t)(std::upper_bound(hs, hs + n, max_hash) - hs);
            mins.assign(hs, hs + keep);
            if (track_abundance) {
                if (counts) abunds.assign(counts, counts + keep);
                else abunds.assign(keep, 1);
            }
            return;
        }
        std::vector<uint64_t> mm, ma;
        mm.reserve(mins.size() + n);
        if (track_abundance) ma.reserve(mins.size() + n);
        size_t i = 0, j = 0;
        while (i < mins.size() || j < n) {
            if (j < n && max_hash != 0 && hs[j] > max_hash) { j = n; continue; }   // sorted: the rest is out of range too
            uint64_t v, a;
            if (j >= n || (i < mins.size() && mins[i] < hs[j])) { v = mins[i]; a = track_abundance ? abunds[i] : 1; ++i; }
            else if (i >= mins.size() || hs[j] < mins[i]) { v = hs[j]; a = counts ? counts[j] : 1; ++j; }
            else { v = mins[i]; a = (track_abundance ? abunds[i] : 1) + (counts ? counts[j] : 1); ++i; ++j; }
            mm.push_back(v);
            if (track_abundance) ma.push_back(a);
            if (num != 0 && mm.size() == (size_t)num) {
                // bottom-k sketch is full: later (larger) hashes can only bump abundances of kept ones -- none left
                break;
            }
        }
        mins.swap(mm);
        if (track_abundance) abunds.swap(ma);
    }

    // minhash.rs:886-912 -- order: ksize, hash_function, max_hash, seed (num is not compared)
    void check_compatible(const KmerMinHash& o) const {
        if (ksize != o.ksize) throw err_mismatch_ksizes();
        if (hash_function != o.hash_function) throw err_mismatch_dnaprot();
        if (max_hash != o.max_hash) throw err_mismatch_scaled();
        if (seed != o.seed) throw err_mismatch_seed();
    }

    // minhash.rs:432-516
    void merge(const KmerMinHash& o) {
        check_compatible(o);
        touch();
        const bool both = track_abundance && o.track_abundance;
        std::vector<uint64_t> mm, ma;
        mm.reserve(mins.size() + o.mins.size());
        if (both) ma.reserve(mins.size() + o.mins.size());
        size_t i = 0, j = 0;
        while (i < mins.size() && j < o.mins.size()) {
            if (mins[i] < o.mins[j]) { mm.push_back(mins[i]); if (both) ma.push_back(abunds[i]); ++i; }
            else if (o.mins[j] < mins[i]) { mm.push_back(o.mins[j]); if (both) ma.push_back(o.abunds[j]); ++j; }
            else { mm.push_back(mins[i]); if (both) ma.push_back(abunds[i] + o.abunds[j]); ++i; ++j; }
        }
        for (; i < mins.size(); ++i) { mm.push_back(mins[i]); if (both) ma.push_back(abunds[i]); }
        for (; j < o.mins.size(); ++j) { mm.push_back(o.mins[j]); if (both) ma.push_back(o.abunds[j]); }
        if (num != 0 && mm.size() > (size_t)num) { mm.resize(num); if (both) ma.resize(num); }
        mins.swap(mm);
        abunds.swap(ma);
        track_abundance = both;     // merged abundances exist only if both sides track (:437-442)
    }

    // minhash.rs:777-798: re-add everything under the coarser max_hash (a prefix of the sorted vector)
    KmerMinHash downsample_scaled(uint64_t new_scaled) const {
        const uint64_t cur = scaled();
        if (cur == new_scaled || cur == 0) return *this;
        if (cur > new_scaled) throw err_cannot_upsample();
        KmerMinHash out(new_scaled, ksize, hash_function, seed, track_abundance, num);   // (a fresh generation)
        const size_t keep = (size_t)(std::upper_bound(mins.begin(), mins.end(), out.max_hash) - mins.begin());
        out.mins.assign(mins.begin(), mins.begin() + (long)keep);
        if (track_abundance) out.abunds.assign(abunds.begin(), abunds.begin() + (long)keep);
        return out;
    }

    // minhash.rs:290-307
    std::string md5sum() const {
        Md5 m;
        m.update_decimal(ksize);
        for (uint64_t h : mins) m.update_decimal(h);
        return m.hexdigest();
    }

    // minhash.rs:223-237, 262-280
    void set_hash_function(uint32_t hf) {
        if (hash_function == hf) return;
        if (!mins.empty()) throw err_non_empty("hash_function");
        hash_function = hf;
    }
    void enable_abundance() {
        if (!mins.empty()) throw err_non_empty("track_abundance=True");
        touch();
        track_abundance = true;
        abunds.clear();
    }
    void disable_abundance() { touch(); track_abundance = false; abunds.clear(); }
};

}  // namespace smg
