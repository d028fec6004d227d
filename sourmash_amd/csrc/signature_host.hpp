// Host-side signature container behind `SourmashSignature*` and
// `SourmashComputeParameters*`.
//
// Mirrors src/core/src/signature.rs:401-912 (Signature: metadata + sketches,
// add_sequence fan-out :661-677, JSON load/save :569-659,786-794),
// src/core/src/cmd.rs:9-20,24-100,108-188 (ComputeParameters, build_template)
// and the serde layout of src/core/src/sketch/minhash.rs:103-184.
#pragma once
#include <stdio.h>
#include <zlib.h>
#include <optional>
#include <string>
#include <vector>
#include "json.hpp"
#include "minhash_host.hpp"

namespace smg {

struct ComputeParameters {   // cmd.rs:24-100 defaults
    std::vector<uint32_t> ksizes{21, 31, 51};
    bool check_sequence = false, dna = true, dayhoff = false, hp = false, singleton = false;
    uint64_t scaled = 0;
    bool force = false;
    uint32_t num_hashes = 500;
    bool protein = false, name_from_first = false;
    uint64_t seed = 42;
    bool input_is_protein = false;
    std::optional<std::string> merge;
    bool track_abundance = false, randomize = false;
    std::string license = "CC0";
    size_t processes = 2;
};

inline const char* molecule_name(uint32_t hf) {   // encodings.rs:54-69
    switch (hf) {
    case HF_DNA: return "DNA";
    case HF_PROTEIN: return "protein";
    case HF_DAYHOFF: return "dayhoff";
    case HF_HP: return "hp";
    default: return "DNA";
    }
}

inline uint32_t molecule_from_name(const std::string& s) {   // encodings.rs:71-84 (case-insensitive)
    std::string l;
    for (char c : s) l += (char)((c >= 'A' && c <= 'Z') ? c + 32 : c);
    if (l == "dna") return HF_DNA;
    if (l == "protein") return HF_PROTEIN;
    if (l == "dayhoff") return HF_DAYHOFF;
    if (l == "hp") return HF_HP;
    throw Error(E_INVALID_HASH_FUNCTION, "Invalid hash function: \"" + s + "\"");
}

struct Signature {
    std::string klass = "sourmash_signature";
    std::string email;
    std::string hash_function = "0.murmur64";
    std::optional<std::string> filename;
    std::optional<std::string> name;
    std::string license = "CC0";
    std::vector<KmerMinHash> sketches;
    double version = 0.4;

    // cmd.rs:9-20 + build_template :108-188 (per ksize: protein, dayhoff, hp, dna)
    static Signature from_params(const ComputeParameters& p) {
        Signature s;
        s.name = p.merge;
        for (uint32_t k : p.ksizes) {
            if (p.protein) s.sketches.emplace_back(p.scaled, k, HF_PROTEIN, p.seed, p.track_abundance, p.num_hashes);
            if (p.dayhoff) s.sketches.emplace_back(p.scaled, k, HF_DAYHOFF, p.seed, p.track_abundance, p.num_hashes);
            if (p.hp) s.sketches.emplace_back(p.scaled, k, HF_HP, p.seed, p.track_abundance, p.num_hashes);
            if (p.dna) s.sketches.emplace_back(p.scaled, k, HF_DNA, p.seed, p.track_abundance, p.num_hashes);
        }
        return s;
    }

    std::string md5sum() const {   // signature.rs:506-517
        if (sketches.size() == 1) return sketches[0].md5sum();
        throw err_internal("md5sum of a signature with several sketches is not defined");
    }

    bool equals(const Signature& o) const {   // signature.rs:869-888
        const bool meta = klass == o.klass && email == o.email && hash_function == o.hash_function &&
                          filename == o.filename && name == o.name;
        if (sketches.empty() || o.sketches.empty()) return meta && sketches.size() == o.sketches.size();
        return meta && sketches[0].md5sum() == o.sketches[0].md5sum();
    }

    // ---- JSON out ---------------------------------------------------------------------------
    static void sketch_to_json(std::string& out, const KmerMinHash& mh) {   // minhash.rs:103-132
        out += "{\"num\":"; json::write_u64(out, mh.num);
        out += ",\"ksize\":"; json::write_u64(out, mh.ksize);
        out += ",\"seed\":"; json::write_u64(out, mh.seed);
        out += ",\"max_hash\":"; json::write_u64(out, mh.max_hash);
        out += ",\"mins\":[";
        for (size_t i = 0; i < mh.mins.size(); ++i) { if (i) out += ','; json::write_u64(out, mh.mins[i]); }
        out += "],\"md5sum\":\""; out += mh.md5sum(); out += '"';
        if (mh.track_abundance) {
            out += ",\"abundances\":[";
            for (size_t i = 0; i < mh.abunds.size(); ++i) { if (i) out += ','; json::write_u64(out, mh.abunds[i]); }
            out += ']';
        }
        out += ",\"molecule\":\""; out += molecule_name(mh.hash_function); out += "\"}";
    }

    void to_json(std::string& out) const {   // field order of signature.rs:406-431
        out += "{\"class\":"; json::write_string(out, klass);
        out += ",\"email\":"; json::write_string(out, email);
        out += ",\"hash_function\":"; json::write_string(out, hash_function);
        out += ",\"filename\":";
        if (filename) json::write_string(out, *filename); else out += "null";
        if (name) { out += ",\"name\":"; json::write_string(out, *name); }
        out += ",\"license\":"; json::write_string(out, license);
        out += ",\"signatures\":[";
        for (size_t i = 0; i < sketches.size(); ++i) { if (i) out += ','; sketch_to_json(out, sketches[i]); }
        out += "],\"version\":";
        char buf[32]; snprintf(buf, sizeof buf, "%.17g", version);
        // shortest representation for the common 0.4
        if (version == 0.4) out += "0.4"; else out += buf;
        out += '}';
    }

    // ---- JSON in ----------------------------------------------------------------------------
    static KmerMinHash sketch_from_json(const json::Value& v) {   // minhash.rs:134-184
        if (v.kind != json::Value::Object) throw Error(E_SERDE, "JSON: sketch must be an object");
        auto need = [&](const char* k) -> const json::Value& {
            const json::Value* x = v.get(k);
            if (!x) throw Error(E_SERDE, std::string("missing field `") + k + "`");
            return *x;
        };
        KmerMinHash mh;
        mh.max_hash = need("max_hash").as_u64();
        const uint64_t num = need("num").as_u64();
        mh.num = mh.max_hash != 0 ? 0 : (uint32_t)num;                       // :150
        mh.ksize = (uint32_t)need("ksize").as_u64();
        mh.seed = need("seed").as_u64();
        (void)need("md5sum");
        mh.hash_function = molecule_from_name(need("molecule").as_str());
        const json::Value& mins = need("mins");
        if (mins.kind != json::Value::Array) throw Error(E_SERDE, "JSON: mins must be an array");
        const json::Value* ab = v.get("abundances");
        const bool has_ab = ab && ab->kind == json::Value::Array;
        std::vector<std::pair<uint64_t, uint64_t>> pairs;
        pairs.reserve(mins.items.size());
        for (size_t i = 0; i < mins.items.size(); ++i) {
            const uint64_t a = has_ab && i < ab->items.size() ? ab->items[i]->as_u64() : 1;
            pairs.emplace_back(mins.items[i]->as_u64(), a);
        }
        std::sort(pairs.begin(), pairs.end());                                // :161-171 (old files were unsorted)
        mh.track_abundance = has_ab;
        mh.mins.reserve(pairs.size());
        if (has_ab) mh.abunds.reserve(pairs.size());
        for (auto& p : pairs) { mh.mins.push_back(p.first); if (has_ab) mh.abunds.push_back(p.second); }
        mh.touch();
        return mh;
    }

    static Signature from_json(const json::Value& v) {
        if (v.kind != json::Value::Object) throw Error(E_SERDE, "JSON: signature must be an object");
        Signature s;
        if (auto* x = v.get("class")) s.klass = x->as_str();
        if (auto* x = v.get("email")) s.email = x->as_str();
        if (auto* x = v.get("hash_function")) s.hash_function = x->as_str();
        else throw Error(E_SERDE, "missing field `hash_function`");
        if (auto* x = v.get("filename")) { if (x->kind == json::Value::String) s.filename = x->text; }
        if (auto* x = v.get("name")) { if (x->kind == json::Value::String) s.name = x->text; }
        if (auto* x = v.get("license")) s.license = x->as_str();
        if (auto* x = v.get("version")) s.version = x->as_f64();
        const json::Value* sk = v.get("signatures");
        if (!sk || sk->kind != json::Value::Array) throw Error(E_SERDE, "missing field `signatures`");
        for (auto& item : sk->items) s.sketches.push_back(sketch_from_json(*item));
        return s;
    }
};

// gunzip if the buffer starts with the gzip magic (niffler sniffing, signature.rs:584)
inline std::string maybe_gunzip(const char* p, size_t n) {
    if (n < 2 || (unsigned char)p[0] != 0x1f || (unsigned char)p[1] != 0x8b) return std::string(p, n);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) throw Error(E_NIFFLER, "cannot initialise gzip reader");
    std::string out;
    std::vector<char> buf(1 << 16);
    zs.next_in = (Bytef*)p;
    zs.avail_in = (uInt)n;
    int rc = Z_OK;
    while (rc != Z_STREAM_END) {
        zs.next_out = (Bytef*)buf.data();
        zs.avail_out = (uInt)buf.size();
        rc = inflate(&zs, Z_NO_FLUSH);
        if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&zs); throw Error(E_NIFFLER, "corrupt gzip stream"); }
        out.append(buf.data(), buf.size() - zs.avail_out);
        if (rc == Z_STREAM_END && zs.avail_in > 0) {       // concatenated members
            if (inflateReset(&zs) != Z_OK) break;
            rc = Z_OK;
        } else if (rc == Z_OK && zs.avail_in == 0 && zs.avail_out != 0) {
            break;
        }
    }
    inflateEnd(&zs);
    return out;
}

inline std::string gzip_bytes(const std::string& in, int level) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, 16 + MAX_WBITS, 8, Z_DEFAULT_STRATEGY) != Z_OK)
        throw Error(E_NIFFLER, "cannot initialise gzip writer");
    std::string out(deflateBound(&zs, (uLong)in.size()) + 64, '\0');
    zs.next_in = (Bytef*)in.data(); zs.avail_in = (uInt)in.size();
    zs.next_out = (Bytef*)&out[0]; zs.avail_out = (uInt)out.size();
    const int rc = deflate(&zs, Z_FINISH);
    if (rc != Z_STREAM_END) { deflateEnd(&zs); throw Error(E_NIFFLER, "gzip compression failed"); }
    out.resize(zs.total_out);
    deflateEnd(&zs);
    return out;
}

// signature.rs:589-659: parse, flatten to one sketch per signature, filter by ksize / moltype
inline std::vector<Signature> load_signatures(const char* p, size_t n, size_t ksize, const uint32_t* moltype) {
    const std::string text = maybe_gunzip(p, n);
    json::Parser parser(text.data(), text.size());
    json::ValuePtr doc = parser.parse_document();
    if (doc->kind != json::Value::Array) throw Error(E_SERDE, "JSON: expected a list of signatures");
    std::vector<Signature> out;
    for (auto& item : doc->items) {
        Signature s = Signature::from_json(*item);
        for (auto& mh : s.sketches) {
            if (ksize && mh.ksize != ksize) continue;
            if (moltype && mh.hash_function != *moltype) continue;
            Signature one = s;
            one.sketches.clear();
            one.sketches.push_back(mh);
            out.push_back(std::move(one));
        }
    }
    return out;
}

}  // namespace smg
