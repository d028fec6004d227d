// collection.hpp -- bulk loading of signature collections into one CSR (host side, multi-threaded).
//
// the step before compare / gather.  The reference builds one Python (or Rust)
// object per sketch -- src/core/src/signature.rs:569-659 (serde JSON, gzip sniffing), src/sourmash/save_load.py:
// 218-234,448-549 (.sig / .sig.gz / .zip / directory / pathlist dispatch), src/sourmash/manifest.py:15-387 (CSV
// manifest: selection without opening the sketches), src/core/src/{collection,manifest}.rs.  Here a collection
// goes from files straight to `hashes: u64[total]` + `offsets: u64[n+1]` plus one manifest row per sketch:
//   * inputs are expanded to work items (a .sig/.sig.gz file, or ONE member of a zip picked through the zip's
//     SOURMASH-MANIFEST.csv so non-matching members are never inflated);
//   * worker threads inflate (zlib) and scan the JSON with a pull scanner that writes the `mins` array straight
//     into a u64 vector -- no DOM, no per-number allocation;
//   * selection (ksize, moltype, scaled <= target, then downsampling to the target) happens in the worker;
//   * the pieces are concatenated in input order, so the row order is deterministic.
#pragma once
#include <dirent.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <unordered_map>
#include <mutex>
#include <memory>
#include <vector>
#include <hip/hip_runtime_api.h>
#include "json.hpp"
#include "md5.hpp"
#include "minhash_host.hpp"
#include "signature_host.hpp"

namespace smg {

// ---- small file helpers -------------------------------------------------------------------------------------
inline std::string read_whole_file(const std::string& path) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw Error(E_IO, "cannot open " + path);
    struct stat st;
    if (fstat(fd, &st) != 0) { ::close(fd); throw Error(E_IO, "cannot stat " + path); }
    std::string out((size_t)st.st_size, '\0');
    size_t got = 0;
    while (got < out.size()) {
        const ssize_t r = ::read(fd, &out[got], out.size() - got);
        if (r <= 0) { ::close(fd); throw Error(E_IO, "short read on " + path); }
        got += (size_t)r;
    }
    ::close(fd);
    return out;
}

inline bool has_suffix(const std::string& s, const char* suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// ---- zip container (the subset Python's zipfile / the Rust zip crate write: stored + deflate, zip64) ---------
struct ZipMember {
    std::string name;
    uint16_t method = 0;
    uint64_t comp_size = 0, size = 0, local_offset = 0;
};

class ZipReader {
  public:
    explicit ZipReader(const std::string& path) : path_(path) {
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) throw Error(E_IO, "cannot open " + path);
        struct stat st;
        if (fstat(fd_, &st) != 0) fail("cannot stat");
        file_size_ = (uint64_t)st.st_size;
        read_directory();
    }
    ~ZipReader() { if (fd_ >= 0) ::close(fd_); }
    ZipReader(const ZipReader&) = delete;
    ZipReader& operator=(const ZipReader&) = delete;

    const std::vector<ZipMember>& members() const { return members_; }
    const ZipMember* find(const std::string& name) const {
        auto it = by_name_.find(name);
        return it == by_name_.end() ? nullptr : &members_[it->second];
    }
    // thread-safe (pread): the bytes of a STORED member as they lie in the archive, into dst[0, m.comp_size); false for a
    // compressed member (the device loader takes .sig.gz members stored, as sourmash writes them)
    bool read_stored_into(const ZipMember& m, uint8_t* dst) const {
        if (m.method != 0) return false;
        uint8_t lh[30];
        pread_exact(m.local_offset, lh, 30);
        if (le32(lh) != 0x04034b50u) fail("bad local file header");
        const uint64_t data = m.local_offset + 30 + le16(lh + 26) + le16(lh + 28);
        if (m.comp_size > file_size_ || data > file_size_ - m.comp_size) fail("member " + m.name + " reaches past the end of the archive");
        if (m.comp_size) pread_exact(data, dst, m.comp_size);
        return true;
    }
    // thread-safe (pread): the member's bytes, inflated
    std::string read(const ZipMember& m) const {
        uint8_t lh[30];
        pread_exact(m.local_offset, lh, 30);
        if (le32(lh) != 0x04034b50u) fail("bad local file header");
        const uint64_t data = m.local_offset + 30 + le16(lh + 26) + le16(lh + 28);
        // the central directory is untrusted input: a member cannot be larger than the archive holds / than deflate can expand
        if (m.comp_size > file_size_ || data > file_size_ - m.comp_size) fail("member " + m.name + " reaches past the end of the archive");
        if (m.method == 8 && m.size / 1032 > m.comp_size + 1) fail("implausible size of member " + m.name);
        std::string comp((size_t)m.comp_size, '\0');
        if (m.comp_size) pread_exact(data, &comp[0], m.comp_size);
        if (m.method == 0) return comp;
        if (m.method != 8) fail("unsupported compression method");
        std::string out((size_t)m.size, '\0');
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, -MAX_WBITS) != Z_OK) fail("cannot initialise inflate");
        // zlib counts in uInt: feed and drain in pieces of at most 1 GiB (zip64 members may exceed 4 GiB)
        const size_t PIECE = (size_t)1 << 30;
        size_t in_pos = 0, out_pos = 0;
        int rc = Z_OK;
        while (rc == Z_OK) {
            if (zs.avail_in == 0 && in_pos < comp.size()) {
                const size_t n = std::min(PIECE, comp.size() - in_pos);
                zs.next_in = (Bytef*)comp.data() + in_pos;
                zs.avail_in = (uInt)n;
                in_pos += n;
            }
            if (zs.avail_out == 0 && out_pos < out.size()) {
                const size_t n = std::min(PIECE, out.size() - out_pos);
                zs.next_out = (Bytef*)&out[0] + out_pos;
                zs.avail_out = (uInt)n;
                out_pos += n;
            }
            const bool last_in = in_pos == comp.size();
            rc = inflate(&zs, last_in ? Z_FINISH : Z_NO_FLUSH);
            // no progress possible now: go round again only if the next pass really refills the side that ran dry
            // (a corrupt stream can ask for input or output that does not exist -- it must fail, not spin)
            if (rc == Z_BUF_ERROR && ((zs.avail_in == 0 && in_pos < comp.size()) || (zs.avail_out == 0 && out_pos < out.size()))) rc = Z_OK;
        }
        const bool ok = (rc == Z_STREAM_END) && (out_pos - zs.avail_out) == out.size() && in_pos == comp.size();
        inflateEnd(&zs);
        if (!ok) fail("corrupt deflate stream in member " + m.name);
        return out;
    }

  private:
    [[noreturn]] void fail(const std::string& what) const { throw Error(E_STORAGE, "zip " + path_ + ": " + what); }
    static uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
    static uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
    static uint64_t le64(const uint8_t* p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }
    void pread_exact(uint64_t off, void* dst, uint64_t n) const {
        uint64_t got = 0;
        while (got < n) {
            const ssize_t r = ::pread(fd_, (char*)dst + got, n - got, (off_t)(off + got));
            if (r <= 0) fail("short read");
            got += (uint64_t)r;
        }
    }
    void read_directory() {
        if (file_size_ < 22) fail("not a zip file");
        const uint64_t tail = std::min<uint64_t>(file_size_, 65536 + 22);
        std::vector<uint8_t> buf(tail);
        pread_exact(file_size_ - tail, buf.data(), tail);
        int64_t pos = -1;
        for (int64_t i = (int64_t)tail - 22; i >= 0; --i)
            if (le32(&buf[i]) == 0x06054b50u) { pos = i; break; }
        if (pos < 0) fail("end-of-central-directory record not found");
        uint64_t n_entries = le16(&buf[pos + 10]), cd_size = le32(&buf[pos + 12]), cd_off = le32(&buf[pos + 16]);
        if (n_entries == 0xffff || cd_size == 0xffffffffu || cd_off == 0xffffffffu) {     // zip64
            if (pos < 20 || le32(&buf[pos - 20]) != 0x07064b50u) fail("zip64 locator missing");
            const uint64_t eocd64 = le64(&buf[pos - 20 + 8]);
            uint8_t rec[56];
            pread_exact(eocd64, rec, 56);
            if (le32(rec) != 0x06064b50u) fail("bad zip64 end-of-central-directory record");
            n_entries = le64(rec + 32);
            cd_size = le64(rec + 40);
            cd_off = le64(rec + 48);
        }
        std::vector<uint8_t> cd(cd_size);
        if (cd_size) pread_exact(cd_off, cd.data(), cd_size);
        uint64_t p = 0;
        for (uint64_t e = 0; e < n_entries; ++e) {
            if (p + 46 > cd_size || le32(&cd[p]) != 0x02014b50u) fail("bad central directory entry");
            ZipMember m;
            m.method = le16(&cd[p + 10]);
            m.comp_size = le32(&cd[p + 20]);
            m.size = le32(&cd[p + 24]);
            const uint16_t nlen = le16(&cd[p + 28]), xlen = le16(&cd[p + 30]), clen = le16(&cd[p + 32]);
            m.local_offset = le32(&cd[p + 42]);
            if (p + 46 + nlen + xlen + clen > cd_size) fail("truncated central directory");
            m.name.assign((const char*)&cd[p + 46], nlen);
            // zip64 extended information (id 1): only the fields that overflowed are present, in this order
            uint64_t x = p + 46 + nlen;
            const uint64_t xend = x + xlen;
            while (x + 4 <= xend) {
                const uint16_t id = le16(&cd[x]), sz = le16(&cd[x + 2]);
                if (id == 1) {
                    uint64_t q = x + 4;
                    if (m.size == 0xffffffffu && q + 8 <= xend) { m.size = le64(&cd[q]); q += 8; }
                    if (m.comp_size == 0xffffffffu && q + 8 <= xend) { m.comp_size = le64(&cd[q]); q += 8; }
                    if (m.local_offset == 0xffffffffu && q + 8 <= xend) { m.local_offset = le64(&cd[q]); q += 8; }
                }
                x += 4 + sz;
            }
            members_.push_back(std::move(m));
            p += 46 + nlen + xlen + clen;
        }
        for (size_t i = 0; i < members_.size(); ++i) by_name_.emplace(members_[i].name, i);
    }

    std::string path_;
    int fd_ = -1;
    uint64_t file_size_ = 0;
    std::vector<ZipMember> members_;
    std::unordered_map<std::string, size_t> by_name_;
};

// ---- CSV (RFC 4180 quoting, as Python's csv module writes manifests) ------------------------------------------
inline std::vector<std::vector<std::string>> parse_csv(const std::string& text) {
    std::vector<std::vector<std::string>> rows;
    std::vector<std::string> row;
    std::string field;
    bool quoted = false, any = false;
    for (size_t i = 0; i < text.size(); ++i) {
        const char c = text[i];
        if (quoted) {
            if (c == '"') {
                if (i + 1 < text.size() && text[i + 1] == '"') { field += '"'; ++i; }
                else quoted = false;
            } else field += c;
            continue;
        }
        if (c == '"') { quoted = true; any = true; }
        else if (c == ',') { row.push_back(std::move(field)); field.clear(); any = true; }
        else if (c == '\n' || c == '\r') {
            if (c == '\r' && i + 1 < text.size() && text[i + 1] == '\n') ++i;
            if (any || !field.empty()) { row.push_back(std::move(field)); rows.push_back(std::move(row)); }
            row.clear(); field.clear(); any = false;
        } else { field += c; any = true; }
    }
    if (any || !field.empty()) { row.push_back(std::move(field)); rows.push_back(std::move(row)); }
    return rows;
}

inline void csv_field(std::string& out, const std::string& s) {
    if (s.find_first_of(",\"\r\n") == std::string::npos) { out += s; return; }
    out += '"';
    for (char c : s) { if (c == '"') out += '"'; out += c; }
    out += '"';
}

// ---- manifest rows (src/sourmash/manifest.py:29-41 required_keys) ----------------------------------------------
struct ManifestRow {
    std::string internal_location, md5, name, filename, moltype;
    uint32_t ksize = 0;
    uint64_t num = 0, scaled = 0, n_hashes = 0;
    bool with_abundance = false;
};

inline std::vector<ManifestRow> parse_manifest_csv(const std::string& text) {
    static const char* VERSION = "# SOURMASH-MANIFEST-VERSION: ";
    const size_t eol = text.find('\n');
    if (text.compare(0, strlen(VERSION), VERSION) != 0 || eol == std::string::npos)
        throw Error(E_CSV, "manifest is missing version header");                       // manifest.py:64-65
    if (atof(text.c_str() + strlen(VERSION)) != 1.0) throw Error(E_CSV, "unknown manifest version number");
    auto rows = parse_csv(text.substr(eol + 1));
    if (rows.empty()) throw Error(E_CSV, "missing column headers in manifest");
    const char* keys[] = {"internal_location", "md5", "md5short", "ksize", "moltype", "num", "scaled", "n_hashes",
                          "with_abundance", "name", "filename"};
    int col[11];
    for (int k = 0; k < 11; ++k) {
        col[k] = -1;
        for (size_t c = 0; c < rows[0].size(); ++c) if (rows[0][c] == keys[k]) col[k] = (int)c;
        if (col[k] < 0) throw Error(E_CSV, std::string("missing column '") + keys[k] + "' in manifest.");
    }
    std::vector<ManifestRow> out;
    for (size_t r = 1; r < rows.size(); ++r) {
        auto& f = rows[r];
        auto get = [&](int k) -> const std::string& {
            static const std::string empty;
            return (size_t)col[k] < f.size() ? f[col[k]] : empty;
        };
        ManifestRow m;
        m.internal_location = get(0);
        m.md5 = get(1);
        m.ksize = (uint32_t)strtoul(get(3).c_str(), nullptr, 10);
        m.moltype = get(4);
        m.num = strtoull(get(5).c_str(), nullptr, 10);
        m.scaled = strtoull(get(6).c_str(), nullptr, 10);
        m.n_hashes = strtoull(get(7).c_str(), nullptr, 10);
        const std::string& ab = get(8);
        m.with_abundance = ab == "1" || ab == "True" || ab == "true";
        m.name = get(9);
        m.filename = get(10);
        out.push_back(std::move(m));
    }
    return out;
}

inline std::string manifest_to_csv(const std::vector<ManifestRow>& rows) {
    std::string out = "# SOURMASH-MANIFEST-VERSION: 1.0\n"
                      "internal_location,md5,md5short,ksize,moltype,num,scaled,n_hashes,with_abundance,name,filename\r\n";
    for (auto& m : rows) {                                               // rows end in \r\n like Python's csv writer
        csv_field(out, m.internal_location); out += ',';
        out += m.md5; out += ',';
        out += m.md5.substr(0, 8); out += ',';
        out += std::to_string(m.ksize); out += ',';
        out += m.moltype; out += ',';
        out += std::to_string(m.num); out += ',';
        out += std::to_string(m.scaled); out += ',';
        out += std::to_string(m.n_hashes); out += ',';
        out += m.with_abundance ? "1" : "0"; out += ',';
        csv_field(out, m.name); out += ',';
        csv_field(out, m.filename);
        out += "\r\n";
    }
    return out;
}

// ---- selection --------------------------------------------------------------------------------------------------
struct LoadSelect {
    uint32_t ksize = 0;            // 0: any
    int hash_function = -1;        // -1: any, else HashFunctions value
    uint64_t scaled = 0;           // 0: keep as stored; else keep sketches with 0 < scaled <= target, downsampled to it
};

inline const char* manifest_moltype(uint32_t hf) { return molecule_name(hf); }   // "DNA", "protein", "dayhoff", "hp"

// a `mins` array the device has parsed (sigload.hpp): where its values lie, how many, how many survive the down-sampling
struct DeviceArray { uint64_t value_off = 0; uint32_t n_values = 0, n_kept = 0; bool is_mins = false, odd = false; };
struct NeedsHost {};                       // thrown by the scanner in device mode: this document is the host parser's

struct LoadedPiece {
    std::vector<uint64_t> hashes;          // rows back to back (host mode)
    std::vector<uint64_t> dev_off;         // per row: where its values lie in the group's value block (device mode)
    int group = -1;                        // device mode: which group's value block
    std::vector<uint64_t> lens;            // per row
    std::vector<ManifestRow> rows;         // as stored (before downsampling), one per row
    std::vector<uint64_t> seeds;
    uint64_t skipped = 0;                  // sketches seen but not selected
};

// ---- pull scanner for signature JSON (signature.rs:569-659 + minhash.rs:134-184 field semantics) ---------------
class SigScanner : public json::Parser {
  public:
    SigScanner(const char* p, size_t n) : json::Parser(p, n) {}
    // device mode: the text's `mins` arrays hold one number each -- the index of the array in `arrays` -- and the values
    // themselves are in HBM (sigload.hpp)
    SigScanner(const char* p, size_t n, const DeviceArray* arrays, size_t n_arrays) : json::Parser(p, n), arrays_(arrays), n_arrays_(n_arrays) {}

    void scan(const LoadSelect& sel, const std::string& location, LoadedPiece& out) {
        skip_ws();
        if (p_ < end_ && *p_ == '[') {
            ++p_;
            if (eat(']')) return;
            for (;;) {
                signature(sel, location, out);
                if (eat(',')) continue;
                if (eat(']')) break;
                fail("expected ',' or ']'");
            }
        } else {
            signature(sel, location, out);                           // a bare signature object
        }
        skip_ws();
        if (p_ != end_) fail("trailing characters");
    }

  private:
    struct Sketch {
        uint64_t num = 0, seed = 42, max_hash = 0;
        uint32_t ksize = 0;
        std::string md5, molecule;
        std::vector<uint64_t> mins;
        bool has_abund = false, seen_mins = false, seen_molecule = false;
    };

    std::string key() {
        skip_ws();
        if (p_ >= end_ || *p_ != '"') fail("expected object key");
        ++p_;
        std::string k = parse_string_body();
        if (!eat(':')) fail("expected ':'");
        return k;
    }
    std::string string_or_empty() {                                  // name / filename may be null
        skip_ws();
        if (p_ < end_ && *p_ == '"') { ++p_; return parse_string_body(); }
        skip_value();
        return std::string();
    }
    void skip_value() {
        skip_ws();
        if (p_ >= end_) fail("unexpected end");
        const char c = *p_;
        if (c == '"') { ++p_; (void)parse_string_body(); return; }
        if (c == '{' || c == '[') {
            const char close = c == '{' ? '}' : ']';
            ++p_;
            if (eat(close)) return;
            for (;;) {
                if (c == '{') (void)key();
                skip_value();
                if (eat(',')) continue;
                if (eat(close)) return;
                fail("expected ',' or close");
            }
        }
        if (c == 't') { expect_word("true"); return; }
        if (c == 'f') { expect_word("false"); return; }
        if (c == 'n') { expect_word("null"); return; }
        if (c == '-' || (c >= '0' && c <= '9')) { (void)number(); return; }
        fail("unexpected character");
    }
    uint64_t number() {
        skip_ws();
        const char* s = p_;
        uint64_t v = 0;
        while (p_ < end_ && *p_ >= '0' && *p_ <= '9') v = v * 10 + (uint64_t)(*p_++ - '0');
        if (p_ < end_ && (*p_ == '.' || *p_ == 'e' || *p_ == 'E' || *p_ == '-' || p_ == s)) {   // 123.0, 1e3: rare
            if (p_ == s && *p_ != '-') fail("expected a number");
            while (p_ < end_ && ((*p_ >= '0' && *p_ <= '9') || *p_ == '.' || *p_ == 'e' || *p_ == 'E' || *p_ == '+' || *p_ == '-')) ++p_;
            const std::string t(s, (size_t)(p_ - s));
            const double d = strtod(t.c_str(), nullptr);
            if (d < 0) fail("expected an unsigned integer");
            return (uint64_t)d;
        }
        return v;
    }
    void u64_array(std::vector<uint64_t>* dst) {
        if (!eat('[')) fail("expected an array");
        if (eat(']')) return;
        for (;;) {
            const uint64_t v = number();
            if (dst) dst->push_back(v);
            skip_ws();
            if (p_ < end_ && *p_ == ',') { ++p_; continue; }
            if (p_ < end_ && *p_ == ']') { ++p_; return; }
            fail("expected ',' or ']'");
        }
    }

    void sketch(Sketch& sk) {
        if (!eat('{')) fail("sketch must be an object");
        bool seen_num = false, seen_ksize = false, seen_seed = false, seen_max = false, seen_md5 = false;
        if (!eat('}')) for (;;) {
            const std::string k = key();
            if (k == "mins") { sk.mins.clear(); u64_array(&sk.mins); sk.seen_mins = true; }
            else if (k == "abundances") {
                skip_ws();
                if (p_ < end_ && *p_ == '[') { u64_array(nullptr); sk.has_abund = true; } else skip_value();
            }
            else if (k == "num") { sk.num = number(); seen_num = true; }
            else if (k == "ksize") { sk.ksize = (uint32_t)number(); seen_ksize = true; }
            else if (k == "seed") { sk.seed = number(); seen_seed = true; }
            else if (k == "max_hash") { sk.max_hash = number(); seen_max = true; }
            else if (k == "md5sum") { sk.md5 = string_or_empty(); seen_md5 = true; }
            else if (k == "molecule") { sk.molecule = string_or_empty(); sk.seen_molecule = true; }
            else skip_value();
            if (eat(',')) continue;
            if (eat('}')) break;
            fail("expected ',' or '}'");
        }
        const char* missing = !seen_num ? "num" : !seen_ksize ? "ksize" : !seen_seed ? "seed" : !seen_max ? "max_hash"
                            : !seen_md5 ? "md5sum" : !sk.seen_mins ? "mins" : !sk.seen_molecule ? "molecule" : nullptr;
        if (missing) throw Error(E_SERDE, std::string("missing field `") + missing + "`");
    }

    void signature(const LoadSelect& sel, const std::string& location, LoadedPiece& out) {
        if (!eat('{')) fail("signature must be an object");
        std::string name, filename;
        bool seen_sigs = false, seen_hf = false;
        const size_t first_row = out.rows.size();
        if (!eat('}')) for (;;) {
            const std::string k = key();
            if (k == "name") name = string_or_empty();
            else if (k == "filename") filename = string_or_empty();
            else if (k == "hash_function") { skip_value(); seen_hf = true; }
            else if (k == "signatures") {
                seen_sigs = true;
                if (!eat('[')) fail("signatures must be an array");
                if (!eat(']')) for (;;) {
                    sketch(tmp_);
                    take(sel, location, out);
                    if (eat(',')) continue;
                    if (eat(']')) break;
                    fail("expected ',' or ']'");
                }
            } else skip_value();
            if (eat(',')) continue;
            if (eat('}')) break;
            fail("expected ',' or '}'");
        }
        if (!seen_hf) throw Error(E_SERDE, "missing field `hash_function`");
        if (!seen_sigs) throw Error(E_SERDE, "missing field `signatures`");
        for (size_t r = first_row; r < out.rows.size(); ++r) {      // name/filename may follow the sketches in the file
            out.rows[r].name = name;
            out.rows[r].filename = filename;
        }
    }

    void take(const LoadSelect& sel, const std::string& location, LoadedPiece& out) {
        Sketch& sk = tmp_;
        const uint32_t hf = molecule_from_name(sk.molecule);
        const uint64_t stored_scaled = sk.max_hash ? scaled_for_max_hash(sk.max_hash) : 0;
        const bool keep = (sel.ksize == 0 || sk.ksize == sel.ksize * (hf == HF_DNA ? 1u : 3u)) &&
                          (sel.hash_function < 0 || (uint32_t)sel.hash_function == hf) &&
                          (sel.scaled == 0 || (stored_scaled != 0 && stored_scaled <= sel.scaled));
        if (!keep) { ++out.skipped; reset(); return; }
        if (arrays_) { take_device(sel, location, out, hf, stored_scaled); return; }
        if (!std::is_sorted(sk.mins.begin(), sk.mins.end())) std::sort(sk.mins.begin(), sk.mins.end());   // minhash.rs:161-171
        sk.mins.erase(std::unique(sk.mins.begin(), sk.mins.end()), sk.mins.end());
        ManifestRow row;
        row.internal_location = location;
        row.ksize = hf == HF_DNA ? sk.ksize : sk.ksize / 3;
        row.moltype = manifest_moltype(hf);
        row.num = sk.max_hash ? 0 : sk.num;                          // minhash.rs:150
        row.scaled = stored_scaled;
        row.n_hashes = sk.mins.size();
        row.with_abundance = sk.has_abund;
        row.md5 = sk.md5.empty() ? mins_md5(sk.ksize, sk.mins) : sk.md5;
        size_t n = sk.mins.size();
        if (sel.scaled && stored_scaled != sel.scaled) {             // downsample: keep h <= max_hash(target)
            const uint64_t mx = max_hash_for_scaled(sel.scaled);
            n = (size_t)(std::upper_bound(sk.mins.begin(), sk.mins.end(), mx) - sk.mins.begin());
        }
        out.hashes.insert(out.hashes.end(), sk.mins.begin(), sk.mins.begin() + n);
        out.lens.push_back(n);
        out.seeds.push_back(sk.seed);
        out.rows.push_back(std::move(row));
        reset();
    }
    void take_device(const LoadSelect& sel, const std::string& location, LoadedPiece& out, uint32_t hf, uint64_t stored_scaled) {
        Sketch& sk = tmp_;
        if (sk.mins.size() != 1 || sk.mins[0] >= n_arrays_ || sk.md5.empty()) throw NeedsHost();
        const DeviceArray& a = arrays_[sk.mins[0]];
        if (!a.is_mins || a.odd) throw NeedsHost();
        ManifestRow row;
        row.internal_location = location;
        row.ksize = hf == HF_DNA ? sk.ksize : sk.ksize / 3;
        row.moltype = manifest_moltype(hf);
        row.num = sk.max_hash ? 0 : sk.num;                          // minhash.rs:150
        row.scaled = stored_scaled;
        row.n_hashes = a.n_values;
        row.with_abundance = sk.has_abund;
        row.md5 = sk.md5;
        const uint64_t n = sel.scaled && stored_scaled != sel.scaled ? a.n_kept : a.n_values;   // downsample: keep h <= max_hash(target)
        out.dev_off.push_back(a.value_off);
        out.lens.push_back(n);
        out.seeds.push_back(sk.seed);
        out.rows.push_back(std::move(row));
        reset();
    }
    void reset() {
        tmp_.num = 0; tmp_.seed = 42; tmp_.max_hash = 0; tmp_.ksize = 0;
        tmp_.md5.clear(); tmp_.molecule.clear(); tmp_.mins.clear();
        tmp_.has_abund = tmp_.seen_mins = tmp_.seen_molecule = false;
    }
    static std::string mins_md5(uint32_t ksize, const std::vector<uint64_t>& mins) {     // minhash.rs:290-307
        Md5 h;
        h.update_decimal(ksize);
        for (uint64_t m : mins) h.update_decimal(m);
        return h.hexdigest();
    }

    Sketch tmp_;
    const DeviceArray* arrays_ = nullptr;
    size_t n_arrays_ = 0;
};

// ---- the loader -----------------------------------------------------------------------------------------------------
struct WorkItem {
    std::string path;          // file on disk
    std::string member;        // zip member name ("" for a plain file)
    const ZipReader* zip = nullptr;
};

struct LoadedCollection {
    std::vector<uint64_t> hashes, offsets;
    std::vector<ManifestRow> rows;
    uint32_t ksize = 0, hash_function = 1;
    uint64_t seed = 42, max_hash = 0, num = 0, skipped = 0;
};

inline bool looks_like_sig_name(const std::string& n) { return has_suffix(n, ".sig") || has_suffix(n, ".sig.gz"); }

inline void walk_directory(const std::string& dir, std::vector<std::string>& out) {    // sourmash_args.py:275-295
    std::vector<std::string> files, dirs;
    DIR* d = opendir(dir.c_str());
    if (!d) throw Error(E_IO, "cannot open directory " + dir);
    while (struct dirent* e = readdir(d)) {
        const std::string n = e->d_name;
        if (n == "." || n == "..") continue;
        const std::string full = dir + "/" + n;
        struct stat st;
        if (stat(full.c_str(), &st) != 0) continue;
        if (S_ISDIR(st.st_mode)) dirs.push_back(full);
        else if (looks_like_sig_name(n)) files.push_back(full);
    }
    closedir(d);
    std::sort(files.begin(), files.end());
    std::sort(dirs.begin(), dirs.end());
    out.insert(out.end(), files.begin(), files.end());
    for (auto& sub : dirs) walk_directory(sub, out);
}

class CollectionLoader {
  public:
    CollectionLoader(const LoadSelect& sel, unsigned n_threads) : sel_(sel), n_threads_(n_threads) {}

    void add_path(const std::string& path, int depth = 0) {
        struct stat st;
        if (stat(path.c_str(), &st) != 0) throw Error(E_IO, "cannot open " + path);
        if (S_ISDIR(st.st_mode)) {
            std::vector<std::string> files;
            walk_directory(path, files);
            for (auto& f : files) items_.push_back(WorkItem{f, "", nullptr});
            return;
        }
        uint8_t magic[4] = {0, 0, 0, 0};
        {
            const int fd = ::open(path.c_str(), O_RDONLY);
            if (fd < 0) throw Error(E_IO, "cannot open " + path);
            const ssize_t r = ::read(fd, magic, 4);
            (void)r;
            ::close(fd);
        }
        if (magic[0] == 'P' && magic[1] == 'K') { add_zip(path); return; }
        if ((magic[0] == 0x1f && magic[1] == 0x8b) || magic[0] == '[' || magic[0] == '{') {
            items_.push_back(WorkItem{path, "", nullptr});
            return;
        }
        // a text file listing one path per line (sourmash_args.py load_pathlist_from_file)
        if (depth > 0) throw Error(E_SERDE, "cannot load signatures from " + path);
        const std::string text = read_whole_file(path);
        size_t a = 0;
        unsigned listed = 0;
        while (a < text.size()) {
            size_t b = text.find('\n', a);
            if (b == std::string::npos) b = text.size();
            std::string line = text.substr(a, b - a);
            while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
            if (!line.empty()) { add_path(line, depth + 1); ++listed; }
            a = b + 1;
        }
        if (!listed) throw Error(E_SERDE, "cannot load signatures from " + path);
    }

    LoadedCollection run() {
        std::vector<LoadedPiece> pieces(items_.size());
        std::atomic<size_t> next(0);
        std::vector<Error> errors;
        std::mutex err_mutex;
        unsigned nt = n_threads_ ? n_threads_ : std::max(1u, std::thread::hardware_concurrency());
        if (nt > items_.size()) nt = (unsigned)std::max<size_t>(items_.size(), 1);
        auto worker = [&]() {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= items_.size()) return;
                try {
                    const WorkItem& w = items_[i];
                    std::string raw = w.zip ? w.zip->read(*w.zip->find(w.member)) : read_whole_file(w.path);
                    const std::string text = maybe_gunzip(raw.data(), raw.size());
                    SigScanner sc(text.data(), text.size());
                    sc.scan(sel_, w.zip ? w.member : w.path, pieces[i]);
                } catch (const Error& e) {
                    std::lock_guard<std::mutex> g(err_mutex);
                    errors.push_back(Error(e.code, items_[i].path + (items_[i].member.empty() ? "" : ":" + items_[i].member) + ": " + e.what()));
                }
            }
        };
        std::vector<std::thread> threads;
        for (unsigned t = 1; t < nt; ++t) threads.emplace_back(worker);
        worker();
        for (auto& t : threads) t.join();
        if (!errors.empty()) throw errors.front();
        LoadedCollection out;
        uint64_t total = 0, n = 0;
        for (auto& p : pieces) { total += p.hashes.size(); n += p.lens.size(); out.skipped += p.skipped; }
        out.skipped += manifest_skipped_;
        out.hashes.reserve(total);
        out.offsets.reserve(n + 1);
        out.offsets.push_back(0);
        out.rows.reserve(n);
        bool first = true;
        for (auto& p : pieces) {
            out.hashes.insert(out.hashes.end(), p.hashes.begin(), p.hashes.end());
            for (size_t r = 0; r < p.lens.size(); ++r) {
                out.offsets.push_back(out.offsets.back() + p.lens[r]);
                const ManifestRow& row = p.rows[r];
                const uint32_t hf = molecule_from_name(row.moltype);
                const uint64_t eff_scaled = sel_.scaled ? sel_.scaled : row.scaled;
                if (first) {
                    out.ksize = row.ksize; out.hash_function = hf; out.seed = p.seeds[r];
                    out.max_hash = max_hash_for_scaled(eff_scaled); out.num = row.num;
                    first = false;
                } else {                                             // one CSR = one parameter set (check_compatible order)
                    if (row.ksize != out.ksize) throw Error(E_MISMATCH_KSIZES, "different ksizes cannot be compared");
                    if (hf != out.hash_function) throw Error(E_MISMATCH_DNA_PROT, "DNA/prot minhashes cannot be compared");
                    if (max_hash_for_scaled(eff_scaled) != out.max_hash) throw Error(E_MISMATCH_SCALED, "mismatch in scaled; comparison fail");
                    if (p.seeds[r] != out.seed) throw Error(E_MISMATCH_SEED, "mismatch in seed; comparison fail");
                    if (row.num != out.num) throw Error(E_MISMATCH_NUM, "mismatch in num; comparison fail");
                }
            }
            for (auto& row : p.rows) out.rows.push_back(std::move(row));
            std::vector<uint64_t>().swap(p.hashes);
        }
        return out;
    }

    size_t work_items() const { return items_.size(); }

    // sigload.hpp: the same collection with inflate and number parsing on the device, the CSR left in HBM
    struct DeviceResult;
    void run_device(hipStream_t stream, DeviceResult& out);

  private:
    void add_zip(const std::string& path) {
        zips_.emplace_back(new ZipReader(path));
        const ZipReader* z = zips_.back().get();
        if (const ZipMember* mf = z->find("SOURMASH-MANIFEST.csv")) {
            // selection on the manifest: members that cannot match are never inflated (manifest.py:256-323)
            std::string last;
            for (const ManifestRow& row : parse_manifest_csv(z->read(*mf))) {
                const uint32_t hf = molecule_from_name(row.moltype);
                const bool keep = (sel_.ksize == 0 || row.ksize == sel_.ksize) &&
                                  (sel_.hash_function < 0 || (uint32_t)sel_.hash_function == hf) &&
                                  (sel_.scaled == 0 || (row.scaled != 0 && row.scaled <= sel_.scaled));
                if (!keep) { ++manifest_skipped_; continue; }
                if (row.internal_location == last) continue;         // one member can hold several selected sketches
                if (!z->find(row.internal_location)) throw Error(E_STORAGE, "zip " + path + ": manifest names a missing member " + row.internal_location);
                items_.push_back(WorkItem{path, row.internal_location, z});
                last = row.internal_location;
            }
            return;
        }
        for (const ZipMember& m : z->members())                      // no manifest: every .sig / .sig.gz member
            if (looks_like_sig_name(m.name)) items_.push_back(WorkItem{path, m.name, z});
    }

    LoadSelect sel_;
    unsigned n_threads_;
    std::vector<WorkItem> items_;
    std::vector<std::unique_ptr<ZipReader>> zips_;
    uint64_t manifest_skipped_ = 0;
};

}  // namespace smg
