// gunzip.hpp -- host side of the device inflate (inflate_core.hpp describes the scheme, gunzip.hip holds the kernels).
//
// gunzip_device() takes the bytes of one or more single-member gzip files, already in HBM (and on the host, where the headers
// and trailers are read), and leaves every member's inflated bytes in one device block.  It strings the launches together with
// three read-backs: the candidate list, pass 1's results, and the final checks (pass 2 against pass 1, window errors, CRC-32
// and length against the trailer).  A member that fails any check comes back with ok == false and a reason: the caller then
// inflates it on the host (ingest.hpp: pargz / zlib).
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <vector>
#include "device_ctx.hpp"
#include "gunzip_api.hpp"
#include "inflate_core.hpp"

namespace smg {

struct GunzipMember {
    // in: the member's file bytes inside the host / device buffers (file_off a multiple of 8)
    uint64_t file_off = 0, file_len = 0;
    // out
    bool ok = false;
    std::string why;
    uint64_t out_off = 0, out_len = 0;       // its inflated bytes inside the output block
    uint32_t n_runs = 0;
    uint8_t first_byte = 0;                  // of the inflated bytes ('>' FASTA, '@' FASTQ: the ingest picks its parser by it)
};

struct GunzipStats {
    uint64_t survivors = 0, candidates = 0, runs = 0, pieces = 0, chunks = 0;
    double scan_ms = 0, pass1_ms = 0, link_ms = 0, pass2_ms = 0, finish_ms = 0, total_ms = 0;
};

struct GunzipCounters { std::atomic<uint64_t> on_device{0}, refused{0}; };
inline GunzipCounters& gunzip_counters() { static GunzipCounters c; return c; }

// Bytes the device buffer of the files must have behind total_bytes (zeroed): the scan reads whole words past the end.
constexpr size_t GUNZIP_PAD = 1024;       // (the block walk keeps 128 words in flight behind its position)

// h_files / d_files: the same total_bytes on the host and on the device; d_files 8-byte aligned with GUNZIP_PAD zero bytes behind.
// *d_out: an arena block (the caller releases it with arena_free(*d_out, stream)) or nullptr when no member inflated.
inline void gunzip_device(const uint8_t* h_files, const uint8_t* d_files, uint64_t total_bytes, std::vector<GunzipMember>& members,
                          void** d_out, hipStream_t stream, GunzipStats* stats = nullptr) {
    using namespace inf;
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
    const auto t_start = clk::now();
    *d_out = nullptr;
    GunzipStats st;
    const uint32_t* words = reinterpret_cast<const uint32_t*>(d_files);
    std::vector<Member> hdr(members.size());
    size_t n_live = 0;
    for (size_t i = 0; i < members.size(); ++i) {
        GunzipMember& m = members[i];
        m.ok = false; m.why.clear(); m.out_off = m.out_len = 0; m.n_runs = 0;
        if ((m.file_off & 7) || m.file_off + m.file_len > total_bytes || total_bytes >= ((uint64_t)1 << 32)) { m.why = "member outside the buffer"; continue; }
        if (!parse_single_member(h_files + m.file_off, m.file_len, hdr[i])) { m.why = "not a gzip member"; continue; }
        m.ok = true;
        ++n_live;
    }
    if (!n_live) return;

    // ---- scan: candidate block starts of the whole buffer ----
    auto t0 = clk::now();
    const uint64_t cap = ((total_bytes / 32 + 65536) + 255) / 256 * 256;
    AsyncBuf surv(cap * 8, stream), valid(cap * 8, stream), counts(GZ_COUNTS * 8, stream);
    hip_check(hipMemsetAsync(counts.p, 0, GZ_COUNTS * 8, stream), "memset");
    hip_check(gz_scan_launch(words, total_bytes, surv.as<uint64_t>(), valid.as<uint64_t>(), counts.as<unsigned long long>(), cap, stream), "gz_scan");
    unsigned long long n_c[3] = {0, 0, 0};
    hip_check(hipMemcpyAsync(n_c, counts.p, 24, hipMemcpyDeviceToHost, stream), "D2H");
    hip_check(hipStreamSynchronize(stream), "sync");
    st.survivors = n_c[0];
    if (n_c[2] > cap / 256 || n_c[1] > cap) {
        for (auto& m : members) if (m.ok) { m.ok = false; m.why = "too many candidate block starts"; }
        return;
    }
    std::vector<uint64_t> bits((size_t)n_c[1]);
    if (!bits.empty()) {
        hip_check(hipMemcpyAsync(bits.data(), valid.p, bits.size() * 8, hipMemcpyDeviceToHost, stream), "D2H");
        hip_check(hipStreamSynchronize(stream), "sync");
    }
    std::sort(bits.begin(), bits.end());
    st.scan_ms = ms_since(t0);

    // ---- pass 1: every candidate inside a member's deflate data, and every member's first block ----
    t0 = clk::now();
    struct Range { uint64_t first_bit, trailer_bit; size_t c0, c1; };
    std::vector<Range> rng(members.size());
    std::vector<GzCand> cands;
    cands.reserve(bits.size() + members.size());
    for (size_t i = 0; i < members.size(); ++i) {
        if (!members[i].ok) continue;
        Range& r = rng[i];
        r.first_bit = (members[i].file_off + hdr[i].deflate_byte) * 8;
        r.trailer_bit = (members[i].file_off + members[i].file_len - 8) * 8;
        r.c0 = cands.size();
        cands.push_back(GzCand{r.first_bit, r.trailer_bit, 0});
        auto it = std::upper_bound(bits.begin(), bits.end(), r.first_bit);
        for (; it != bits.end() && *it < r.trailer_bit; ++it) cands.push_back(GzCand{*it, r.trailer_bit, 0});
        r.c1 = cands.size();
        // a candidate's records go to [its bit, the next candidate's bit) of the record buffer: no run can write into another's
        for (size_t k = r.c0; k < r.c1; ++k) cands[k].rec_cap = (k + 1 < r.c1 ? cands[k + 1].bit : r.trailer_bit) - cands[k].bit;
    }
    st.candidates = cands.size();
    if (cands.size() > 0x7fffffffull) throw err_internal("gunzip: too many candidates");
    AsyncBuf d_cands(cands.size() * sizeof(GzCand), stream), d_res(cands.size() * sizeof(GzRunResult), stream);
    AsyncBuf d_rec((size_t)total_bytes * 8 * 4 + 256, stream);         // one record slot per bit of the buffer (inflate_core.hpp: RecordSink)
    hip_check(hipMemcpyAsync(d_cands.p, cands.data(), cands.size() * sizeof(GzCand), hipMemcpyHostToDevice, stream), "H2D");
    hip_check(gz_pass1_launch(words, d_cands.as<GzCand>(), (uint32_t)cands.size(), d_rec.as<uint32_t>(), d_res.as<GzRunResult>(), stream), "gz_pass1");
    std::vector<GzRunResult> res(cands.size());
    hip_check(hipMemcpyAsync(res.data(), d_res.p, res.size() * sizeof(GzRunResult), hipMemcpyDeviceToHost, stream), "D2H");
    hip_check(hipStreamSynchronize(stream), "sync");
    st.pass1_ms = ms_since(t0);

    // ---- link: the chain of runs of every member; the output layout ----
    t0 = clk::now();
    std::vector<GzRunDesc> runs;
    std::vector<GzMemberDesc> mdesc;
    std::vector<size_t> mindex;                                       // mdesc[j] describes members[mindex[j]]
    std::vector<GzGroupDesc> groups;
    std::vector<GzPiece> tail_pieces, pieces;                         // tails against their group's window; the rest against their run's
    std::vector<GzChunk> chunks;
    std::vector<size_t> chunk0;
    uint64_t total_out = 0;
    for (size_t i = 0; i < members.size(); ++i) {
        GunzipMember& m = members[i];
        if (!m.ok) continue;
        const Range& r = rng[i];
        std::vector<Cand> cs(r.c1 - r.c0);
        for (size_t k = r.c0; k < r.c1; ++k) {
            Cand& c = cs[k - r.c0];
            c.bit = cands[k].bit; c.end_bit = res[k].end_bit; c.out_len = res[k].out_len; c.status = res[k].status; c.n_records = res[k].n_records;
        }
        const std::vector<uint32_t> chain = link_chain(cs, r.first_bit, r.trailer_bit, m.why);
        if (chain.empty()) { m.ok = false; continue; }
        uint64_t len = 0;
        for (uint32_t k : chain) len += cs[k].out_len;
        if ((uint32_t)(len & 0xffffffffu) != hdr[i].want_isize) { m.ok = false; m.why = "inflated length differs from the trailer's"; continue; }
        m.out_off = total_out;
        m.out_len = len;
        m.n_runs = (uint32_t)chain.size();
        GzMemberDesc md;
        md.base = total_out; md.run0 = (uint32_t)runs.size(); md.n_runs = (uint32_t)chain.size();
        // groups of ~sqrt(runs) consecutive runs: the tails take (runs per group) + (groups) dependent steps instead of (runs)
        size_t per_group = 1;
        while (per_group * per_group < chain.size()) ++per_group;
        md.group0 = (uint32_t)groups.size();
        md.n_groups = (uint32_t)((chain.size() + per_group - 1) / per_group);
        std::vector<uint64_t> run_at(chain.size() + 1, 0);
        for (size_t k = 0; k < chain.size(); ++k) run_at[k + 1] = run_at[k] + cs[chain[k]].out_len;
        for (size_t k = 0; k < chain.size(); ++k) {
            const Cand& c = cs[chain[k]];
            const uint64_t at = run_at[k];
            if (k % per_group == 0) {
                GzGroupDesc g;
                g.base = total_out; g.start = at; g.run0 = (uint32_t)runs.size();
                g.n_runs = (uint32_t)std::min(per_group, chain.size() - k);
                groups.push_back(g);
            }
            GzRunDesc d;
            d.bit = c.bit; d.out_off = total_out + at; d.out_len = c.out_len; d.n_records = c.n_records;
            d.first_of_member = k == 0; d.member = (uint32_t)mdesc.size(); d.pad = 0;
            runs.push_back(d);
            GzPiece p;
            p.base = total_out; p.member = (uint32_t)mdesc.size(); p.pad = 0;
            // the run's last 32 KB: one lookup in the 32 KB in front of its GROUP -- but for what gz_tails_b_kernel has made
            // bytes already, the 32 KB in front of the next group
            const size_t g_first = k - k % per_group, g_next = std::min(g_first + per_group, chain.size());
            const uint64_t tail_from = c.out_len > WIN ? at + c.out_len - WIN : at;
            uint64_t tail_to = at + c.out_len;
            if (g_next < chain.size()) tail_to = std::min<uint64_t>(tail_to, run_at[g_next] > WIN ? run_at[g_next] - WIN : 0);
            if (tail_to > tail_from) {
                p.run_start = run_at[g_first]; p.from = tail_from; p.to = tail_to;
                tail_pieces.push_back(p);
            }
            if (c.out_len > WIN)                                      // everything in front of the run's last 32 KB, 64 KB a piece
                for (uint64_t f = at; f < at + c.out_len - WIN; f += 65536) {
                    p.run_start = at; p.from = f; p.to = std::min<uint64_t>(f + 65536, at + c.out_len - WIN);
                    pieces.push_back(p);
                }
        }
        chunk0.push_back(chunks.size());
        for (uint64_t f = 0; f < len; f += 65536) chunks.push_back(GzChunk{total_out + f, (uint32_t)std::min<uint64_t>(65536, len - f), 0});
        mdesc.push_back(md);
        mindex.push_back(i);
        total_out += (len + 255) & ~255ull;
    }
    chunk0.push_back(chunks.size());
    st.runs = runs.size(); st.pieces = pieces.size() + tail_pieces.size(); st.chunks = chunks.size();
    st.link_ms = ms_since(t0);
    if (mdesc.empty()) { if (stats) *stats = st; return; }
    if (runs.size() > 0x7fffffffull || tail_pieces.size() > 0x7fffffffull || pieces.size() > 0x7fffffffull || chunks.size() > 0x7fffffffull) throw err_internal("gunzip: member too large");

    // ---- pass 2, tails, resolve, crc ----
    t0 = clk::now();
    void* out = nullptr;
    hip_check(arena_alloc(&out, total_out + 256, stream), "arena_alloc");
    struct FreeOut { void*& p; hipStream_t st; bool keep = false; ~FreeOut() { if (p && !keep) { arena_free(p, st); p = nullptr; } } } free_out{out, stream};
    {
        AsyncBuf sym(total_out * 2 + 256, stream);
        AsyncBuf d_runs(runs.size() * sizeof(GzRunDesc), stream), d_res2(runs.size() * sizeof(GzRunResult), stream);
        AsyncBuf d_m(mdesc.size() * sizeof(GzMemberDesc), stream), d_err(mdesc.size() * 4 + 4, stream);
        AsyncBuf d_groups(groups.size() * sizeof(GzGroupDesc) + 8, stream), d_tpieces(tail_pieces.size() * sizeof(GzPiece) + 8, stream);
        AsyncBuf d_pieces(pieces.size() * sizeof(GzPiece) + 8, stream), d_chunks(chunks.size() * sizeof(GzChunk) + 8, stream), d_crc(chunks.size() * 4 + 8, stream);
        hip_check(hipMemcpyAsync(d_runs.p, runs.data(), runs.size() * sizeof(GzRunDesc), hipMemcpyHostToDevice, stream), "H2D");
        hip_check(hipMemcpyAsync(d_m.p, mdesc.data(), mdesc.size() * sizeof(GzMemberDesc), hipMemcpyHostToDevice, stream), "H2D");
        if (!pieces.empty()) hip_check(hipMemcpyAsync(d_pieces.p, pieces.data(), pieces.size() * sizeof(GzPiece), hipMemcpyHostToDevice, stream), "H2D");
        if (!tail_pieces.empty()) hip_check(hipMemcpyAsync(d_tpieces.p, tail_pieces.data(), tail_pieces.size() * sizeof(GzPiece), hipMemcpyHostToDevice, stream), "H2D");
        hip_check(hipMemcpyAsync(d_groups.p, groups.data(), groups.size() * sizeof(GzGroupDesc), hipMemcpyHostToDevice, stream), "H2D");
        if (!chunks.empty()) hip_check(hipMemcpyAsync(d_chunks.p, chunks.data(), chunks.size() * sizeof(GzChunk), hipMemcpyHostToDevice, stream), "H2D");
        hip_check(hipMemsetAsync(d_err.p, 0, mdesc.size() * 4 + 4, stream), "memset");
        hip_check(gz_pass2_launch(words, d_rec.as<uint32_t>(), d_runs.as<GzRunDesc>(), (uint32_t)runs.size(), sym.as<uint16_t>(), d_res2.as<GzRunResult>(), stream), "gz_pass2");
        if (stats) { hip_check(hipStreamSynchronize(stream), "sync"); st.pass2_ms = ms_since(t0); t0 = clk::now(); }
        hip_check(gz_tails_launch(sym.as<uint16_t>(), (uint8_t*)out, d_runs.as<GzRunDesc>(), d_groups.as<GzGroupDesc>(), (uint32_t)groups.size(),
                                  d_m.as<GzMemberDesc>(), (uint32_t)mdesc.size(), d_err.as<uint32_t>(), stream), "gz_tails");
        hip_check(gz_resolve_launch(sym.as<uint16_t>(), (uint8_t*)out, d_tpieces.as<GzPiece>(), (uint32_t)tail_pieces.size(), d_err.as<uint32_t>(), stream), "gz_resolve");
        hip_check(gz_resolve_launch(sym.as<uint16_t>(), (uint8_t*)out, d_pieces.as<GzPiece>(), (uint32_t)pieces.size(), d_err.as<uint32_t>(), stream), "gz_resolve");
        hip_check(gz_crc_launch((const uint8_t*)out, d_chunks.as<GzChunk>(), (uint32_t)chunks.size(), d_crc.as<uint32_t>(), stream), "gz_crc");
        std::vector<GzRunResult> res2(runs.size());
        std::vector<uint32_t> err(mdesc.size()), crc(chunks.size());
        hip_check(hipMemcpyAsync(res2.data(), d_res2.p, res2.size() * sizeof(GzRunResult), hipMemcpyDeviceToHost, stream), "D2H");
        hip_check(hipMemcpyAsync(err.data(), d_err.p, err.size() * 4, hipMemcpyDeviceToHost, stream), "D2H");
        if (!crc.empty()) hip_check(hipMemcpyAsync(crc.data(), d_crc.p, crc.size() * 4, hipMemcpyDeviceToHost, stream), "D2H");
        AsyncBuf d_first(mdesc.size() + 8, stream);
        std::vector<uint8_t> first(mdesc.size());
        hip_check(gz_first_bytes_launch((const uint8_t*)out, d_m.as<GzMemberDesc>(), (uint32_t)mdesc.size(), d_first.as<uint8_t>(), stream), "gz_first_bytes");
        hip_check(hipMemcpyAsync(first.data(), d_first.p, first.size(), hipMemcpyDeviceToHost, stream), "D2H");
        hip_check(hipStreamSynchronize(stream), "sync");
        for (size_t j = 0; j < mdesc.size(); ++j) members[mindex[j]].first_byte = members[mindex[j]].out_len ? first[j] : 0;
        // ---- the checks ----
        const uint32_t x64k = crc_xpow8(65536);
        for (size_t j = 0; j < mdesc.size(); ++j) {
            GunzipMember& m = members[mindex[j]];
            const GzMemberDesc& md = mdesc[j];
            for (uint32_t k = 0; k < md.n_runs && m.ok; ++k) {
                const GzRunDesc& d = runs[md.run0 + k];
                const GzRunResult& r2 = res2[md.run0 + k];
                if ((r2.status != RUN_OK && r2.status != RUN_FINAL) || r2.out_len != d.out_len) {
                    m.ok = false;
                    m.why = "the second pass over the run at bit " + std::to_string(d.bit) + " disagrees with the first (status " + std::to_string(r2.status) + ")";
                }
            }
            if (m.ok && err[j]) { m.ok = false; m.why = "a reference points in front of the member's first byte"; }
            if (m.ok) {
                uint32_t c = 0;
                for (size_t k = chunk0[j]; k < chunk0[j + 1]; ++k)
                    c = chunks[k].len == 65536u ? crc_join(c, crc[k], x64k) : crc_join(c, crc[k], crc_xpow8(chunks[k].len));
                if (c != hdr[mindex[j]].want_crc) { m.ok = false; m.why = "CRC-32 of the inflated bytes differs from the trailer's"; }
            }
        }
    }
    st.finish_ms = ms_since(t0);
    st.total_ms = ms_since(t_start);
    if (stats) *stats = st;
    free_out.keep = true;
    *d_out = out;
}

}  // namespace smg
