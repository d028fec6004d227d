// sigload.hpp -- a signature collection loaded with the device doing the heavy parts.
//
// collection.hpp does everything on host threads: inflate every .sig.gz (zlib) and turn ~85 KB of decimal text per sketch into
// u64 -- 0.41 s for 10,000 sketches against 6 ms to compare them.  Here, per group of a few thousand documents:
//   host     the members' compressed bytes -> pinned memory (parallel pread) -> HBM
//   device   gunzip.hpp inflates all of them at once; sigjson.hip finds every document's `mins` / `abundances` arrays, parses the
//            `mins` numbers into a value block, checks their order and counts them against the down-sampling threshold, and packs
//            what lies OUTSIDE the arrays (a few hundred bytes a document) for the host
//   host     the unchanged scanner of collection.hpp reads that remainder -- every array replaced by its index -- and so applies
//            the same selection, field checks and error messages as before (signature.rs:569-659, minhash.rs:134-184 semantics)
//   device   the selected arrays' kept prefixes -> rows of the final CSR, which never leaves HBM
// A document the device does not take (a compressed zip member, plain JSON, several gzip members, an array with floats or out
// of order, more than 8 arrays, an empty md5sum) goes through the host path of collection.hpp as before.
#pragma once
#include "collection.hpp"
#include "gunzip.hpp"
#include "sigjson_api.hpp"

namespace smg {

struct SigloadCounters { std::atomic<uint64_t> on_device{0}, on_host{0}; };
inline SigloadCounters& sigload_counters() { static SigloadCounters c; return c; }

struct CollectionLoader::DeviceResult {
    void* d_hashes = nullptr;              // arena block (the caller owns it: arena_free(d_hashes, stream))
    uint64_t total = 0;
    std::vector<uint64_t> offsets;
    std::vector<ManifestRow> rows;
    uint32_t ksize = 0, hash_function = 1;
    uint64_t seed = 42, max_hash = 0, num = 0, skipped = 0;
    uint64_t on_device = 0, on_host = 0;   // documents by the path that parsed them
};

inline void CollectionLoader::run_device(hipStream_t st, DeviceResult& out) {
    const size_t n_items = items_.size();
    std::vector<LoadedPiece> pieces(n_items);
    std::vector<uint8_t> by_host(n_items, 0);
    unsigned nt = n_threads_ ? n_threads_ : std::max(1u, std::thread::hardware_concurrency());
    nt = std::min(nt, 32u);
    auto parallel = [&](size_t n, const std::function<void(size_t)>& fn) {
        std::atomic<size_t> next(0);
        std::vector<Error> errors;
        std::mutex mu;
        auto work = [&] {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= n) return;
                try { fn(i); } catch (const Error& e) { std::lock_guard<std::mutex> g(mu); errors.push_back(e); }
            }
        };
        std::vector<std::thread> th;
        const unsigned k = (unsigned)std::min<size_t>(nt, std::max<size_t>(n, 1));
        for (unsigned t = 1; t < k; ++t) th.emplace_back(work);
        work();
        for (auto& t : th) t.join();
        if (!errors.empty()) throw errors.front();
    };
    auto item_error = [&](size_t i, const Error& e) {
        return Error(e.code, items_[i].path + (items_[i].member.empty() ? "" : ":" + items_[i].member) + ": " + e.what());
    };

    // ---- groups of documents: at most 256 MB of compressed bytes (the inflater keeps 32 B of records per input byte) ----
    struct Group { size_t i0, i1; };
    std::vector<Group> groups;
    std::vector<uint64_t> item_bytes(n_items, 0);
    for (size_t i = 0; i < n_items; ++i) {
        const WorkItem& w = items_[i];
        if (w.zip) { const ZipMember* m = w.zip->find(w.member); item_bytes[i] = m && m->method == 0 ? m->comp_size : 0; }
        else { struct stat sb; item_bytes[i] = stat(w.path.c_str(), &sb) == 0 ? (uint64_t)sb.st_size : 0; }
    }
    // (two groups are under way at a time: see the workers below.  SMG_SIGLOAD_GROUP_BYTES: tests cut small collections into many groups)
    static const uint64_t GROUP_BYTES = [] { const char* e = getenv("SMG_SIGLOAD_GROUP_BYTES"); const long long v = e ? atoll(e) : 0;
                                             return v >= 1024 ? (uint64_t)v : (uint64_t)128 << 20; }();
    for (size_t i = 0; i < n_items;) {
        size_t j = i;
        uint64_t bytes = 0;
        while (j < n_items && j - i < 16384 && (j == i || bytes + item_bytes[j] <= GROUP_BYTES)) bytes += item_bytes[j++] + 8;
        groups.push_back(Group{i, j});
        i = j;
    }
    const uint64_t keep_max = sel_.scaled ? max_hash_for_scaled(sel_.scaled) : ~0ull;
    std::vector<std::unique_ptr<AsyncBuf>> value_blocks(groups.size());
    // (pinning a quarter of a gigabyte costs tens of milliseconds: the staging buffer outlives the call, like the ingest's;
    //  callers hold the context's mutex)
    static PinnedBuf& host_a = *new PinnedBuf();
    static PinnedBuf& host_b = *new PinnedBuf();
    static const bool trace = getenv("SMG_SIGLOAD_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    // one group, start to finish, on the stream and the staging buffer of the worker that took it
    auto do_group = [&](size_t g, hipStream_t st, PinnedBuf& host) {
        const size_t i0 = groups[g].i0, i1 = groups[g].i1, n = i1 - i0;
        const double t_g0 = now();
        double t_read = 0, t_inflate = 0, t_spans = 0, t_parse = 0;
        GunzipStats gstats;
        // the bytes of the group's gzip documents, side by side
        std::vector<GunzipMember> ms(n);
        uint64_t total = 0;
        for (size_t k = 0; k < n; ++k) {
            ms[k].file_off = total;
            ms[k].file_len = item_bytes[i0 + k] >= 26 && item_bytes[i0 + k] <= GROUP_BYTES ? item_bytes[i0 + k] : 0;
            total += (ms[k].file_len + 7) & ~7ull;
        }
        host.reserve((size_t)total + GUNZIP_PAD + 64);
        std::vector<uint8_t> have(n, 0);
        parallel(n, [&](size_t k) {
            if (!ms[k].file_len) return;
            const WorkItem& w = items_[i0 + k];
            uint8_t* dst = host.p + ms[k].file_off;
            try {
                if (w.zip) { if (!w.zip->read_stored_into(*w.zip->find(w.member), dst)) return; }
                else {
                    const int fd = ::open(w.path.c_str(), O_RDONLY);
                    if (fd < 0) return;
                    uint64_t got = 0;
                    while (got < ms[k].file_len) {
                        const ssize_t r = ::pread(fd, dst + got, (size_t)(ms[k].file_len - got), (off_t)got);
                        if (r <= 0) break;
                        got += (uint64_t)r;
                    }
                    ::close(fd);
                    if (got != ms[k].file_len) return;
                }
            } catch (const Error&) { return; }                        // (the host path will report it)
            memset(dst + ms[k].file_len, 0, (size_t)(((ms[k].file_len + 7) & ~7ull) - ms[k].file_len));
            have[k] = dst[0] == 0x1f && dst[1] == 0x8b;
        });
        t_read = now() - t_g0;
        std::vector<size_t> live;                                     // group-relative numbers of the documents on the device path
        std::vector<GunzipMember> gm;
        for (size_t k = 0; k < n; ++k) if (have[k]) { gm.push_back(ms[k]); live.push_back(k); } else by_host[i0 + k] = 1;
        if (gm.empty()) return;
        memset(host.p + total, 0, GUNZIP_PAD);
        AsyncBuf d_files((size_t)total + GUNZIP_PAD + 64, st);
        hip_check(hipMemcpyAsync(d_files.p, host.p, (size_t)total + GUNZIP_PAD, hipMemcpyHostToDevice, st), "H2D");
        void* d_text = nullptr;
        gunzip_device(host.p, d_files.as<uint8_t>(), total, gm, &d_text, st, trace ? &gstats : nullptr);
        t_inflate = now() - t_g0 - t_read;
        struct FreeText { void*& p; hipStream_t st; ~FreeText() { if (p) arena_free(p, st); } } free_text{d_text, st};
        std::vector<SjDoc> docs;
        std::vector<size_t> doc_item;                                 // docs[d] is item i0 + doc_item[d]
        for (size_t q = 0; q < gm.size(); ++q) {
            if (!gm[q].ok || gm[q].out_len == 0) { by_host[i0 + live[q]] = 1; continue; }
            docs.push_back(SjDoc{gm[q].out_off, gm[q].out_len});
            doc_item.push_back(live[q]);
        }
        if (docs.empty()) return;
        // ---- where the arrays are ----
        AsyncBuf d_docs(docs.size() * sizeof(SjDoc), st), d_spans(docs.size() * SJ_MAX_SPANS * sizeof(SjSpan), st), d_flags(docs.size() * 4, st);
        hip_check(hipMemcpyAsync(d_docs.p, docs.data(), docs.size() * sizeof(SjDoc), hipMemcpyHostToDevice, st), "H2D");
        hip_check(sj_spans_launch((const uint8_t*)d_text, d_docs.as<SjDoc>(), (uint32_t)docs.size(), d_spans.as<SjSpan>(), d_flags.as<uint32_t>(), st), "sj_spans");
        std::vector<SjSpan> spans(docs.size() * SJ_MAX_SPANS);
        std::vector<uint32_t> flags(docs.size());
        hip_check(hipMemcpyAsync(spans.data(), d_spans.p, spans.size() * sizeof(SjSpan), hipMemcpyDeviceToHost, st), "D2H");
        hip_check(hipMemcpyAsync(flags.data(), d_flags.p, flags.size() * 4, hipMemcpyDeviceToHost, st), "D2H");
        hip_check(hipStreamSynchronize(st), "sync");
        t_spans = now() - t_g0 - t_read - t_inflate;
        // ---- the numbers of every `mins` array; what lies outside the arrays, packed for the host ----
        std::vector<SjParse> jobs;
        std::vector<SjPiece> rest;                                    // pieces of text outside the arrays
        struct DocPlan { size_t job0 = 0, rest0 = 0, rest1 = 0; uint64_t rest_off = 0; bool take = false; };
        std::vector<DocPlan> plan(docs.size());
        uint64_t n_values = 0, rest_bytes = 0;
        for (size_t d = 0; d < docs.size(); ++d) {
            const uint32_t ns = flags[d] & 0xffu;
            bool odd = (flags[d] & SJ_DOC_ODD) != 0;
            for (uint32_t s = 0; s < ns && !odd; ++s) odd = (spans[d * SJ_MAX_SPANS + s].flags & SJ_SPAN_ODD) != 0;
            if (odd) { by_host[i0 + doc_item[d]] = 1; continue; }
            DocPlan& pl = plan[d];
            pl.take = true;
            pl.job0 = jobs.size();
            pl.rest0 = rest.size();
            pl.rest_off = rest_bytes;
            uint64_t at = 0;
            for (uint32_t s = 0; s < ns; ++s) {
                const SjSpan& sp = spans[d * SJ_MAX_SPANS + s];
                rest.push_back(SjPiece{docs[d].off + at, rest_bytes, sp.begin - at});
                rest_bytes += sp.begin - at;
                at = sp.end;
                if (sp.kind == SJ_MINS) {
                    jobs.push_back(SjParse{docs[d].off + sp.begin, sp.end - sp.begin, n_values, sp.n_values});
                    n_values += sp.n_values;
                }
            }
            rest.push_back(SjPiece{docs[d].off + at, rest_bytes, docs[d].len - at});
            rest_bytes += docs[d].len - at;
            pl.rest1 = rest.size();
        }
        if (jobs.size() > 0x7fffffffull || rest.size() > 0x7fffffffull) throw err_internal("collection group too large");
        value_blocks[g].reset(new AsyncBuf((size_t)n_values * 8 + 256, st));
        AsyncBuf d_jobs(jobs.size() * sizeof(SjParse) + 8, st), d_parsed(jobs.size() * sizeof(SjParsed) + 8, st);
        AsyncBuf d_rest(rest.size() * sizeof(SjPiece) + 8, st), d_rest_bytes((size_t)rest_bytes + 256, st);
        std::vector<SjParsed> parsed(jobs.size());
        std::vector<char> rest_text((size_t)rest_bytes);
        if (!jobs.empty()) {
            hip_check(hipMemcpyAsync(d_jobs.p, jobs.data(), jobs.size() * sizeof(SjParse), hipMemcpyHostToDevice, st), "H2D");
            hip_check(sj_parse_launch((const uint8_t*)d_text, d_jobs.as<SjParse>(), (uint32_t)jobs.size(), value_blocks[g]->as<uint64_t>(),
                                      d_parsed.as<SjParsed>(), keep_max, st), "sj_parse");
            hip_check(hipMemcpyAsync(parsed.data(), d_parsed.p, parsed.size() * sizeof(SjParsed), hipMemcpyDeviceToHost, st), "D2H");
        }
        if (!rest.empty()) {
            hip_check(hipMemcpyAsync(d_rest.p, rest.data(), rest.size() * sizeof(SjPiece), hipMemcpyHostToDevice, st), "H2D");
            hip_check(sj_take_bytes_launch((const uint8_t*)d_text, d_rest.as<SjPiece>(), (uint32_t)rest.size(), d_rest_bytes.as<uint8_t>(), st), "sj_take_bytes");
            if (rest_bytes) hip_check(hipMemcpyAsync(rest_text.data(), d_rest_bytes.p, (size_t)rest_bytes, hipMemcpyDeviceToHost, st), "D2H");
        }
        hip_check(hipStreamSynchronize(st), "sync");
        t_parse = now() - t_g0 - t_read - t_inflate - t_spans;
        // ---- the metadata, by the host's scanner on the remainder (every array replaced by its index) ----
        parallel(docs.size(), [&](size_t d) {
            const DocPlan& pl = plan[d];
            if (!pl.take) return;
            const size_t item = i0 + doc_item[d];
            const uint32_t ns = flags[d] & 0xffu;
            DeviceArray arrays[SJ_MAX_SPANS];
            size_t job = pl.job0;
            std::string text;
            text.reserve((size_t)(rest[pl.rest1 - 1].dst + rest[pl.rest1 - 1].n - pl.rest_off) + 16 * ns);
            for (uint32_t s = 0; s < ns; ++s) {
                const SjSpan& sp = spans[d * SJ_MAX_SPANS + s];
                const SjPiece& pc = rest[pl.rest0 + s];
                text.append(rest_text.data() + pc.dst, (size_t)pc.n);
                text += std::to_string(s);
                DeviceArray& a = arrays[s];
                a.is_mins = sp.kind == SJ_MINS;
                if (a.is_mins) {
                    a.value_off = jobs[job].value_off;
                    a.n_values = sp.n_values;
                    a.n_kept = parsed[job].n_kept;
                    a.odd = (parsed[job].flags & SJ_SPAN_ODD) != 0;
                    ++job;
                }
            }
            const SjPiece& tail = rest[pl.rest1 - 1];
            text.append(rest_text.data() + tail.dst, (size_t)tail.n);
            const WorkItem& w = items_[item];
            LoadedPiece piece;
            try {
                SigScanner sc(text.data(), text.size(), arrays, ns);
                sc.scan(sel_, w.zip ? w.member : w.path, piece);
            } catch (const NeedsHost&) {
                by_host[item] = 1;
                return;
            } catch (const Error&) {
                by_host[item] = 1;                                    // the host parser reports what is wrong with the document itself
                return;
            }
            piece.group = (int)g;
            pieces[item] = std::move(piece);
        });
        if (trace)
            fprintf(stderr, "[sigload] group of %zu documents (%.0f MB): %.1f ms (read %.1f, H2D + inflate %.1f [scan %.1f pass1 %.1f link %.1f pass2 %.1f finish %.1f], "
                            "arrays %.1f, numbers + remainder %.1f, metadata %.1f)\n", n, total / 1e6, now() - t_g0, t_read, t_inflate, gstats.scan_ms, gstats.pass1_ms,
                    gstats.link_ms, gstats.pass2_ms, gstats.finish_ms, t_spans, t_parse, now() - t_g0 - t_read - t_inflate - t_spans - t_parse);
    };
    // Two groups under way at a time (round 6): a group is a chain of device passes with host steps between them -- file reads, the
    // link of the inflater's runs, the plan of the number parser, the metadata scan -- and alone it keeps the device busy about half
    // of its time.  The second worker has a stream and a staging buffer of its own (leaked like the first); groups are dealt from a
    // counter; whatever a worker throws is raised here after both have stopped.
    {
        static hipStream_t st_b = nullptr;
        const bool two = groups.size() >= 2;
        if (two && !st_b) hip_check(hipStreamCreateWithFlags(&st_b, hipStreamNonBlocking), "hipStreamCreate");
        int device = 0;
        (void)hipGetDevice(&device);
        std::atomic<size_t> next_g(0);
        std::mutex err_mu;
        std::vector<Error> errs;
        auto worker = [&](hipStream_t ws, PinnedBuf* wh) {
            (void)hipSetDevice(device);
            try {
                for (;;) {
                    const size_t g = next_g.fetch_add(1);
                    if (g >= groups.size()) break;
                    do_group(g, ws, *wh);
                }
                hip_check(hipStreamSynchronize(ws), "sync");
            } catch (const Error& e) {
                std::lock_guard<std::mutex> lk(err_mu);
                errs.push_back(e);
                next_g.store(groups.size());                          // the other worker finishes its group and stops
            } catch (const std::exception& e) {
                std::lock_guard<std::mutex> lk(err_mu);
                errs.push_back(err_internal(std::string("collection loader: ") + e.what()));
                next_g.store(groups.size());
            }
        };
        std::thread second;
        if (two) second = std::thread(worker, st_b, &host_b);
        worker(st, &host_a);
        if (second.joinable()) second.join();
        if (!errs.empty()) throw errs.front();
    }
    const double t_groups = now();
    // ---- the documents the device did not take: the host path of collection.hpp ----
    std::vector<size_t> host_items;
    for (size_t i = 0; i < n_items; ++i) if (by_host[i]) host_items.push_back(i);
    parallel(host_items.size(), [&](size_t k) {
        const size_t i = host_items[k];
        try {
            const WorkItem& w = items_[i];
            std::string raw = w.zip ? w.zip->read(*w.zip->find(w.member)) : read_whole_file(w.path);
            const std::string text = maybe_gunzip(raw.data(), raw.size());
            SigScanner sc(text.data(), text.size());
            LoadedPiece piece;
            sc.scan(sel_, w.zip ? w.member : w.path, piece);
            pieces[i] = std::move(piece);
        } catch (const Error& e) { throw item_error(i, e); }
    });
    out.on_host = host_items.size();
    out.on_device = n_items - host_items.size();
    // ---- rows in input order: offsets, the one parameter set of a CSR, then the hashes into their rows ----
    uint64_t n_rows = 0;
    for (auto& p : pieces) { n_rows += p.lens.size(); out.skipped += p.skipped; }
    out.skipped += manifest_skipped_;
    out.offsets.reserve(n_rows + 1);
    out.offsets.push_back(0);
    out.rows.reserve(n_rows);
    std::vector<std::vector<SjPiece>> takes(groups.size());
    struct HostRow { const uint64_t* src; uint64_t dst, n; };
    std::vector<HostRow> host_rows;
    bool first = true;
    for (auto& p : pieces) {
        uint64_t host_at = 0;
        for (size_t r = 0; r < p.lens.size(); ++r) {
            const uint64_t dst = out.offsets.back();
            out.offsets.push_back(dst + p.lens[r]);
            if (p.group >= 0) { if (p.lens[r]) takes[p.group].push_back(SjPiece{p.dev_off[r], dst, p.lens[r]}); }
            else { if (p.lens[r]) host_rows.push_back(HostRow{p.hashes.data() + host_at, dst, p.lens[r]}); host_at += p.lens[r]; }
            const ManifestRow& row = p.rows[r];
            const uint32_t hf = molecule_from_name(row.moltype);
            const uint64_t eff_scaled = sel_.scaled ? sel_.scaled : row.scaled;
            if (first) {
                out.ksize = row.ksize; out.hash_function = hf; out.seed = p.seeds[r];
                out.max_hash = max_hash_for_scaled(eff_scaled); out.num = row.num;
                first = false;
            } else {                                                 // one CSR = one parameter set (check_compatible order)
                if (row.ksize != out.ksize) throw Error(E_MISMATCH_KSIZES, "different ksizes cannot be compared");
                if (hf != out.hash_function) throw Error(E_MISMATCH_DNA_PROT, "DNA/prot minhashes cannot be compared");
                if (max_hash_for_scaled(eff_scaled) != out.max_hash) throw Error(E_MISMATCH_SCALED, "mismatch in scaled; comparison fail");
                if (p.seeds[r] != out.seed) throw Error(E_MISMATCH_SEED, "mismatch in seed; comparison fail");
                if (row.num != out.num) throw Error(E_MISMATCH_NUM, "mismatch in num; comparison fail");
            }
        }
        for (auto& row : p.rows) out.rows.push_back(std::move(row));
    }
    out.total = out.offsets.back();
    const double t_rows = now();
    hip_check(arena_alloc(&out.d_hashes, (size_t)out.total * 8 + 16, st), "arena_alloc");
    try {
        for (size_t g = 0; g < groups.size(); ++g) {
            if (takes[g].empty()) continue;
            if (takes[g].size() > 0x7fffffffull) throw err_internal("collection too large");
            AsyncBuf d_takes(takes[g].size() * sizeof(SjPiece), st);
            hip_check(hipMemcpyAsync(d_takes.p, takes[g].data(), takes[g].size() * sizeof(SjPiece), hipMemcpyHostToDevice, st), "H2D");
            hip_check(sj_take_u64_launch(value_blocks[g]->as<uint64_t>(), d_takes.as<SjPiece>(), (uint32_t)takes[g].size(), (uint64_t*)out.d_hashes, st), "sj_take_u64");
            hip_check(hipStreamSynchronize(st), "sync");              // (takes[g] is read by the copy above)
        }
        for (const HostRow& h : host_rows)
            hip_check(hipMemcpyAsync((uint64_t*)out.d_hashes + h.dst, h.src, (size_t)h.n * 8, hipMemcpyHostToDevice, st), "H2D");
        hip_check(hipStreamSynchronize(st), "sync");
    } catch (...) {
        arena_free(out.d_hashes, st);
        out.d_hashes = nullptr;
        throw;
    }
    if (trace) fprintf(stderr, "[sigload] %zu documents by the host parser and the rows' bookkeeping %.1f ms, hashes into their rows %.1f ms\n",
                       host_items.size(), t_rows - t_groups, now() - t_rows);
}

}  // namespace smg
