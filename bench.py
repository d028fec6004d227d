#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X FracMinHash engine.

Metric (BASELINE.json): Gbase/s sketched, k=31, scaled=1000, DNA, seed 42.
Workload (BASELINE.json configs[1], "C2"): 10 GB of synthetic random DNA per GPU,
1,000 records of 10^7 bases, generated directly in HBM; one
"step" = one full pass of the hot path over that resident batch: the k-mer kernel
(canonicalise + MurmurHash3 + keep h <= max_hash), the device radix sort and the
unique pass, leaving the sorted unique hash vector (the sketch) in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--bases B]

N > 1 is launched by the driver with torch.distributed.run.  Every rank sketches
its own 10 GB slice of the stream (weak scaling); value = (bases sketched by all
ranks) / (max over ranks of the elapsed time).  After the timed region the ranks
also run the two multi-GPU configurations of BASELINE.json through
sourmash_amd.parallel (the same code at every N; N = 1 takes the same functions,
SMG_BENCH_FORCE_COLLECTIVES=1 makes a single rank issue the collectives too):
    extra.compare_c4_dist   10,000 x 10,000 compare, row tiles dealt to the ranks, ONE all-gather
    extra.gather_c5_dist    10^6-hash query vs 100,000 sketches sharded 100,000 / N per rank,
                            one all-gather of candidate rows per batch of rounds

Prints ONE JSON line on rank 0 with the contract fields plus
  roofline:     HBM roofline of the dominant kernel (algorithmic bytes / measured kernel time)
  cpu_baseline: the oracle (CPU restatement of the reference algorithm) on bounded samples,
                one thread and as many threads as this container may run (N = 1 only)
  extra:        secondary metrics, each with its own roofline object
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
HBM_ACHIEVABLE_GBS = 6290.0  # same guide: 6.29 TB/s measured with a float4 copy
LDS_PEAK_GBS = 150_000.0   # same guide, LDS: ~150 TB/s aggregate for ds_read_b64/b128 with every CU streaming
# Counter readings (HBM traffic, SQ counters) are READ from the committed rocprofv3 --pmc summaries under profiles/ through
# profiles/pmcfile.py, which also refuses them (null + note) when the kernel's source files have changed since the
# profile was taken -- bench.py itself cannot run counters (they need their own rocprofv3 passes, tools/prof_r03.sh).
sys.path.insert(0, os.path.join(ROOT, "profiles"))
from pmcfile import PmcFile  # noqa: E402
PMC_FILE = "profiles/r03_pmc.txt"
PMC_GATHER_FILE = "profiles/r03_gather_pmc.txt"
SKETCH_SOURCES = ["sketch.hip", "kmer_core.hpp", "murmur3.hpp"]
GATHER_SOURCES = ["gather.hip", "qindex.hpp"]
PMC_C2_INPUT_BYTES = 9_990_000_999                        # the launch the sketch counters were taken on (default C2 batch)
N_SIMDS = 1024


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bases", type=float, default=1e10, help="bases per GPU (default: the 10 GB of config C2)")
    ap.add_argument("--record-len", type=int, default=10_000_000)
    ap.add_argument("--ksize", type=int, default=31)
    ap.add_argument("--scaled", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-compare", action="store_true", help="skip every secondary metric (compare / gather)")
    ap.add_argument("--no-xl", action="store_true", help="skip the XL multi-GPU configurations (compare_xl_dist, gather_xl_dist)")
    ap.add_argument("--cpu-sample", type=float, default=0.0, help="bases for the N-thread CPU sketch leg (0 = auto)")
    return ap.parse_args()


def main():
    args = parse()
    # stdout carries ONE line, the JSON of rank 0: everything else any library writes to file descriptor 1 while the bench runs
    # (RCCL prints a version banner through C stdio) goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    import numpy as np

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the product path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # SMG_BENCH_FORCE_COLLECTIVES=1 exercises the RCCL code path even with a single rank (1-GPU test boxes)
    use_dist = world > 1 or os.environ.get("SMG_BENCH_FORCE_COLLECTIVES") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import sourmash_amd as sm  # noqa: F401
    from sourmash_amd import device as smd, parallel
    from sourmash_amd.synth import synth_sketches, synth_gather, synth_gather_device

    # the interpreter's cyclic GC stays out of the timed regions: a full pass over torch's object graph costs 40-70 ms of
    # host time wherever an allocation happens to trigger it (it showed up as "loop time" of a 42 ms gather,
    # profiles/r02_gather_host_variance.txt); nothing below creates reference cycles worth collecting
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()

    n_bases = int(args.bases)
    rec = args.record_len
    # per-rank slice of one global stream, aligned to whole records (record = rec bases + 1 separator)
    stride = rec + 1
    n_bytes = (n_bases // stride) * stride if n_bases >= stride else n_bases
    start = rank * n_bytes
    seq = smd.synth_dna(n_bytes, seed=42, record_len=rec, start=start, device=dev)
    torch.cuda.synchronize()
    bases_per_step = n_bytes - n_bytes // stride          # separators are not bases

    sk = smd.DeviceSketcher(ksize=args.ksize, scaled=args.scaled, seed=42, device=dev)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if not use_dist:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    hashes = None
    for _ in range(args.warmup):
        hashes = sk.sketch(seq)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        hashes = sk.sketch(seq)
    barrier()
    elapsed = time.perf_counter() - t0

    # one exchange: every rank ends up with the sketch of the whole input (not in the timed region of
    # the per-step metric; it is one 10 MB all-gather per job, reported separately)
    n_unique_local = int(hashes.numel())
    gather_ms = None
    n_unique_total = n_unique_local
    if use_dist:
        tg = time.perf_counter()
        n_unique_total = int(parallel.allgather_union(hashes, force=True).numel())
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg) * 1e3
    elapsed = max_over_ranks(elapsed)

    total_bases = bases_per_step * world * args.steps
    value = total_bases / elapsed / 1e9

    # ---- roofline of the dominant kernel (sketch_dna_kernel): HIP events on the launch stream (rank 0) ----
    roofline = None
    if rank == 0:
        cap = sk.cap
        raw = torch.empty(cap, dtype=torch.int64, device=dev)
        cnt = torch.zeros(2, dtype=torch.int64, device=dev)
        sk.kernel_only(seq, raw, cnt)
        torch.cuda.synchronize()
        reps = max(3, min(args.steps, 10))
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:
            cnt.zero_()
            a.record()
            sk.kernel_only(seq, raw, cnt)
            b.record()
        torch.cuda.synchronize()
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
        kept = int(cnt[0].item())
        alg_bytes = n_bytes + 8 * kept                      # 1 B/base in + 8 B per kept hash out
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "sketch_dna_kernel<31,16>", "achieved": round(achieved, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "kernel_ms": round(kern_ms, 3), "algorithmic_bytes": alg_bytes,
                    **sketch_counters(n_bytes, args.ksize, bases_per_step),
                    "note": "VALU-integer bound (12 x 64-bit multiplies per k-mer), see DESIGN.md; "
                            "kernel-only Gbase/s = %.1f" % (bases_per_step / (kern_ms * 1e-3) / 1e9)}
        del raw, cnt

    # ---- CPU baseline (rank 0, N = 1 only): the oracle on bounded samples of the same workloads ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, seq, n_bytes, sk, np)

    extra = {}
    del seq, hashes                                          # 10 GB back before the matrices
    torch.cuda.empty_cache()
    be = parallel.DeviceBackend(dev)

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    # ---- BASELINE configs C4 / C5 through the multi-GPU drivers (every rank; same code at N = 1) ----
    if not args.no_compare:
        try:
            extra["compare_c4_dist"] = bench_compare_dist(torch, dist, np, dev, be, parallel, smd, synth_sketches, world, rank,
                                                          use_dist, barrier, max_over_ranks)
        except Exception as e:   # the headline metric must still print
            extra["compare_c4_dist"] = {"error": repr(e)}
        torch.cuda.empty_cache()
        try:
            extra["gather_c5_dist"] = bench_gather_dist(torch, np, dev, be, parallel, synth_gather_device, world, rank,
                                                        use_dist, barrier, max_over_ranks)
        except Exception as e:
            extra["gather_c5_dist"] = {"error": repr(e)}
        torch.cuda.empty_cache()

    # ---- XL configurations for the scaling curve (every rank): C4 / C5 are over in milliseconds on one GPU ----
    if not args.no_compare and not args.no_xl:
        from sourmash_amd.synth import synth_sketches_device
        try:
            extra["compare_xl_dist"] = bench_compare_xl_dist(torch, dev, be, parallel, synth_sketches_device, world, rank, use_dist,
                                                             barrier, max_over_ranks)
        except Exception as e:
            extra["compare_xl_dist"] = {"error": repr(e)}
        torch.cuda.empty_cache()
        try:
            extra["gather_xl_dist"] = bench_gather_xl_dist(torch, dev, be, parallel, synth_gather_device, world, rank, use_dist,
                                                           barrier, max_over_ranks)
        except Exception as e:
            extra["gather_xl_dist"] = {"error": repr(e)}
        torch.cuda.empty_cache()

    # ---- single-GPU secondary metrics (N = 1): config C3 and the kernels behind C4 / C5 one by one ----
    if rank == 0 and world == 1 and not args.no_compare:
        try:
            single_gpu_extras(extra, torch, np, dev, be, smd, synth_sketches, synth_gather, synth_gather_device, timed)
        except Exception as e:
            extra["error"] = repr(e)

    if rank == 0:
        extra["arena"] = {**smd.arena_stats(), "what": "the library's device arena (csrc/arena.hpp) over the whole run: driver "
                          "allocator calls, nanoseconds inside them, allocations served from cached blocks"}
    out = None
    if rank == 0:
        out = {
            "metric": "Gbase/s sketched (k=31, scaled=1000)", "value": round(value, 3), "unit": "Gbase/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "C2: sketch 10 GB synthetic random-DNA per GPU (1,000 records x 1e7 bases, "
                                   "ASCII resident in HBM), k=31 scaled=1000 seed=42; kernel + radix sort + unique",
                       "bases_per_gpu": bases_per_step, "bytes_per_gpu": n_bytes, "ksize": args.ksize,
                       "scaled": args.scaled, "unique_hashes_rank0": n_unique_local,
                       "unique_hashes_job": n_unique_total, "allgather_ms": gather_ms,
                       "collectives": "rccl" if use_dist else "none (single rank)"},
            "roofline": roofline, "cpu_baseline": cpu, "extra": extra,
        }
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    # whatever sits in libc's or Python's buffers for descriptor 1 leaves (towards stderr) before stdout comes back
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    if out is not None:
        print(json.dumps(out), flush=True)


def sketch_counters(n_bytes, ksize, bases_per_step):
    """traffic + VALU counters of sketch_dna_kernel<31,16,false> from the committed PMC summary of this same command, or
    nulls with the reason when they cannot be quoted (other input size, no summary, kernel sources changed since)"""
    pmc = PmcFile(PMC_FILE)
    why = None
    if n_bytes != PMC_C2_INPUT_BYTES or ksize != 31:
        why = "counters were taken on the default C2 batch only"
    else:
        why = pmc.stale(SKETCH_SOURCES)
    K = "sketch_dna_kernel<31, 16, false>"
    fetch, write = (pmc.get(K, "FETCH_SIZE"), pmc.get(K, "WRITE_SIZE")) if not why else (None, None)
    if why or fetch is None or write is None:
        return {"traffic": None, "traffic_note": why or f"{PMC_FILE} has no FETCH_SIZE / WRITE_SIZE rows for {K}", "valu": None}
    out = {"traffic": int((2 * fetch + write) * 1024),
           "traffic_from": PMC_FILE + ": 2 x FETCH_SIZE (gfx950 counts half of a 16 B/lane coalesced read, MI355X_MICROARCH.md) + "
                           "WRITE_SIZE, KiB per dispatch, separate --pmc passes of `bench.py --steps 1 --warmup 0 --no-cpu-baseline "
                           "--no-compare`; source hashes of " + ", ".join(SKETCH_SOURCES) + " match the present tree"}
    g = {c: pmc.get(K, c) for c in ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE")}
    if all(v for v in g.values()):
        gui = g["GRBM_GUI_ACTIVE"] / 8                     # summed over the 8 XCDs
        out["valu"] = {"insts_per_kmer": round(g["SQ_INSTS_VALU"] * 64 / bases_per_step, 1),
                       # VALU wave-instructions per SIMD and shader cycle (the ubenchmarked cost of this kernel's mix is 2.4
                       # cycles for and/or/xor/add/shift, 4.3 for multiplies, v_add3, permutes, selects: r01_ubench_valu.txt)
                       "valu_insts_per_simd_cycle": round(g["SQ_INSTS_VALU"] / N_SIMDS / gui, 4),
                       "cycles_per_valu_inst_per_simd": round(N_SIMDS * gui / g["SQ_INSTS_VALU"], 2),
                       "mix_cycles_per_valu_inst": 3.76,
                       "valu_issue_busy_frac": round(3.76 * g["SQ_INSTS_VALU"] / N_SIMDS / gui, 3),
                       "wave_cycles_issuing_frac": round(g["SQ_ACTIVE_INST_ANY"] / g["SQ_WAVE_CYCLES"], 3),
                       "wave_cycles_waiting_to_issue_frac": round(g["SQ_WAIT_INST_ANY"] / g["SQ_WAVE_CYCLES"], 3),
                       "from": PMC_FILE + " (SQ_INSTS_VALU, SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY, GRBM_GUI_ACTIVE)"}
    else:
        out["valu"] = None
    return out


def gather_counters(db_bytes):
    "(build traffic, overlap traffic, note): FETCH_SIZE + WRITE_SIZE per build / per overlap pass from the committed C5 PMC summary"
    pmc = PmcFile(PMC_GATHER_FILE)
    why = pmc.stale(GATHER_SOURCES) if db_bytes == 3997497344 else "counters were taken on config C5 only"
    if why:
        return None, None, why
    build = 0.0
    for k in ("build_bounds_kernel", "build_range_kernel<0>", "build_merge_counts_kernel", "build_partition_kernel", "build_scatter_kernel",
              "build_bounds_table_kernel", "qtable_kernel", "qrec_kernel"):        # (qtable_kernel also runs in front of every overlap pass: averaged per dispatch)
        f, w = pmc.get(k, "FETCH_SIZE"), pmc.get(k, "WRITE_SIZE")
        if f is None or w is None:
            return None, None, f"{PMC_GATHER_FILE} has no rows for {k}"
        build += f + w
    ok = "overlap_wide_kernel" if pmc.get("overlap_wide_kernel", "FETCH_SIZE") is not None else "stream_lookup_kernel<3>"
    fo, wo = pmc.get(ok, "FETCH_SIZE"), pmc.get(ok, "WRITE_SIZE")
    over = None if fo is None or wo is None else int((fo + wo) * 1024)
    return int(build * 1024), over, (PMC_GATHER_FILE + ": FETCH_SIZE + WRITE_SIZE as counted (KiB per dispatch, tools/bench_gather.py under separate "
                                     "--pmc passes); FETCH_SIZE counts half of the bytes of wide coalesced reads on gfx950, so reads of the "
                                     "database proper are under-counted by up to 2x")


def cpu_baseline(args, seq, n_bytes, sk, np):
    """The oracle (kind "port": a C restatement of the reference's algorithm, not the Rust crate) on this box's host
    cores: one thread -- the analogue of single-threaded sourmash (doc/faq.md:307) -- and as many threads as the
    container may run (affinity capped by the cgroup quota; os.cpu_count() is the machine's figure)."""
    import oracle
    from sourmash_amd.synth import synth_sketches, synth_gather
    threads = oracle.usable_cpus()
    # sketch: ~10 s per leg
    s1 = int(min(n_bytes, 2.5e8))
    sn = int(args.cpu_sample) if args.cpu_sample else int(min(n_bytes, 2.5e8 * threads, 4e9))
    host = seq[:max(s1, sn)].cpu().numpy()
    legs = {}
    for name, nthr, size in (("threads_1", 1, s1), ("threads_n", threads, sn)):
        tc = time.perf_counter()
        ref = oracle.sketch_dna_bulk(host[:size], args.ksize, scaled=args.scaled, nthreads=nthr)
        dt = time.perf_counter() - tc
        bases = int((host[:size] != 10).sum())
        got = sk.sketch(seq[:size]).cpu().numpy().view(np.uint64)     # parity spot check of the GPU path on the very same sample
        legs[name] = {"threads": nthr, "Gbase_per_s": round(bases / dt / 1e9, 4),
                      "Mbase_per_s_per_thread": round(bases / dt / 1e6 / nthr, 2), "sample_bases": bases,
                      "seconds": round(dt, 2), "gpu_matches_oracle_on_sample": bool(np.array_equal(got, ref))}
    per1, pern = legs["threads_1"]["Mbase_per_s_per_thread"], legs["threads_n"]["Mbase_per_s_per_thread"]
    oversub = pern < 0.25 * per1
    if oversub:
        print(f"bench.py: CPU baseline is oversubscribed: {pern} Mbase/s/thread with {threads} threads vs {per1} with one; "
              f"reporting the single-thread leg as the baseline value", file=sys.stderr)
    best = legs["threads_1"] if oversub else legs["threads_n"]
    cpu = {"value": best["Gbase_per_s"], "unit": "Gbase/s", "cores": best["threads"], "kind": "port",
           "sample": f"first {best['sample_bases']} bases of the same synthetic stream, oracle.sketch_dna_bulk "
                     f"(OpenMP, contiguous slice per thread), {best['seconds']} s",
           "cpu_model": oracle.cpu_model(), "os_cpu_count": os.cpu_count(), "usable_cpus": threads,
           "oversubscribed": bool(oversub), "sketch": legs,
           "gpu_matches_oracle_on_sample": all(v["gpu_matches_oracle_on_sample"] for v in legs.values()),
           "note": "CPU restatement of the reference algorithm (the Rust crate cannot be built in this image); a "
                   "scalar port, byte-wise canonicalisation + MurmurHash3 per k-mer like signature.rs:246-306"}
    # compare: config C3 (1,000 x 1,000), the reference's loop (every pair, two-pointer walk)
    sk3 = synth_sketches(1000, seed=1234)
    h3, o3 = oracle.make_csr(sk3)
    pairs = 1000 * 999 // 2
    cmp_legs = {}
    for name, nthr in (("threads_1", 1), ("threads_n", threads)):
        tc = time.perf_counter()
        oracle.compare_all_pairs(h3, o3, nthreads=nthr)
        dt = time.perf_counter() - tc
        cmp_legs[name] = {"threads": nthr, "pairs_per_s": round(pairs / dt, 1), "seconds": round(dt, 2)}
    cpu["compare_c3"] = {"pairs": pairs, **cmp_legs, "what": "oracle.compare_all_pairs on config C3 (minhash.rs:915-953 walk per pair)"}
    # gather: scaled-down C5 (2e5-hash query vs 5,000 x ~1,000), the reference's loop (every dataset every round)
    qh, dbh = synth_gather(n_query=200_000, n_db=5000, db_size=1000)
    gh, go = oracle.make_csr(dbh)
    tc = time.perf_counter()
    res = oracle.gather(qh, gh, go, threshold_bp=50_000, scaled=1000, nthreads=threads)
    dt = time.perf_counter() - tc
    cpu["gather_200k_vs_5000"] = {"threads": threads, "rounds": len(res), "seconds": round(dt, 2),
                                  "us_per_round": round(dt * 1e6 / max(len(res), 1), 1),
                                  "what": "oracle.gather (CounterGather walk, index/__init__.py:856-909) on the scaled-down C5 of extra.gather_200k_vs_5000"}
    return cpu


def bench_compare_dist(torch, dist, np, dev, be, parallel, smd, synth_sketches, world, rank, use_dist, barrier, max_over_ranks):
    "config C4 through parallel.compare_all_pairs_distributed: CSR replicated, 16-row tiles dealt to the ranks, ONE all-gather"
    n = 10_000
    if rank == 0:
        big = synth_sketches(n, seed=1234)
        bh, boff = smd.pack_csr(big, device=dev)
        meta = torch.tensor([bh.numel()], dtype=torch.int64, device=dev)
    else:
        meta = torch.zeros(1, dtype=torch.int64, device=dev)
    if world > 1:                                            # the collection reaches the other ranks once (replicated CSR)
        dist.broadcast(meta, 0)
        if rank != 0:
            bh = torch.empty(int(meta.item()), dtype=torch.int64, device=dev)
            boff = torch.empty(n + 1, dtype=torch.int64, device=dev)
        dist.broadcast(bh, 0)
        dist.broadcast(boff, 0)
    pairs = n * (n - 1) // 2
    timing = {}
    parallel.compare_all_pairs_distributed(bh, boff, n, be, force_collectives=use_dist)     # warm: pool, code objects, RCCL buffers
    be._index, be._index_key = None, None                   # the timed call builds its compare index again
    barrier()
    t0 = time.perf_counter()
    full, jac = parallel.compare_all_pairs_distributed(bh, boff, n, be, force_collectives=use_dist, timing=timing)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    checksum = int(full.to(torch.int64).sum().item())
    out = {"ranks": world, "pairs": pairs, "ms": round(dt * 1e3, 2), "pairs_per_s": round(pairs / dt, 1),
           "tiles_ms_rank0": round(timing.get("tiles_ms", 0.0), 2), "allgather_ms": round(timing.get("allgather_ms", 0.0), 2),
           "mirror_and_jaccard_ms": round(timing.get("finish_ms", 0.0), 2), "collective": "all-gather (rccl)" if use_dist else "none",
           "exchange_bytes": int(((n + 15) // 16 + world - 1) // world * world * 16 * n * timing.get("exchange_bytes_per_entry", 4))
           if use_dist else 0,
           "counts_checksum": checksum,
           "note": "wall clock from the resident CSR to the symmetric u32 matrix + f64 Jaccard on every rank: cost model + "
                   "compare-index build + owned row tiles + all-gather + mirror + Jaccard"}
    del full, jac
    return out


def bench_gather_dist(torch, np, dev, be, parallel, synth_gather_device, world, rank, use_dist, barrier, max_over_ranks):
    "config C5 through parallel.gather_distributed: the database sharded by dataset, candidate exchange per batch of rounds"
    nq, ndb, dbsize, thr_bp = 1_000_000, 100_000, 5000, 50_000
    lo, hi = ndb * rank // world, ndb * (rank + 1) // world
    q, gh, goff = synth_gather_device(nq, ndb, dbsize, dev, row_lo=lo, row_hi=hi)
    torch.cuda.synchronize()
    best = None
    for _ in range(2):                                      # second pass: allocator / code objects / RCCL buffers warm
        stats = {}
        barrier()
        t0 = time.perf_counter()
        res = parallel.gather_distributed(q, q.numel(), gh, goff, hi - lo, lo, thr_bp, 1000, be, force_collectives=use_dist,
                                          stats=stats)
        barrier()
        best = max_over_ranks(time.perf_counter() - t0)
    iso = [r[1] for r in res]
    out = {"ranks": world, "datasets": ndb, "datasets_per_rank": hi - lo, "query_hashes": int(q.numel()),
           "db_bytes_per_rank": int(gh.numel() * 8), "rounds": len(res), "total_ms": round(best * 1e3, 2),
           "us_per_round_incl_index_build": round(best * 1e6 / max(len(res), 1), 1),
           "index_build_kernels_ms": stats.get("build_kernels_ms"), "index_build_host_ms": stats.get("build_host_ms"),
           "index_build_driver_alloc_ms": stats.get("build_driver_alloc_ms"), "index_build_driver_allocs": stats.get("build_driver_allocs"),
           "index_build_syncs": stats.get("build_syncs"), "loop_kernels_ms": stats.get("loop_gpu_ms"), "loop_host_ms": stats.get("loop_host_ms"),
           "exchanges": stats.get("exchanges"), "rounds_per_exchange": stats.get("rounds_per_exchange"),
           "records_per_rank": stats.get("records_per_rank"),
           "exchange_bytes_per_rank": (stats["records_per_rank"] * stats["record_words"] * 8) if stats.get("records_per_rank") else None,
           "collective": (stats.get("protocol") or "all-gather of candidate rows (rccl)") if use_dist else "none (native single-shard loop)",
           "overlaps_non_increasing": bool(all(a >= b for a, b in zip(iso, iso[1:]))),
           "first": res[:2], "last": res[-1:] if res else None,
           "note": "wall clock of index build + every round, threshold_bp 50,000; exact parity at this size: "
                   "tests/test_gpu_full_configs.py"}
    return out


def bench_compare_xl_dist(torch, dev, be, parallel, synth_sketches_device, world, rank, use_dist, barrier, max_over_ranks):
    """40,000 x 40,000 compare (8e8 pairs; 16 x C4) through the same distributed driver: the collection is generated in HBM,
    identically on every rank; strong scaling (the matrix is fixed).  Checked through size-independent properties."""
    n = 40_000
    bh, boff = synth_sketches_device(n, dev)
    torch.cuda.synchronize()
    pairs = n * (n - 1) // 2
    timing = {}
    full, jac = parallel.compare_all_pairs_distributed(bh, boff, n, be, force_collectives=use_dist)   # warm
    del full, jac
    be._index, be._index_key = None, None
    barrier()
    t0 = time.perf_counter()
    full, jac = parallel.compare_all_pairs_distributed(bh, boff, n, be, force_collectives=use_dist, timing=timing)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    sizes = (boff[1:] - boff[:-1])
    idx = torch.randint(0, n, (4096,), device=dev)
    jdx = torch.randint(0, n, (4096,), device=dev)
    checks = {"diagonal_equals_sizes": bool((full.diagonal().to(torch.int64) == sizes).all().item()),
              "symmetric_on_4096_samples": bool((full[idx, jdx] == full[jdx, idx]).all().item()),
              "counts_at_most_smaller_sketch": bool((full[idx, jdx].to(torch.int64) <= torch.minimum(sizes[idx], sizes[jdx])).all().item()),
              "jaccard_diagonal_is_one": bool((jac.diagonal() == 1.0).all().item())}
    out = {"ranks": world, "sketches": n, "hashes": int(boff[-1].item()), "pairs": pairs, "ms": round(dt * 1e3, 2),
           "pairs_per_s": round(pairs / dt, 1), "tiles_ms_rank0": round(timing.get("tiles_ms", 0.0), 2),
           "allgather_ms": round(timing.get("allgather_ms", 0.0), 2), "mirror_and_jaccard_ms": round(timing.get("finish_ms", 0.0), 2),
           "collective": "all-gather (rccl)" if use_dist else "none", "scaling": "strong", "checks": checks,
           "counts_checksum": int(full.to(torch.int64).sum().item()),
           "note": "cost model + compare-index build (replicated) + owned 16-row tiles + ONE all-gather of 16-bit counts (3.2 GB in all) "
                   "+ mirror + Jaccard on every rank; what does not shard: the index build and the N x N finish"}
    del full, jac, bh, boff
    return out


def bench_gather_xl_dist(torch, dev, be, parallel, synth_gather_device, world, rank, use_dist, barrier, max_over_ranks):
    """Weak scaling of the gather: every rank holds 125,000 datasets (5 GB; 10^6 datasets = 40 GB on 8 ranks), the 10^6-hash
    query is replicated.  What shards is the index build and the memory; the loop is a chain of dependent rounds."""
    nq, per_rank, dbsize, thr_bp = 1_000_000, 125_000, 5000, 50_000
    ndb = per_rank * world
    lo, hi = per_rank * rank, per_rank * (rank + 1)
    q, gh, goff = synth_gather_device(nq, ndb, dbsize, dev, row_lo=lo, row_hi=hi)
    torch.cuda.synchronize()
    best, stats, res = None, {}, []
    for _ in range(2):
        stats = {}
        barrier()
        t0 = time.perf_counter()
        res = parallel.gather_distributed(q, q.numel(), gh, goff, hi - lo, lo, thr_bp, 1000, be, force_collectives=use_dist, stats=stats)
        barrier()
        best = max_over_ranks(time.perf_counter() - t0)
    iso = [r[1] for r in res]
    return {"ranks": world, "datasets": ndb, "datasets_per_rank": per_rank, "query_hashes": int(q.numel()),
            "db_bytes_per_rank": int(gh.numel() * 8), "db_bytes_total": int(gh.numel() * 8) * world, "rounds": len(res),
            "total_ms": round(best * 1e3, 2), "datasets_per_s": round(ndb / best, 1), "scaling": "weak",
            "index_build_kernels_ms": stats.get("build_kernels_ms"), "loop_kernels_ms": stats.get("loop_gpu_ms"),
            "exchanges": stats.get("exchanges"), "rounds_per_exchange": stats.get("rounds_per_exchange"),
            "collective": (stats.get("protocol") or "all-gather of candidate rows (rccl)") if use_dist else "none (native single-shard loop)",
            "overlaps_non_increasing": bool(all(a >= b for a, b in zip(iso, iso[1:]))),
            "winners_distinct": bool(len({r[0] for r in res}) == len(res)),
            "first": res[:2], "last": res[-1:] if res else None}


def single_gpu_extras(extra, torch, np, dev, be, smd, synth_sketches, synth_gather, synth_gather_device, timed):
    sketches = synth_sketches(1000, seed=1234)
    h, off = smd.pack_csr(sketches, device=dev)
    n = len(sketches)
    pairs = n * (n - 1) // 2
    sizes = (off[1:] - off[:-1]).cpu().numpy().astype(np.int64)
    alg = 8 * int((sizes.sum() * (n - 1)))             # sum over pairs of 8*(n_i+n_j)

    common, jac = smd.compare_rows(h, off)
    ms_merge = timed(lambda: smd.compare_rows(h, off, common=common, jaccard=jac))
    extra["compare_1000x1000_merge"] = {
        "pairs_per_s": round(pairs / (ms_merge * 1e-3), 1), "ms": round(ms_merge, 3), "pairs": pairs,
        "roofline": merge_roofline(alg, ms_merge),
        "kernel": "compare_hash_kernel (LDS hash table per 16 x 32 tile and round; the general path, no index needed)"}
    build_ms = 0.0
    for _ in range(3):                                  # last build: memory pool warm
        idx = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx = smd.BitIndex.build(h, off)
        torch.cuda.synchronize()
        build_ms = (time.perf_counter() - t0) * 1e3
    if idx is not None:
        c2, j2 = smd.compare_rows(h, off, index=idx)
        ms_bits = timed(lambda: smd.compare_rows(h, off, common=c2, jaccard=j2, index=idx))
        extra["compare_1000x1000_bits"] = {
            "pairs_per_s_incl_index_build": round(pairs / ((ms_bits + build_ms) * 1e-3), 1),
            "matrix_ms": round(ms_bits, 3), "index_build_ms": round(build_ms, 3), "universe": idx.universe,
            "identical_to_merge": bool((c2 == common).all().item() and (j2 == jac).all().item()),
            "kernel": "bitmatrix_kernel (hashes held by many sketches as bit columns + popcount, triangle + mirror; "
                      "auto-selected); index built without a sort (csrc/dictindex.hip)"}
        auto_ms = 0.0
        for _ in range(3):                              # what smgpu_compare_all_pairs does: decide, build, compare
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ca, ja = smd.compare_rows(h, off, method="auto")
            torch.cuda.synchronize()
            auto_ms = (time.perf_counter() - t0) * 1e3
        extra["compare_1000x1000_auto"] = {"ms": round(auto_ms, 3), "pairs_per_s": round(pairs / (auto_ms * 1e-3), 1),
                                           "identical_to_merge": bool((ca == common).all().item() and (ja == jac).all().item()),
                                           "note": "one-shot: cost model + index build + matrix + Jaccard, data resident in HBM"}
    del idx
    # gather: 2e5-hash query vs 5,000 x ~1,000-hash database, threshold_bp = 50 kbp
    qh, dbh = synth_gather(n_query=200_000, n_db=5000, db_size=1000)
    gh, goff = smd.pack_csr(dbh, device=dev)
    gq = torch.from_numpy(qh.view(np.int64).copy()).to(dev)
    for _ in range(2):                                  # second pass: allocator / code objects warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        state = be.gather_state(gq, len(qh), gh, goff, len(dbh), 0)      # invert the database against the query
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        state.begin(50, len(dbh))
        res = state.run()                                # every round on the device
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    extra["gather_200k_vs_5000"] = {"rounds": len(res), **{k: v for k, v in state.stats().items() if k in ("build_kernels_ms", "loop_gpu_ms")},
                                    "index_build_ms": round((t1 - t0) * 1e3, 2),
                                    "loop_ms": round((t2 - t1) * 1e3, 2),
                                    "us_per_round": round((t2 - t1) * 1e6 / max(len(res), 1), 1)}
    del state, gh, goff, gq
    # ---- the kernels behind C4 and C5 one by one, each against its roof ----
    big = synth_sketches(10_000, seed=1234)
    bh, boff = smd.pack_csr(big, device=dev)
    bn = len(big)
    bpairs = bn * (bn - 1) // 2
    bsizes = (boff[1:] - boff[:-1]).cpu().numpy().astype(np.int64)
    balg = 8 * int(bsizes.sum() * (bn - 1))
    bc, bj = smd.compare_rows(bh, boff)
    ms_big = timed(lambda: smd.compare_rows(bh, boff, common=bc, jaccard=bj), reps=1)
    auto_big = None
    for _ in range(2):                                  # decide + build the index + matrix + Jaccard, every time; second call
        t0 = time.perf_counter()
        ca, ja = smd.compare_rows(bh, boff, method="auto")
        torch.cuda.synchronize()
        auto_big = (time.perf_counter() - t0) * 1e3
    big_build_ms = big_matrix_ms = None
    for _ in range(2):                                  # the two parts of the auto path on their own (second pass)
        idx_big = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx_big = smd.BitIndex.build(bh, boff)
        torch.cuda.synchronize()
        big_build_ms = (time.perf_counter() - t0) * 1e3
        if idx_big is None:
            break
        t0 = time.perf_counter()
        smd.compare_rows(bh, boff, common=ca, index=idx_big, want_jaccard=False)
        torch.cuda.synchronize()
        big_matrix_ms = (time.perf_counter() - t0) * 1e3
    idx_big = None
    extra["compare_10000x10000"] = {
        "index_build_ms": None if big_build_ms is None else round(big_build_ms, 3),
        "matrix_triangle_and_mirror_ms": None if big_matrix_ms is None else round(big_matrix_ms, 3),
        "pairs": bpairs, "merge_ms": round(ms_big, 2), "merge_pairs_per_s": round(bpairs / (ms_big * 1e-3), 1),
        "merge_roofline": merge_roofline(balg, ms_big),
        "auto_ms": round(auto_big, 2), "auto_pairs_per_s": round(bpairs / (auto_big * 1e-3), 1),
        "identical": bool((ca == bc).all().item() and (ja == bj).all().item()),
        "note": "config C4 (pool-drawn sketches: the cost model picks bit columns); auto includes the index build; "
                "bitmatrix_kernel is VALU-bound: 2.55 instructions per 32-bit AND+popcount against a floor of 2, ~87 % "
                "issue-busy at the measured instruction costs (profiles/r03_compare_pmc.txt), upper triangle + mirror; the "
                "index is built without a sort (csrc/dictindex.hip)"}
    del bc, bj, ca, ja, bh, boff
    torch.cuda.empty_cache()
    gq5, gh5, goff5 = synth_gather_device(1_000_000, 100_000, 5000, dev)
    torch.cuda.synchronize()
    st5 = None
    for _ in range(2):
        st5 = None                                      # the previous index goes back to the pool before the next is built
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st5 = be.gather_state(gq5, gq5.numel(), gh5, goff5, 100_000, 0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        st5.begin(50, 100_000)
        res5 = st5.run()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    postings = int(be.lib.smgpu_gather_postings(st5._ptr))
    st5_stats = st5.stats()
    pmc_build, pmc_overlap, pmc_note = gather_counters(int(gh5.numel() * 8))                             # HIP events around the build's kernels and the loop's rounds
    db_bytes = int(gh5.numel() * 8)
    # index build: every database hash is read once (8 B), one u32 row id is written per posting, the per-element query
    # position (4 B) is written by pass 1 and read by pass 2
    build_alg = db_bytes + 4 * postings + 2 * 4 * int(gh5.numel())
    extra["gather_1M_vs_100000"] = {
        "db_bytes": db_bytes, "postings": postings, "rounds": len(res5), "index_build_ms": round((t1 - t0) * 1e3, 2),
        "loop_ms": round((t2 - t1) * 1e3, 2), "total_ms": round((t2 - t0) * 1e3, 2),
        "us_per_round": round((t2 - t1) * 1e6 / max(len(res5), 1), 2),
        "index_build_kernels_ms": st5_stats["build_kernels_ms"], "index_build_host_ms": st5_stats["build_host_ms"],
        "index_build_driver_alloc_ms": st5_stats["build_driver_alloc_ms"], "index_build_driver_allocs": st5_stats["build_driver_allocs"],
        "index_build_syncs": st5_stats["build_syncs"], "index_build_sync_wait_ms": st5_stats["build_sync_wait_ms"],
        "loop_kernels_ms": st5_stats["loop_gpu_ms"], "loop_host_ms": st5_stats["loop_host_ms"],
        "index_build_roofline": hbm_roofline(build_alg, (t1 - t0) * 1e3,
                                             "8 B per database hash + 4 B per posting + 2 x 4 B query position per element; wall clock of smgpu_gather_new_raw",
                                             traffic=pmc_build, traffic_note=pmc_note, kernels_ms=st5_stats["build_kernels_ms"]),
        "loop_floor_ms": round(postings / 23.0e9 * 1e3, 2),
        "loop_note": "a dependent chain of small kernels (latency, not bandwidth): %d rounds touch %.1f MB of postings in all; "
                     "one 64-bit counter decrement per posting, and the device does 23 G such atomics/s on 100,000 counters "
                     "(profiles/r02_ubench_atomics.txt): loop_floor_ms; the rest is ~7 dependent memory trips per round"
                     % (len(res5), postings * 4 / 1e6)}
    # overlap pass (search / prefetch over the resident collection): |Q ∩ row| for every row
    cnt = be.zeros((100_000,), torch.int64)
    ms_ov = timed(lambda: be.overlaps(gq5, gq5.numel(), gh5, goff5, 100_000, cnt, 0), reps=3)
    extra["overlaps_1M_vs_100000"] = {"ms": round(ms_ov, 3), "sketches_per_s": round(100_000 / (ms_ov * 1e-3), 1),
                                      "roofline": hbm_roofline(db_bytes + 8 * int(gq5.numel()), ms_ov,
                                                               "8 B per database hash + the query once; overlap_wide_kernel (one workgroup per CU, ranges of ~10,000 query hashes in LDS, a wave per row visit)",
                                                               traffic=pmc_overlap, traffic_note=pmc_note)}


def hbm_roofline(alg_bytes, ms, what, traffic=None, traffic_note=None, kernels_ms=None):
    "ms: wall clock of the call; kernels_ms (HIP events around the kernels, when the library reports them) prices `achieved`"
    use = kernels_ms if kernels_ms else ms
    achieved = alg_bytes / (use * 1e-3) / 1e9
    out = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_of_achievable": round(achieved / HBM_ACHIEVABLE_GBS, 4),
           "algorithmic_bytes": int(alg_bytes), "ms": round(use, 3), "wall_ms": round(ms, 3),
           "frac_wall": round(alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic, "what": what}
    if traffic_note:
        out["traffic_note"] = traffic_note
    return out


def merge_roofline(alg_bytes, ms):
    """The general compare path keeps a tile's hashes in LDS and serves 512 pairs from them, so HBM is the wrong roof (the
    collection is L2 / Infinity-Cache resident and the SURVEY.md 8(d) convention figure exceeds the HBM peak).  The
    convention bytes -- 8 B x (n_i + n_j) per pair, what one two-pointer walk per pair would read -- are priced against
    the aggregate LDS read bandwidth; the hash-table kernel does less LDS work than that per pair (one insert or lookup
    per hash of the TILE, not per pair), which is how it passes the walk kernel; its time is shared between VALU issue,
    LDS cycles and the three barriers of a round (profiles/r03_compare_pmc.txt)."""
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    return {"bound": "lds", "achieved": round(achieved, 1), "peak": LDS_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / LDS_PEAK_GBS, 4), "algorithmic_bytes": int(alg_bytes), "ms": round(ms, 3),
            "hbm_convention_frac": round(achieved / HBM_PEAK_GBS, 3),
            "what": "8 B x (n_i + n_j) per pair (SURVEY.md 8d convention) against the aggregate LDS read bandwidth (~150 TB/s, "
                    "MI355X_MICROARCH.md); HBM sees each hash of a tile once per round"}


if __name__ == "__main__":
    main()
