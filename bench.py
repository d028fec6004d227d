#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X FracMinHash engine.

Metric (BASELINE.json): Gbase/s sketched, k=31, scaled=1000, DNA, seed 42.
Workload (BASELINE.json configs[1], "C2"): 10^10 bases of synthetic random DNA per GPU,
exactly 1,000 records of 10^7 bases (10,000,001,000 bytes with their separators),
generated directly in HBM; one
"step" = one full pass of the hot path over that resident batch: the k-mer kernel
(canonicalise + MurmurHash3 + keep h <= max_hash), the device radix sort and the
unique pass, leaving the sorted unique hash vector (the sketch) in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--records R] [--record-len L]

N > 1 is launched by the driver with torch.distributed.run.  Every rank sketches
its own 1,000-record slice of the stream (weak scaling); value = (bases sketched by all
ranks) / (max over ranks of the elapsed time).  After the timed region the ranks
also run the two multi-GPU configurations of BASELINE.json through
sourmash_amd.parallel (the same code at every N; N = 1 takes the same functions,
SMG_BENCH_FORCE_COLLECTIVES=1 makes a single rank issue the collectives too):
    extra.compare_c4_dist   10,000 x 10,000 compare, row tiles dealt to the ranks, ONE all-gather
    extra.gather_c5_dist    10^6-hash query vs 100,000 sketches sharded 100,000 / N per rank,
                            one all-gather of candidate rows per batch of rounds

Prints ONE JSON line of at most 8 KB on rank 0 (finalize_line) with the contract fields plus
  roofline:     HBM roofline of the dominant kernel (algorithmic bytes / measured kernel time)
  cpu_baseline: the oracle (CPU restatement of the reference algorithm) on bounded samples,
                one thread and as many threads as this container may run (N = 1 only)
  summary:      the headline figure of every secondary metric
The secondary metrics in full (`extra`, each with its own roofline object, ~25 KB) go to
gpurun_out/bench_extra.json and to stderr as one `BENCH_EXTRA {...}` line (write_extras).
"""
import argparse
import json
import os
import sys
import threading
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
from pmcfile import Calibration, PmcFile, ValuMix  # noqa: E402


def _newest(*names):
    "the first of these summaries that exists (a round re-profiles what it changed; untouched kernels keep their older summary)"
    for n in names:
        if os.path.exists(os.path.join(ROOT, n)):
            return n
    return names[-1]


PMC_FILE = _newest("profiles/r06_pmc.txt", "profiles/r05_pmc.txt", "profiles/r04_pmc.txt", "profiles/r03_pmc.txt")
PMC_GATHER_FILE = _newest("profiles/r06_gather_pmc.txt", "profiles/r05_gather_pmc.txt", "profiles/r04_gather_pmc.txt", "profiles/r03_gather_pmc.txt")
PMC_COMPARE_FILE = _newest("profiles/r04_compare_pmc.txt", "profiles/r03_compare_pmc.txt")
COMPARE_BITS_SOURCES = ["bitindex.hip"]
SKETCH_SOURCES = ["sketch.hip", "sketch_kernel.hpp", "kmer_core.hpp", "murmur3.hpp"]
GATHER_SOURCES = ["gather.hip", "gather_build.hip", "overlap.hip", "gather_parts.hpp", "qindex.hpp"]
# the launch the sketch counters were taken on: the default C2 batch of the round that took them (round 6: the literal 1,000 records)
PMC_C2_INPUT_BYTES = 10_000_001_000 if "r06" in PMC_FILE else 9_990_000_999
N_SIMDS = 1024


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--records", type=int, default=1000, help="records per GPU (default: the 1,000 records of config C2, SURVEY.md 8d)")
    ap.add_argument("--record-len", type=int, default=10_000_000, help="bases per record (default 10^7: 1,000 of them = the 10^10 bases of C2)")
    ap.add_argument("--bases", type=float, default=0.0, help="bases per GPU when not a whole number of default records (0 = --records x --record-len)")
    ap.add_argument("--ksize", type=int, default=31)
    ap.add_argument("--scaled", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-compare", action="store_true", help="skip every secondary metric (compare / gather)")
    ap.add_argument("--no-xl", action="store_true", help="skip the XL multi-GPU configurations (compare_xl_dist, gather_xl_dist)")
    ap.add_argument("--no-io", action="store_true", help="skip the file ingest / signature loading metrics (they write ~1.8 GB to a temp directory)")
    ap.add_argument("--cpu-sample", type=float, default=0.0, help="bases for the N-thread CPU sketch leg (0 = auto)")
    ap.add_argument("--extras-timeout", type=float, default=900.0,
                    help="seconds the secondary metrics may take before the line is printed without the unfinished ones and every rank "
                         "leaves (a rank stuck in a collective cannot be interrupted any other way); 0 = no limit")
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
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: launch N > 1 through torch.distributed.run "
              f"(python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus})",
              file=sys.stderr)
        sys.exit(2)
    if os.environ.get("SMG_BENCH_SHARE_GPU") != "1" and torch.cuda.device_count() < world:
        print(f"bench.py: {world} ranks asked for, {torch.cuda.device_count()} GPU(s) visible on this node: one rank per GPU is the "
              f"contract (SMG_BENCH_SHARE_GPU=1 rehearses the launch with all ranks on device 0)", file=sys.stderr)
        sys.exit(2)
    # SMG_BENCH_SHARE_GPU=1: a rehearsal of the N > 1 launch on a box with ONE GPU -- every rank uses device 0 and the process
    # group is gloo (RCCL refuses two ranks on one device; sourmash_amd.parallel stages device tensors through the host around
    # gloo collectives).  Same code path as `--gpus N` otherwise; the numbers it prints mean nothing.
    share_gpu = os.environ.get("SMG_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # SMG_BENCH_FORCE_COLLECTIVES=1 exercises the RCCL code path even with a single rank (1-GPU test boxes)
    use_dist = world > 1 or os.environ.get("SMG_BENCH_FORCE_COLLECTIVES") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # what the process group REALLY is: every `collective` string below is built from these, nothing is assumed
    comm = comm_info(torch, dist, use_dist, dev, share_gpu)

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

    rec = args.record_len
    # per-rank slice of one global stream, aligned to whole records (record = rec bases + 1 separator).  Default = the literal C2 of
    # 1,000 records of 10^7 bases = 10^10 bases = 10,000,001,000 bytes with their separators.
    stride = rec + 1
    if args.bases:
        n_bases = int(args.bases)
        n_bytes = (n_bases // stride) * stride if n_bases >= stride else n_bases
    else:
        n_bytes = args.records * stride
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

    def build_line():
        """the ONE stdout line: contract fields + roofline + cpu_baseline + summary, at most LINE_LIMIT bytes (finalize_line); the
        secondary metrics themselves (`extra`, ~25 KB) go to a side file and to stderr (write_extras)"""
        return {
            "metric": "Gbase/s sketched (k=31, scaled=1000)", "value": round(value, 3), "unit": "Gbase/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "C2: sketch synthetic random DNA resident in HBM as ASCII, per GPU %d records of %d bases (%d bases, %d bytes "
                                   "with their separators), k=%d scaled=%d seed=42; one step = k-mer kernel + radix sort + unique"
                                   % (n_bytes // stride if n_bytes >= stride else 1, rec, bases_per_step, n_bytes, args.ksize, args.scaled),
                       "records_per_gpu": n_bytes // stride if n_bytes >= stride else 1, "record_len": rec,
                       "bases_per_gpu": bases_per_step, "bytes_per_gpu": n_bytes, "ksize": args.ksize,
                       "scaled": args.scaled, "unique_hashes_rank0": n_unique_local,
                       "unique_hashes_job": n_unique_total, "allgather_ms": gather_ms,
                       "collectives": coll(comm, "all-gather of the hash vectors (union)") + (" -- REHEARSAL: ranks share one GPU" if share_gpu else ""),
                       "comm": {k: v for k, v in comm.items() if k != "ranks"}},
            "roofline": roofline, "cpu_baseline": cpu,
            "summary": extras_summary(extra),
            "extra_file": EXTRA_FILE,
        }

    # The headline metric is measured; what follows are secondary metrics, some of them collectives among the ranks.  A rank stuck in
    # one cannot be interrupted from Python, and a job killed from outside prints nothing: after --extras-timeout seconds a timer
    # thread prints the line with whatever is finished (rank 0) and every rank leaves.
    def give_up():
        GIVING_UP.set()
        try:
            if rank == 0:
                extra["timed_out"] = ("the secondary metrics were still running after %.0f s: printed without the unfinished ones, "
                                      "every rank left through os._exit" % args.extras_timeout)
                write_extras(extra, comm)
                os.write(real_stdout, (finalize_line(build_line()) + "\n").encode())
                time.sleep(3.0)                              # the other ranks leave first: none of them sees a peer vanish
        finally:
            os._exit(0)
    watchdog = None
    if args.extras_timeout > 0 and not args.no_compare:
        watchdog = threading.Timer(args.extras_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()

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
                                                          use_dist, barrier, max_over_ranks, comm)
        except Exception as e:   # the headline metric must still print
            extra["compare_c4_dist"] = {"error": repr(e)}
        torch.cuda.empty_cache()
        try:
            extra["gather_c5_dist"] = bench_gather_dist(torch, np, dev, be, parallel, synth_gather_device, world, rank,
                                                        use_dist, barrier, max_over_ranks, comm)
        except Exception as e:
            extra["gather_c5_dist"] = {"error": repr(e)}
        torch.cuda.empty_cache()

    # ---- XL configurations for the scaling curve (every rank): C4 / C5 are over in milliseconds on one GPU ----
    if not args.no_compare and not args.no_xl:
        from sourmash_amd.synth import synth_sketches_device
        try:
            extra["compare_xl_dist"] = bench_compare_xl_dist(torch, dev, be, parallel, synth_sketches_device, world, rank, use_dist,
                                                             barrier, max_over_ranks, comm)
        except Exception as e:
            extra["compare_xl_dist"] = {"error": repr(e)}
        torch.cuda.empty_cache()
        try:
            extra["gather_xl_dist"] = bench_gather_xl_dist(torch, dev, be, parallel, synth_gather_device, world, rank, use_dist,
                                                           barrier, max_over_ranks, comm)
        except Exception as e:
            extra["gather_xl_dist"] = {"error": repr(e)}
        torch.cuda.empty_cache()

    # ---- single-GPU secondary metrics (N = 1): config C3 and the kernels behind C4 / C5 one by one ----
    if rank == 0 and world == 1 and not args.no_compare:
        try:
            single_gpu_extras(extra, torch, np, dev, be, smd, synth_sketches, synth_gather, synth_gather_device, timed)
        except Exception as e:
            extra["error"] = repr(e)

    # ---- the reference-shaped entry points (lists of signature objects in, numpy matrices / result lists out) and the protein
    #      sketch kernels, N = 1 ----
    if rank == 0 and world == 1 and not args.no_compare:
        try:
            api_extras(extra, torch, np, dev, smd, synth_sketches, synth_gather_device)
        except Exception as e:
            extra["api_error"] = repr(e)
        try:
            protein_extras(extra, torch, np, dev, smd, args)
        except Exception as e:
            extra["protein_error"] = repr(e)
        try:
            ksize_extras(extra, torch, np, dev, smd, args)
        except Exception as e:
            extra["ksize_error"] = repr(e)
        torch.cuda.empty_cache()

    # ---- the steps either side of the kernels, on this box (N = 1): file ingest and bulk signature loading ----
    if rank == 0 and world == 1 and not args.no_compare and not args.no_io:
        try:
            io_extras(extra, torch, np, dev, smd)
        except Exception as e:
            extra["io_error"] = repr(e)

    if rank == 0:
        extra["arena"] = {**smd.arena_stats(), "what": "the library's device arena (csrc/arena.hpp) over the whole run: driver "
                          "allocator calls, nanoseconds inside them, allocations served from cached blocks"}
    out = build_line() if rank == 0 else None
    if rank == 0:
        write_extras(extra, comm)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if watchdog is not None:
        watchdog.cancel()
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
        print(finalize_line(out), flush=True)


LINE_LIMIT = 8000          # bytes of the stdout line: logs keep about 8 KB of tail, and a line that does not parse is no measurement
EXTRA_FILE = "gpurun_out/bench_extra.json"


def write_extras(extra, comm):
    """the secondary metrics, each with its roofline object, in full: to EXTRA_FILE (relative to the repo; gpurun_out/ is what a
    GPU box sends back) and as one `BENCH_EXTRA {...}` line on stderr -- never on stdout, which carries the one contract line"""
    doc = {"extra": extra, "comm": comm}
    try:
        text = json.dumps(_strict(doc))
    except Exception as e:                                  # noqa: BLE001
        text = json.dumps({"error": "extras not serialisable: %r" % (e,)})
    try:
        path = os.path.join(ROOT, EXTRA_FILE)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(text + "\n")
    except OSError as e:
        print("bench.py: could not write %s: %r" % (EXTRA_FILE, e), file=sys.stderr)
    print("BENCH_EXTRA " + text, file=sys.stderr, flush=True)


def _strict(v):
    "the same value with NaN / +-Infinity turned into null (strict JSON) and numpy scalars into Python numbers"
    if isinstance(v, dict):
        return {str(k): _strict(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_strict(x) for x in v]
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        return v if v == v and v not in (float("inf"), float("-inf")) else None
    if hasattr(v, "item"):
        return _strict(v.item())
    return str(v)


# what finalize_line drops, in this order, while the line is over the limit: prose first, detail next, never a contract field
_SHED = [("roofline", "traffic_from"), ("roofline", "valu", "mix_from"), ("roofline", "valu", "from"), ("cpu_baseline", "note"),
         ("roofline", "note"), ("cpu_baseline", "gather_200k_vs_5000"), ("cpu_baseline", "compare_c3"), ("cpu_baseline", "sketch"),
         ("config", "comm"), ("summary", "sketch_Gbase_per_s_by_k"), ("summary", "dist_compare_c4_collective"),
         ("summary", "dist_gather_c5_collective"), ("roofline", "valu"), ("config", "collectives"), ("summary",)]
REQUIRED_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline")


def finalize_line(out, limit=LINE_LIMIT):
    """`out` as ONE line of strict JSON of at most `limit` bytes.  Nothing is dropped when it fits (it does: ~5 KB); otherwise the
    entries of _SHED leave one by one and `shed` names them.  tests/test_bench_line_cpu.py pins the size, the strictness and the keys."""
    out = _strict(out)
    missing = [k for k in REQUIRED_KEYS if k not in out]
    if missing:
        raise ValueError("bench line lacks %s" % missing)
    line = json.dumps(out, allow_nan=False, separators=(",", ":"))
    shed = []
    for path in _SHED:
        if len(line.encode()) <= limit:
            break
        d = out
        for k in path[:-1]:
            d = d.get(k) if isinstance(d, dict) else None
        if isinstance(d, dict) and path[-1] in d:
            del d[path[-1]]
            shed.append(".".join(path))
            out["shed"] = shed
            line = json.dumps(out, allow_nan=False, separators=(",", ":"))
    if len(line.encode()) > limit:
        raise ValueError("bench line is %d bytes after shedding %s" % (len(line.encode()), shed))
    return line


GIVING_UP = threading.Event()      # set by the watchdog of main(): from then on an exception means a peer has left, not a failure


def extras_summary(extra):
    """The secondary figures of `extra` once more, in a few hundred bytes at the END of the line (the extras themselves are ~25 KB;
    a log that keeps the last 8 KB of stdout would otherwise lose the C3 / C4 / C5 numbers).  Names say the unit; None = not run."""
    def get(*path):
        v = extra
        for k in path:
            if not isinstance(v, dict) or k not in v:
                return None
            v = v[k]
        return v
    def r(v, nd=3):
        return round(v, nd) if isinstance(v, (int, float)) else v
    bk = get("sketch_by_k") or {}
    return {
        "c3_compare_1000_auto_ms": r(get("compare_1000x1000_auto", "ms")),
        "c4_compare_10000_auto_ms": r(get("compare_10000x10000", "auto_ms")), "c4_auto_pairs_per_s": r(get("compare_10000x10000", "auto_pairs_per_s"), 0),
        "c4_bitmatrix_valu_frac": r(get("compare_10000x10000", "auto_roofline", "frac")),
        "c4_general_kernel_pairs_per_s": r(get("compare_10000x10000", "merge_pairs_per_s"), 0),
        "compare_num_1000_ms": r(get("compare_num_1000x500", "ms")), "compare_abund_1000_ms": r(get("compare_abund_1000x1000", "ms")),
        "compare_abund_pairs_per_s": r(get("compare_abund_1000x1000", "pairs_per_s"), 0),
        "compare_api_10000_objects_ms": r(get("compare_api_10000", "ms")), "compare_api_1000_objects_ms": r(get("compare_api_1000", "ms")),
        "c5_overlap_pass_ms": r(get("overlaps_1M_vs_100000", "ms")), "c5_overlap_hbm_frac": r(get("overlaps_1M_vs_100000", "roofline", "frac")),
        "c5_overlap_traffic_bytes": get("overlaps_1M_vs_100000", "roofline", "traffic"),
        "c5_gather_total_ms": r(get("gather_1M_vs_100000", "total_ms")), "c5_index_build_ms": r(get("gather_1M_vs_100000", "index_build_ms")),
        "c5_index_build_hbm_frac": r(get("gather_1M_vs_100000", "index_build_roofline", "frac")),
        "c5_gather_us_per_round": r(get("gather_1M_vs_100000", "us_per_round")), "c5_gather_rounds": get("gather_1M_vs_100000", "rounds"),
        "c5_loop_frac_of_floor": r(get("gather_1M_vs_100000", "loop_roofline", "frac")),
        "gather_api_c5_objects_ms": r(get("gather_api_c5", "total_ms")),
        "dist_compare_c4_pairs_per_s": r(get("compare_c4_dist", "pairs_per_s"), 0), "dist_compare_c4_collective": get("compare_c4_dist", "collective"),
        "dist_compare_xl_pairs_per_s": r(get("compare_xl_dist", "pairs_per_s"), 0),
        "dist_gather_c5_total_ms": r(get("gather_c5_dist", "total_ms")), "dist_gather_c5_collective": get("gather_c5_dist", "collective"),
        "dist_gather_c5_protocol": get("gather_c5_dist", "protocol"), "dist_gather_c5_fell_back": get("gather_c5_dist", "fell_back"),
        "dist_gather_xl_datasets_per_s": r(get("gather_xl_dist", "datasets_per_s"), 0),
        "protein_k10_G_windows_per_s": r(get("sketch_protein", "protein_k10", "G_windows_per_s")),
        "translate_k10_G_windows_per_s": r(get("sketch_translate", "translate_protein_k10", "G_windows_per_s")),
        "sketch_Gbase_per_s_by_k": {k[1:]: v.get("Gbase_per_s") for k, v in bk.items() if k.startswith("k") and isinstance(v, dict)},
        "ingest_fasta_Gbase_per_s": r(get("ingest_fasta", "Gbase_per_s")), "ingest_gz_Gbase_per_s": r(get("ingest_gz", "Gbase_per_s")),
        "ingest_gz_level6_Gbase_per_s": r(get("ingest_gz_level6", "Gbase_per_s")), "ingest_gz_256_files_Gbase_per_s": r(get("ingest_gz_256_files", "Gbase_per_s")),
        "sigload_10k_signatures_per_s": r(get("sigload_10k", "signatures_per_s"), 0), "sigload_10k_seconds": r(get("sigload_10k", "seconds")),
        "errors": [k for k in extra if k.endswith("error")],
    }


def comm_info(torch, dist, use_dist, dev, share_gpu):
    """backend / world size / library version / device of every rank, as the running process group reports them (gathered once;
    the bench line's `collective` strings say `comm["backend"]`, never a literal)"""
    if not use_dist:
        return {"backend": None, "world_size_observed": 1, "devices": [torch.cuda.get_device_name(dev) + " #" + str(dev.index)],
                "ranks_share_one_gpu": False, "library": None}
    backend = dist.get_backend()
    lib = None
    if backend == "nccl":
        try:
            lib = "RCCL " + ".".join(str(v) for v in torch.cuda.nccl.version())      # (nccl == RCCL on ROCm)
        except Exception as e:                                                   # noqa: BLE001
            lib = "RCCL (version unavailable: %r)" % (e,)
    mine = {"rank": dist.get_rank(), "device_index": dev.index, "device": torch.cuda.get_device_name(dev),
            "uuid": str(getattr(torch.cuda.get_device_properties(dev), "uuid", "")), "pid": os.getpid()}
    everyone = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, mine)
    return {"backend": backend, "world_size_observed": dist.get_world_size(), "library": lib, "ranks": everyone,
            "devices": ["%s #%s" % (r["device"], r["device_index"]) for r in everyone], "ranks_share_one_gpu": bool(share_gpu),
            "distinct_devices": len({(r["uuid"] or r["device_index"]) for r in everyone})}


def coll(comm, what):
    "a `collective` string: what travels, and the backend that moved it as the process group reports it"
    if comm["backend"] is None:
        return "none (single rank)"
    return "%s (%s, %d rank%s)" % (what, comm["library"] or comm["backend"], comm["world_size_observed"], "" if comm["world_size_observed"] == 1 else "s")


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
    cal = Calibration()
    if not cal.ok:
        return {"traffic": None, "traffic_note": f"{cal.path} is absent or its access widths disagree: the counters cannot be turned into bytes", "valu": None}
    out = {"traffic": int(cal.bytes_read(fetch) + cal.bytes_written(write)),
           "traffic_from": PMC_FILE + ": FETCH_SIZE / WRITE_SIZE (KiB per dispatch, separate --pmc passes of `bench.py --steps 1 --warmup 0 "
                           "--no-cpu-baseline --no-compare`), turned into bytes with " + cal.describe() + "; source hashes of " +
                           ", ".join(SKETCH_SOURCES) + " match the present tree"}
    g = {c: pmc.get(K, c) for c in ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE")}
    if all(v for v in g.values()):
        gui = g["GRBM_GUI_ACTIVE"] / 8                     # summed over the 8 XCDs
        mix = ValuMix()
        mix_why = mix.stale()
        cost = None if mix_why else mix.doc["mix_cycles_per_valu_inst"]
        out["valu"] = {"insts_per_kmer": round(g["SQ_INSTS_VALU"] * 64 / bases_per_step, 1),
                       "valu_insts_per_simd_cycle": round(g["SQ_INSTS_VALU"] / N_SIMDS / gui, 4),
                       "cycles_per_valu_inst_per_simd": round(N_SIMDS * gui / g["SQ_INSTS_VALU"], 2),
                       # average issue cost of THIS kernel's instruction mix, from its disassembly and the per-opcode costs of
                       # profiles/r01_ubench_valu.txt (tools/valu_mix.py; refused when the kernel's sources changed since)
                       "mix_cycles_per_valu_inst": cost,
                       "mix_from": mix_why or (mix.path + ": static histogram of the main loop, hot path + wave-conditional blocks weighted by their "
                                               "probability; predicts %.1f VALU instructions per k-mer" % mix.doc["expected_valu_insts_per_kmer"]),
                       "valu_issue_busy_frac": None if cost is None else round(cost * g["SQ_INSTS_VALU"] / N_SIMDS / gui, 3),
                       "wave_cycles_issuing_frac": round(g["SQ_ACTIVE_INST_ANY"] / g["SQ_WAVE_CYCLES"], 3),
                       "wave_cycles_waiting_to_issue_frac": round(g["SQ_WAIT_INST_ANY"] / g["SQ_WAVE_CYCLES"], 3),
                       "from": PMC_FILE + " (SQ_INSTS_VALU, SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY, GRBM_GUI_ACTIVE)"}
    else:
        out["valu"] = None
    return out


def gather_counters(db_bytes):
    "(build traffic, overlap traffic, note): bytes per build / per overlap pass from the committed C5 PMC summary, calibrated"
    pmc = PmcFile(PMC_GATHER_FILE)
    why = pmc.stale(GATHER_SOURCES) if db_bytes == 3997497344 else "counters were taken on config C5 only"
    cal = Calibration()
    if not why and not cal.ok:
        why = f"{cal.path} is absent or its access widths disagree"
    if why:
        return None, None, why
    build = 0.0
    # the builder's kernels on its default path at C5 (pass 1 + 2a = the lean kernel's staging form, overlap_lean_kernel<2>)
    for k in ("range_plan_kernel", "overlap_lean_kernel<2>", "build_count_runs_kernel", "build_merge_counts_kernel", "build_scatter_kernel",
              "build_bounds_table_kernel", "qtable_kernel", "qrec_kernel"):        # (qtable_kernel also runs in front of every overlap pass: averaged per dispatch)
        f, w = pmc.get(k, "FETCH_SIZE"), pmc.get(k, "WRITE_SIZE")
        if f is None or w is None:
            return None, None, f"{PMC_GATHER_FILE} has no rows for {k}"
        build += cal.bytes_read(f) + cal.bytes_written(w)
    ok = next((k for k in ("overlap_lean_kernel<0>", "stream_lookup_kernel") if pmc.get(k, "FETCH_SIZE") is not None), None)
    fo, wo = (pmc.get(ok, "FETCH_SIZE"), pmc.get(ok, "WRITE_SIZE")) if ok else (None, None)
    over = None if fo is None else int(cal.bytes_read(fo) + (cal.bytes_written(wo) if wo is not None else 0))
    return int(build), over, (PMC_GATHER_FILE + ": FETCH_SIZE / WRITE_SIZE (KiB per dispatch, tools/bench_gather.py under separate --pmc passes) turned "
                              "into bytes with " + cal.describe())


def bitmatrix_roofline(n, universe, matrix_ms, counters=True):
    """VALU-issue roofline of the dense compare path (bitmatrix_kernel): the work is one AND + one population count per 32-bit word
    of every pair of the tiles on or above the diagonal; the floor is those two instructions at their measured issue costs
    (v_and_b32 2.43, v_bcnt_u32_b32 4.26 cycles per wave-instruction per SIMD: profiles/r01_ubench_valu.txt) on 1,024 SIMDs."""
    words = (universe + 31) // 32
    nt = (n + 63) // 64
    pair_words = nt * (nt + 1) // 2 * 4096 * words
    clock = 2.4e9
    peak = N_SIMDS * clock / (2.43 + 4.26) * 64
    achieved = pair_words / (matrix_ms * 1e-3)
    out = {"bound": "valu", "kernel": "bitmatrix_kernel", "achieved": round(achieved / 1e12, 3), "peak": round(peak / 1e12, 3),
           "unit": "T(pair x 32-bit word)/s", "frac": round(achieved / peak, 4), "pair_words": int(pair_words), "ms": round(matrix_ms, 3),
           "what": "AND + popcount of one 32-bit word of one pair = 2 VALU instructions at 2.43 + 4.26 cycles per wave-instruction per "
                   "SIMD (profiles/r01_ubench_valu.txt) x 1,024 SIMDs x 2.4 GHz; tiles on or above the diagonal (64 x 64 pairs each); "
                   "the time includes the mirror pass"}
    if not counters:                                        # (the committed counters were taken at C4's shape)
        return out
    pmc = PmcFile(PMC_COMPARE_FILE)
    why = pmc.stale(COMPARE_BITS_SOURCES)
    insts = None if why else pmc.get("bitmatrix_kernel", "SQ_INSTS_VALU")
    if insts:
        out["valu_insts_per_word_step"] = round(insts / (pair_words / 64), 3)
        out["counters_from"] = PMC_COMPARE_FILE + " (SQ_INSTS_VALU per triangle launch at C4)"
    else:
        out["counters_note"] = why or f"{PMC_COMPARE_FILE} has no SQ_INSTS_VALU row for bitmatrix_kernel"
    return out


def loop_roofline(rounds, loop_ms):
    """Latency roofline of the resident gather loop.  A round cannot start before the round in front has finished (the next
    winner depends on every decrement), and it cannot avoid (i) ONE agreement of the whole grid on the winner -- every workgroup's
    record to every workgroup: MI355X_MICROARCH.md prices a 256-workgroup grid agreement at 4.1-4.7 us (barrier-xcd) and an
    8 KB all-gather of tagged granules at 2.4-3.0 us (allgather row); the lower figure is taken -- and (ii) the dependent trips
    behind it: the winner's positions, then the postings of the newly covered hashes (two trips to memory of ~900 cycles each
    once the run bounds sit with the postings; three with a separate bounds table).  Round 4's floor (1.29 us) left the
    agreement out."""
    clock = 2.4e9
    agree_us, trips = 2.4, 2
    floor_us = agree_us + trips * 900 / clock * 1e6
    achieved = rounds / (loop_ms * 1e-3)
    peak = 1e6 / floor_us
    return {"bound": "latency", "kernel": "gather_loop_kernel", "achieved": round(achieved, 1), "peak": round(peak, 1), "unit": "rounds/s",
            "frac": round(achieved / peak, 4), "us_per_round": round(loop_ms * 1e3 / max(rounds, 1), 2), "floor_us_per_round": round(floor_us, 2),
            "what": "per round: one grid-wide agreement among 256 workgroups (2.4 us: the guide's 8 KB granule all-gather, parked) + "
                    "%d dependent trips to memory (~900 cycles each at 2.4 GHz)" % trips}


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


def root_only_compare(torch, parallel, be, bh, boff, n, use_dist, barrier, max_over_ranks, want_checksum, comm):
    """The same compare with the matrices wanted on rank 0 only (result_on="root": the shards are gathered to rank 0, mirror +
    Jaccard run there alone) -- SURVEY.md 8(d)'s "matrix assembled on rank 0".  None without an exchange."""
    if not use_dist:
        return None
    be._index, be._index_key = None, None
    timing = {}
    barrier()
    t0 = time.perf_counter()
    try:
        full, jac = parallel.compare_all_pairs_distributed(bh, boff, n, be, force_collectives=True, timing=timing, result_on="root")
    except Exception as ex:                                  # (an extra: a backend without gather must not cost the bench line)
        barrier()
        return {"error": repr(ex)[:200]}
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    same = None if full is None else bool(int(full.to(torch.int64).sum().item()) == want_checksum)
    del full, jac
    return {"ms": round(dt * 1e3, 2), "pairs_per_s": round(n * (n - 1) // 2 / dt, 1), "collective": coll(comm, "gather of the count shards to rank 0"),
            "gather_ms_rank0": round(timing.get("allgather_ms", 0.0), 2), "mirror_and_jaccard_ms_rank0": round(timing.get("finish_ms", 0.0), 2),
            "same_counts_as_on_every_rank": same}


def bench_compare_dist(torch, dist, np, dev, be, parallel, smd, synth_sketches, world, rank, use_dist, barrier, max_over_ranks, comm):
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
    root = root_only_compare(torch, parallel, be, bh, boff, n, use_dist, barrier, max_over_ranks, checksum if rank == 0 else None, comm)
    out = {"ranks": world, "pairs": pairs, "ms": round(dt * 1e3, 2), "pairs_per_s": round(pairs / dt, 1), "result_on_rank0_only": root,
           "tiles_ms_rank0": round(timing.get("tiles_ms", 0.0), 2), "allgather_ms": round(timing.get("allgather_ms", 0.0), 2),
           "mirror_and_jaccard_ms": round(timing.get("finish_ms", 0.0), 2), "collective": coll(comm, "ONE all-gather of the count shards"),
           "exchange_bytes": int(((n + 15) // 16 + world - 1) // world * world * 16 * n * timing.get("exchange_bytes_per_entry", 4))
           if use_dist else 0,
           "counts_checksum": checksum,
           "note": "wall clock from the resident CSR to the symmetric u32 matrix + f64 Jaccard on every rank: cost model + "
                   "compare-index build + owned row tiles + all-gather + mirror + Jaccard"}
    del full, jac
    return out


def bench_gather_dist(torch, np, dev, be, parallel, synth_gather_device, world, rank, use_dist, barrier, max_over_ranks, comm):
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
           **gather_transport(stats, use_dist, comm),
           "overlaps_non_increasing": bool(all(a >= b for a, b in zip(iso, iso[1:]))),
           "first": res[:2], "last": res[-1:] if res else None,
           "note": "wall clock of index build + every round, threshold_bp 50,000; exact parity at this size: "
                   "tests/test_gpu_full_configs.py"}
    return out


def gather_transport(stats, use_dist, comm):
    """how the ranks' gather loops agreed: `protocol` is what parallel.gather_distributed reports having USED (resident loop kernels
    through hipIpc-mapped device memory / shared host memory, or the candidate-record protocol over the process group's
    all-gather); fell_back says that the resident form was tried and given up (3x slower at C5, still correct)"""
    if not use_dist:
        return {"collective": "none (native single-shard loop)", "protocol": None, "fell_back": False}
    proto = stats.get("protocol")
    fell = bool(stats.get("shared_exchange")) or bool(stats.get("loop_fallbacks"))
    return {"collective": proto or coll(comm, "all-gather of candidate rows per batch of rounds"), "protocol": proto or "candidate records",
            "fell_back": fell, "fell_back_note": stats.get("shared_exchange")}


def bench_compare_xl_dist(torch, dev, be, parallel, synth_sketches_device, world, rank, use_dist, barrier, max_over_ranks, comm):
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
    root = root_only_compare(torch, parallel, be, bh, boff, n, use_dist, barrier, max_over_ranks,
                             int(full.to(torch.int64).sum().item()) if rank == 0 else None, comm)
    out = {"ranks": world, "sketches": n, "hashes": int(boff[-1].item()), "pairs": pairs, "ms": round(dt * 1e3, 2),
           "result_on_rank0_only": root,
           "pairs_per_s": round(pairs / dt, 1), "tiles_ms_rank0": round(timing.get("tiles_ms", 0.0), 2),
           "allgather_ms": round(timing.get("allgather_ms", 0.0), 2), "mirror_and_jaccard_ms": round(timing.get("finish_ms", 0.0), 2),
           "collective": coll(comm, "ONE all-gather of the count shards"), "scaling": "strong", "checks": checks,
           "counts_checksum": int(full.to(torch.int64).sum().item()),
           "note": "cost model + compare-index build (replicated) + owned 16-row tiles + ONE all-gather of 16-bit counts (3.2 GB in all) "
                   "+ mirror + Jaccard on every rank; what does not shard: the index build and the N x N finish"}
    del full, jac, bh, boff
    return out


def bench_gather_xl_dist(torch, dev, be, parallel, synth_gather_device, world, rank, use_dist, barrier, max_over_ranks, comm):
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
            **gather_transport(stats, use_dist, comm),
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
            "roofline": bitmatrix_roofline(n, idx.universe, ms_bits, counters=False),
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
                                           "roofline": dict(bitmatrix_roofline(n, idx.universe, auto_ms, counters=False),
                                                            note="BASELINE config 3 end to end: the WHOLE one-shot call (cost model + index build + "
                                                                 "matrix + Jaccard, host wall clock) priced against the matrix kernel's VALU floor -- "
                                                                 "at this size launches and the index build, not the matrix, are most of the time"),
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
    try:
        compare_ext_extras(extra, torch, np, dev, be, smd, synth_sketches, timed)
    except Exception as e:
        extra["compare_ext_error"] = repr(e)
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
    big_build_ms = big_matrix_ms = big_universe = None
    for _ in range(2):                                  # the two parts of the auto path on their own (second pass)
        idx_big = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx_big = smd.BitIndex.build(bh, boff)
        torch.cuda.synchronize()
        big_build_ms = (time.perf_counter() - t0) * 1e3
        if idx_big is None:
            break
        big_universe = idx_big.universe
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
        "auto_roofline": None if big_matrix_ms is None or big_universe is None else bitmatrix_roofline(bn, big_universe, big_matrix_ms),
        "identical": bool((ca == bc).all().item() and (ja == bj).all().item()),
        "note": "config C4 (pool-drawn sketches: the cost model picks bit columns); auto includes the index build (built without a "
                "sort, csrc/dictindex.hip); auto_roofline prices the matrix part (triangle + mirror) against the VALU issue floor"}
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
    # position (4 B) is written once (the loop's apply step reads the winner's).  Rounds 1-3 counted the positions twice
    # (pass 2 of the builder read them back then; it no longer does): that figure is kept beside the new one.
    build_alg = db_bytes + 4 * postings + 4 * int(gh5.numel())
    build_alg_r03 = build_alg + 4 * int(gh5.numel())
    extra["gather_1M_vs_100000"] = {
        "db_bytes": db_bytes, "postings": postings, "rounds": len(res5), "index_build_ms": round((t1 - t0) * 1e3, 2),
        "loop_ms": round((t2 - t1) * 1e3, 2), "total_ms": round((t2 - t0) * 1e3, 2),
        "us_per_round": round((t2 - t1) * 1e6 / max(len(res5), 1), 2),
        "index_build_kernels_ms": st5_stats["build_kernels_ms"], "index_build_host_ms": st5_stats["build_host_ms"],
        "index_build_driver_alloc_ms": st5_stats["build_driver_alloc_ms"], "index_build_driver_allocs": st5_stats["build_driver_allocs"],
        "index_build_syncs": st5_stats["build_syncs"], "index_build_sync_wait_ms": st5_stats["build_sync_wait_ms"],
        "loop_kernels_ms": st5_stats["loop_gpu_ms"], "loop_host_ms": st5_stats["loop_host_ms"],
        "index_build_roofline": dict(hbm_roofline(build_alg, (t1 - t0) * 1e3,
                                                  "8 B per database hash + 4 B per posting + 4 B query position per element; wall clock of smgpu_gather_new_raw",
                                                  traffic=pmc_build, traffic_note=pmc_note, kernels_ms=st5_stats["build_kernels_ms"]),
                                     frac_r03_convention=round(build_alg_r03 / ((st5_stats["build_kernels_ms"] or (t1 - t0) * 1e3) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                     r03_convention="the positions counted twice (written by pass 1, read by pass 2): %d bytes -- the figure BENCH_r03 quotes 0.154 under" % build_alg_r03),
        "loop_roofline": loop_roofline(len(res5), st5_stats["loop_gpu_ms"] or (t2 - t1) * 1e3),
        "loop_fallbacks": st5_stats.get("loop_fallbacks"),
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
                                                               "8 B per database hash + the query once; overlap_lean_kernel<0> (one workgroup per CU, ranges of ~9,600 query hashes in LDS, a wave per row visit)",
                                                               traffic=pmc_overlap, traffic_note=pmc_note)}


def compare_ext_extras(extra, torch, np, dev, be, smd, synth_sketches, timed):
    """The batched launches of csrc/compare_ext.hip on resident collections (kernel time, HIP events): bottom-k sketches
    (num = 500, the reference's golden compare shape: tests/test_compare.py:48-63) and abundance-weighted similarity on a
    config-C3-shaped collection (compare.py:14-64 with track_abundance sketches; minhash.rs:593-680)."""
    import ctypes as C
    lib, p, st = be.lib, be._p, be._s
    n = 1000
    pairs = n * (n - 1) // 2
    rng = np.random.default_rng(3)
    pool = np.unique(rng.integers(1, 2**62, 20_000, dtype=np.int64).astype(np.uint64))
    rows = [np.sort(rng.choice(pool, size=2000, replace=False))[:500] for _ in range(n)]
    h, off = smd.pack_csr(rows, device=dev)
    nums = torch.full((n,), 500, dtype=torch.int32, device=dev)
    common = torch.zeros((n, n), dtype=torch.int32, device=dev)
    union = torch.zeros((n, n), dtype=torch.int32, device=dev)
    jac = torch.zeros((n, n), dtype=torch.float64, device=dev)
    ms = timed(lambda: be.rustcall(lib.smgpu_compare_num_raw, p(h), p(off), p(nums), n, p(common), p(union), p(jac), st()))
    alg = 8 * 2 * 500 * pairs
    extra["compare_num_1000x500"] = {"pairs": pairs, "ms": round(ms, 3), "pairs_per_s": round(pairs / (ms * 1e-3), 1),
                                     "roofline": merge_roofline(alg, ms),
                                     "kernel": "compare_ext_kernel<num>: 16 x 16 tiles, one lane per pair, the merge walk stopped after num steps "
                                               "(minhash.rs:593-621); parity: tests/test_gpu_compare.py::test_num_all_pairs_batched_vs_oracle"}
    sk = synth_sketches(n, seed=1234)
    h, off = smd.pack_csr(sk, device=dev)
    ab = (h % 7 + 1) * ((h >> 3) % 11 + 1)
    tot = int(off[-1].item())               # the caller of a raw device entry point knows its array sizes
    prod = torch.zeros((n, n), dtype=torch.int64, device=dev)
    sq = torch.zeros((n,), dtype=torch.int64, device=dev)
    sizes = (off[1:] - off[:-1]).cpu().numpy().astype(np.int64)
    alg = 2 * 8 * int(sizes.sum() * (n - 1))                 # hashes and abundances of both sketches of every pair
    ms = timed(lambda: be.rustcall(lib.smgpu_compare_abund_raw_n, p(h), p(ab), p(off), n, tot, True, p(common), p(prod), p(sq), st()))
    extra["compare_abund_1000x1000"] = {"pairs": pairs, "ms": round(ms, 3), "pairs_per_s": round(pairs / (ms * 1e-3), 1),
                                        "roofline": merge_roofline(alg, ms),
                                        "kernel": "csrc/abund_pairs.hip: per-block hash-sorted lists (ap_slice_kernel: the sorted rows merged slice by slice in LDS, "
                                                  "no sort) joined tile by tile (ap_join_kernel, minhash.rs:635-680); parity: "
                                                  "tests/test_gpu_compare.py::test_angular_all_pairs_batched_vs_oracle"}


PCIE_GBS = 57.0            # H2D rate measured on this class of box (tests/probe_host.py); the bound of every host-pointer entry point


def _xfer(lib, reset=False):
    import ctypes as C
    out = (C.c_uint64 * 5)()
    lib.smgpu_xfer_stats(out, reset)
    return {"h2d_bytes": int(out[0]), "d2h_bytes": int(out[1]), "h2d_ms": round(out[2] / 1e6, 2), "d2h_ms": round(out[3] / 1e6, 2)}


def api_extras(extra, torch, np, dev, smd, synth_sketches, synth_gather_device):
    """The entry points the reference's callers use, wall clock: compare_all_pairs over a LIST OF SourmashSignature objects -> numpy
    matrix (compare.py:326-358), and a gather over a collection built from sketch objects (SketchSet -> CounterGather's device
    counter, index/__init__.py:783-909).  Beside each: the bytes that crossed PCIe and the time that alone costs at the measured
    H2D rate, and the device-resident kernel time of the same work from the lines above -- the API wall clock should be about
    kernel + transfer."""
    import sourmash_amd as sm
    from sourmash_amd._lowlevel import lib
    from sourmash_amd.compare import compare_all_pairs
    from sourmash_amd.index import SketchSet
    for n, resident in ((1000, "compare_1000x1000_auto"), (10_000, "compare_10000x10000")):
        sk = synth_sketches(n, seed=1234)
        sigs = []
        for i, a in enumerate(sk):
            mh = sm.MinHash(0, 31, scaled=1000)
            mh.add_many(a)
            sigs.append(sm.SourmashSignature(mh, name="s%d" % i))
        h, off = smd.pack_csr(sk, device=dev)
        want = smd.compare_rows(h, off, method="auto")[1]
        best, stats = None, None
        for _ in range(3):
            _xfer(lib, reset=True)
            t0 = time.perf_counter()
            m = compare_all_pairs(sigs, True)
            dt = time.perf_counter() - t0
            if best is None or dt < best:
                best, stats = dt, _xfer(lib)
        same = bool(np.array_equal(m.view(np.uint64), want.cpu().numpy().view(np.uint64)))
        r = extra.get(resident, {})
        kernel_ms = r.get("auto_ms", r.get("ms"))
        moved = stats["h2d_bytes"] + stats["d2h_bytes"]
        pcie_ms = moved / (PCIE_GBS * 1e9) * 1e3
        pairs = n * (n - 1) // 2
        extra["compare_api_%d" % n] = {
            "what": "sourmash_amd.compare.compare_all_pairs(list of %d SourmashSignature objects, ignore_abundance=True) -> f64 numpy matrix, "
                    "wall clock of the call (best of 3)" % n,
            "ms": round(best * 1e3, 2), "pairs_per_s": round(pairs / best, 1), **stats,
            "pcie_bound_ms": round(pcie_ms, 2), "resident_kernel_ms": kernel_ms,
            "kernel_plus_transfer_ms": None if kernel_ms is None else round(kernel_ms + pcie_ms, 2),
            "wall_over_kernel_plus_transfer": None if kernel_ms is None else round(best * 1e3 / (kernel_ms + pcie_ms), 3),
            "bit_identical_to_the_resident_path": same}
        del sigs, sk, m, want, h, off
    # gather: 100,000 sketch objects -> collection in HBM -> every round of the min-set-cover
    nq, ndb, dbsize, thr_bp = 1_000_000, 100_000, 5000, 50_000
    q, gh, goff = synth_gather_device(nq, ndb, dbsize, dev)
    hh, oo = gh.cpu().numpy().view(np.uint64), goff.cpu().numpy()
    del gh, goff
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    mhs = []
    for d in range(ndb):
        mh = sm.MinHash(0, 31, scaled=1000)
        mh.add_many(hh[oo[d]:oo[d + 1]])
        mhs.append(mh)
    qmh = sm.MinHash(0, 31, scaled=1000)
    qmh.add_many(q.cpu().numpy().view(np.uint64))
    objects_s = time.perf_counter() - t0
    db_bytes = int(hh.nbytes)
    del hh
    best = None
    for _ in range(2):
        _xfer(lib, reset=True)
        t0 = time.perf_counter()
        ss = SketchSet(mhs)
        t1 = time.perf_counter()
        res = ss.gather(qmh, threshold_bp=thr_bp)
        t2 = time.perf_counter()
        if best is None or t2 - t0 < best[0]:
            best = (t2 - t0, t1 - t0, t2 - t1, _xfer(lib))
        del ss
    r = extra.get("gather_1M_vs_100000", {})
    kernel_ms = None if "total_ms" not in r else r["total_ms"]     # index build + every round on resident inputs (the line above)
    pcie_ms = db_bytes / (PCIE_GBS * 1e9) * 1e3
    extra["gather_api_c5"] = {
        "what": "SketchSet(list of 100,000 MinHash objects) -> .gather(query MinHash of 1e6 hashes, threshold_bp=50,000): wall clock of "
                "packing + upload, then counter + index build + every round + the result list (best of 2); the objects themselves "
                "took objects_s to create and are not timed",
        "total_ms": round(best[0] * 1e3, 2), "pack_and_upload_ms": round(best[1] * 1e3, 2), "gather_ms": round(best[2] * 1e3, 2),
        "rounds": len(res), **best[3], "db_bytes": db_bytes, "pcie_bound_ms": round(pcie_ms, 2), "resident_kernel_ms": kernel_ms,
        "kernel_plus_transfer_ms": None if kernel_ms is None else round(kernel_ms + pcie_ms, 2),
        "wall_over_kernel_plus_transfer": None if kernel_ms is None else round(best[0] * 1e3 / (kernel_ms + pcie_ms), 3),
        "same_rounds_as_the_resident_path": None if "rounds" not in r else bool(r["rounds"] == len(res)), "objects_s": round(objects_s, 1)}


def ksize_extras(extra, torch, np, dev, smd, args):
    """The sketch step at other ksizes (signature.rs:246-306 treats every k alike): kernel + sort + unique on 10^9 resident bases,
    scaled = 1000 -- the unrolled register-window kernel to k = 88, the run-time-k kernel of sketch_words.hip beyond -- with the CPU
    restatement timed beside each on a bounded sample of the same input (and the GPU's hashes of that sample compared with its)."""
    if not args.no_cpu_baseline:
        import oracle
    n = 1_000_000_000
    seq = smd.synth_dna(n, seed=44, record_len=10_000_000, device=dev)
    sample = 400_000
    host = bytes(seq[:sample].cpu().numpy())
    out = {}
    for k in (21, 51, 88, 89, 128, 200, 256, 1000):
        sk = smd.DeviceSketcher(k, 1000)
        cpu_port = None
        if not args.no_cpu_baseline:
            got = np.sort(sk.sketch(seq[:sample]).cpu().numpy().view(np.uint64))
            t0 = time.perf_counter()
            want = oracle.sketch_dna_bulk(host, k, scaled=1000, nthreads=4)  # the CPU restatement on a bounded sample: timed, and the checker
            cpu_s = time.perf_counter() - t0
            cpu_port = {"sample_bases": sample, "seconds": round(cpu_s, 3), "threads": 4, "Mbase_per_s": round(sample / cpu_s / 1e6, 1),
                        "gpu_matches_oracle_on_sample": bool(np.array_equal(got, want))}
        sk.sketch(seq)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
        for a, b in evs:
            a.record()
            h = sk.sketch(seq)
            b.record()
        torch.cuda.synchronize()
        ms = float(np.median([a.elapsed_time(b) for a, b in evs]))
        blocks = k // 16 + (1 if k % 16 else 0)
        # what binds these kernels is VALU issue, not HBM: instructions per k-mer as counted (profiles/r05_long_k_pmc.txt: k = 88, 128,
        # 200; profiles/r06_pmc.txt: k = 31) or, for the run-time-k kernel at other k, from its two counted points (41.1 per 16 bytes
        # of key), priced at the 3.74-cycle instruction mix of profiles/valu_mix_sketch.json over the step's time
        counted = {88: 264.6, 128: 414.2, 200: 599.2}
        insts = counted.get(k) if k in counted else (414.2 + 41.1 * (k - 128) / 16.0 if k >= 89 else None)
        valu = None
        if insts is not None:
            valu = {"insts_per_kmer": round(insts, 1), "counted": k in counted,
                    "issue_busy_frac": round(3.74 * insts * (n / 64.0) / (ms * 1e-3 * 2.4e9 * N_SIMDS), 3),
                    "from": "profiles/r05_long_k_pmc.txt (SQ_INSTS_VALU per dispatch), over the whole step's time (kernel + sort + unique)"}
        out["k%d" % k] = {"ms": round(ms, 3), "Gbase_per_s": round(n / (ms * 1e-3) / 1e9, 1), "hashes": int(h.numel()),
                          "kernel": "register window (unrolled)" if k <= 88 else "run-time k (sketch_words.hip)",
                          "cpu_port": cpu_port, "valu": valu,
                          "roofline": hbm_roofline(n + 8 * int(h.numel()), ms, "1 B per base + 8 B per kept hash; whole step (kernel + sort + unique); "
                                                   "the kernels are bound by instruction issue: MurmurHash3 is 4 64-bit multiplies per 16-byte "
                                                   "block of the key, %d blocks at this k" % blocks)}
    extra["sketch_by_k"] = {"bases": n, "scaled": 1000, **out}


def protein_extras(extra, torch, np, dev, smd, args):
    """Protein / dayhoff / hp sketches (signature.rs:307-393, encodings.rs:103-368): the kernels on resident
    input -- residues of a protein sequence, and DNA translated in six frames -- in residue windows hashed per second, with the
    CPU restatement timed beside them on a bounded sample."""
    import ctypes as C
    import oracle
    from sourmash_amd._lowlevel import lib
    from sourmash_amd.utils import rustcall
    from sourmash_amd.minhash import _get_max_hash_for_scaled
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {}
    n = 1_000_000_000
    aas = torch.tensor(list(b"ACDEFGHIKLMNPQRSTVWY"), dtype=torch.uint8, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    prot = aas[torch.randint(0, 20, (n,), device=dev, generator=gen)]
    dna = smd.synth_dna(n, seed=43, record_len=n + 1, device=dev)
    aa = torch.empty(2 * n + 16, dtype=torch.uint8, device=dev)
    mh200 = _get_max_hash_for_scaled(200)
    cap = int(2 * n / 200 * 1.3) + 65536
    hashes = torch.empty(cap, dtype=torch.int64, device=dev)
    cnt = torch.zeros(2, dtype=torch.int64, device=dev)
    for name, src, hf, k_aa, translate in (("protein_k10", prot, 2, 10, False), ("dayhoff_k16", prot, 3, 16, False), ("hp_k42", prot, 4, 42, False),
                                           ("translate_protein_k10", dna, 2, 10, True)):
        def run():
            cnt.zero_()
            return rustcall(lib.smgpu_sketch_residues_kernels_raw, p(src), n, k_aa, hf, 42, mh200, translate, p(aa), aa.numel(), p(hashes), cap, p(cnt), st)
        n_aa = run()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
        for a, b in evs:
            a.record()
            run()
            b.record()
        torch.cuda.synchronize()
        ms = float(np.median([a.elapsed_time(b) for a, b in evs]))
        kept = int(cnt[0].item())
        # CPU restatement on a bounded sample of the same input, and parity of the kept hashes on it
        sample = 2_000_000 if not translate else 1_000_000
        host = bytes(src[:sample].cpu().numpy())
        moltype = {2: "protein", 3: "dayhoff", 4: "hp"}[hf]
        t0 = time.perf_counter()
        ref = oracle.seq_to_hashes_protein(host, k_aa, moltype, is_protein=not translate)      # (the oracle's function calls itself twice: count, then fill)
        cpu_s = (time.perf_counter() - t0) / 2
        want = np.sort(ref[(ref >= 1) & (ref <= np.uint64(mh200))])
        cnt.zero_()
        rustcall(lib.smgpu_sketch_residues_kernels_raw, p(src), sample, k_aa, hf, 42, mh200, translate, p(aa), aa.numel(), p(hashes), cap, p(cnt), st)
        torch.cuda.synchronize()
        got = np.sort(hashes[:int(cnt[0].item())].cpu().numpy().view(np.uint64))
        windows = int(n_aa) - (6 if translate else 0)
        out[name] = {"input_bytes": n, "residues": int(n_aa), "ms": round(ms, 3), "G_windows_per_s": round(windows / (ms * 1e-3) / 1e9, 2),
                     "input_Gbytes_per_s": round(n / (ms * 1e-3) / 1e9, 2), "kept_at_scaled_200": kept,
                     "roofline": hbm_roofline(n + int(n_aa) * 2 + 8 * kept, ms,
                                              "input once + residues written and read once (1 B each) + 8 B per kept hash; residues / translate kernel + window kernel"),
                     "cpu_port": {"sample_bytes": sample, "seconds": round(cpu_s, 2), "threads": 1, "M_windows_per_s": round(len(ref) / cpu_s / 1e6, 2),
                                  "gpu_matches_oracle_on_sample": bool(np.array_equal(got, want))}}
    extra["sketch_protein"] = {k: v for k, v in out.items() if not k.startswith("translate")}
    extra["sketch_translate"] = {k: v for k, v in out.items() if k.startswith("translate")}


def io_extras(extra, torch, np, dev, smd):
    """SURVEY.md 8(f) ranks 1-2 measured on THIS box inside the driver's run: a FASTA file -> sketch through the native ingest
    (command_sketch.py:697,746-768: the reference reads through screed, record by record), the same file as ONE gzip member
    (many-thread inflate, csrc/pargz.hpp), and a zip of 10,000 signatures -> CSR in HBM (signature.rs:569-659 per file in the
    reference).  Each against the bound it sits under.  Files go to a temporary directory and are removed."""
    import gzip
    import hashlib
    import io
    import shutil
    import tempfile
    import zipfile
    import zlib
    from sourmash_amd import index
    from sourmash_amd.sketch import sketch_file
    from sourmash_amd.synth import splitmix64, MAX_HASH_1000
    tmp = tempfile.mkdtemp(prefix="smg_bench_")
    try:
        # ---- 1 GB FASTA (100 records of 1e7 bases, 80-column lines) ----
        n, rec = 1_000_000_000, 10_000_000
        seq = smd.synth_dna(n + n // rec, seed=42, record_len=rec, device=dev).cpu().numpy()
        path = os.path.join(tmp, "synth.fa")
        bases = 0
        with open(path, "wb") as fh:
            for i, r in enumerate(bytes(seq).split(b"\n")):
                a = np.frombuffer(r, dtype=np.uint8)
                bases += len(a)
                fh.write(b">synth_%d\n" % i)
                full = (len(a) // 80) * 80
                if full:
                    fh.write(np.concatenate([a[:full].reshape(-1, 80), np.full((full // 80, 1), 10, dtype=np.uint8)], axis=1).tobytes())
                if len(a) > full:
                    fh.write(a[full:].tobytes() + b"\n")
        del seq
        file_bytes = os.path.getsize(path)
        sketch_file(path, "k=31,scaled=1000")                 # warm: page cache, pinned ring, code objects
        t0 = time.perf_counter()
        sig, = sketch_file(path, "k=31,scaled=1000")
        dt = time.perf_counter() - t0
        plain_md5 = sig.md5sum()
        extra["ingest_fasta"] = {
            "file_bytes": file_bytes, "bases": bases, "seconds": round(dt, 3), "Gbase_per_s": round(bases / dt / 1e9, 2),
            "hashes": len(sig.minhash), "bound": "PCIe H2D of the raw file (~57 GB/s measured on this class of box, tests/probe_host.py) "
                                                  "and the page-cache read; the record structure is resolved on the GPU",
            "frac_of_h2d_57GBps": round(file_bytes / dt / 57e9, 3),
            "what": "smgpu_signature_add_file on a 1 GB FASTA in the page cache -> k=31 scaled=1000 sketch, end to end"}
        # ---- the same bases as ONE gzip member (level 1), first 400 MB of the file: many-thread inflate vs one zlib stream ----
        gz = os.path.join(tmp, "synth_part.fa.gz")
        part_bytes = 0
        co = zlib.compressobj(1, zlib.DEFLATED, 31)
        part = os.path.join(tmp, "synth_part.fa")
        with open(path, "rb") as fi, open(gz, "wb") as fo, open(part, "wb") as fp:
            while part_bytes < 400_000_000:
                block = fi.read(16 << 20)
                if not block:
                    break
                # cut at a line end so that the part is a well-formed FASTA
                block = block[:block.rfind(b"\n") + 1] if part_bytes + len(block) >= 400_000_000 else block
                fo.write(co.compress(block))
                fp.write(block)
                part_bytes += len(block)
            fo.write(co.flush())
        sig_part, = sketch_file(part, "k=31,scaled=1000")
        raw = np.fromfile(part, dtype=np.uint8)
        nl = np.flatnonzero(raw == 10)
        heads = np.flatnonzero(raw == ord(">"))
        part_bases = int(raw.size - nl.size - (nl[np.searchsorted(nl, heads)] - heads).sum())     # minus newlines, minus the header lines' text
        del raw
        sketch_file(gz, "k=31,scaled=1000")
        t0 = time.perf_counter()
        sig_gz, = sketch_file(gz, "k=31,scaled=1000")
        dt = time.perf_counter() - t0
        from sourmash_amd.sketch import gunzip_files, gunzip_counters, sketch_files
        c0 = gunzip_counters()
        _, stages = gunzip_files([gz])
        extra["ingest_gz"] = {
            "gz_bytes": os.path.getsize(gz), "inflated_bytes": part_bytes, "bases": part_bases, "seconds": round(dt, 3),
            "Gbase_per_s": round(part_bases / dt / 1e9, 2), "same_sketch_as_the_plain_file": bool(sig_gz.md5sum() == sig_part.md5sum()),
            "inflater_stages_ms": {k: round(v, 2) for k, v in stages.items()},
            "bound": "the device inflater (csrc/gunzip.hip): pass 1 -- one wavefront per deflate block, ~40 instructions a symbol, every "
                     "CU's issue slots full -- then the expansion of its records; PCIe carries the compressed third of the bytes",
            "what": "one gzip member (zlib level 1) of 400 MB of the same FASTA -> sketch, end to end; the member is inflated in HBM "
                    "(every block at once, CRC-32 checked), the host inflater is the fallback"}
        # the same member at zlib's default level (what `gzip` writes: longer matches, fewer symbols per byte)
        gz6 = os.path.join(tmp, "synth_part6.fa.gz")
        co = zlib.compressobj(6, zlib.DEFLATED, 31)
        with open(part, "rb") as fi, open(gz6, "wb") as fo:
            while True:
                block = fi.read(16 << 20)
                if not block:
                    break
                fo.write(co.compress(block))
            fo.write(co.flush())
        sketch_file(gz6, "k=31,scaled=1000")
        t0 = time.perf_counter()
        sig_gz6, = sketch_file(gz6, "k=31,scaled=1000")
        dt6 = time.perf_counter() - t0
        extra["ingest_gz_level6"] = {"gz_bytes": os.path.getsize(gz6), "bases": part_bases, "seconds": round(dt6, 3), "Gbase_per_s": round(part_bases / dt6 / 1e9, 2),
                                     "same_sketch_as_the_plain_file": bool(sig_gz6.md5sum() == sig_part.md5sum())}
        # 256 genomes: copies of the reference's E. coli K-12 fixture (1.3 MB .fna.gz each), `sketch` over all of them
        src = os.path.join(ROOT, "tests", "golden", "ecoli", "GCF_000005845.2_ASM584v2_genomic.fna.gz")
        many = []
        for i in range(256):
            many.append(os.path.join(tmp, f"g{i}.fna.gz"))
            shutil.copyfile(src, many[-1])
        sketch_files(many, "k=21,k=31,k=51,scaled=1000", threads=16)
        t0 = time.perf_counter()
        sigs = sketch_files(many, "k=21,k=31,k=51,scaled=1000", threads=16)
        dtm = time.perf_counter() - t0
        c1 = gunzip_counters()
        extra["ingest_gz_256_files"] = {
            "files": 256, "bases": 4_641_652 * 256, "seconds": round(dtm, 3), "Gbase_per_s": round(4_641_652 * 256 / dtm / 1e9, 2),
            "files_per_s": round(256 / dtm, 1), "golden_md5_k31": all(s.minhashes()[1].md5sum() == "0a8632c67e6d88f737ddb510bef90337" for s in sigs),
            "what": "smgpu_sketch_files: the members of a batch of files inflated in one pass and sketched together (k = 21, 31, 51)"}
        extra["gunzip_counters"] = {"inflated_on_the_device": c1[0] - c0[0], "handed_to_the_host_inflater": c1[1] - c0[1]}
        # ---- 10,000 signatures of ~5,000 hashes as a sourmash-style zip -> CSR in HBM ----
        zpath = os.path.join(tmp, "coll.zip")
        nsig = 10_000
        with zipfile.ZipFile(zpath, "w", zipfile.ZIP_STORED) as zf:
            man = io.StringIO()
            man.write("# SOURMASH-MANIFEST-VERSION: 1.0\n")
            man.write("internal_location,md5,md5short,ksize,moltype,num,scaled,n_hashes,with_abundance,name,filename\r\n")
            for i in range(nsig):
                mins = np.unique(splitmix64((np.uint64(i) << np.uint64(32)) + np.arange(5000, dtype=np.uint64)) % np.uint64(MAX_HASH_1000))
                text_mins = ",".join(map(str, mins.tolist()))
                md5 = hashlib.md5(("31" + text_mins.replace(",", "")).encode()).hexdigest()
                doc = ('[{"class":"sourmash_signature","email":"","hash_function":"0.murmur64","filename":"g%d.fa","name":"genome %d",'
                       '"license":"CC0","signatures":[{"num":0,"ksize":31,"seed":42,"max_hash":%d,"mins":[%s],"md5sum":"%s",'
                       '"molecule":"dna"}],"version":0.4}]' % (i, i, MAX_HASH_1000, text_mins, md5))
                loc = f"signatures/{md5}.sig.gz"
                zf.writestr(loc, gzip.compress(doc.encode(), compresslevel=1))
                man.write(f"{loc},{md5},{md5[:8]},31,DNA,0,1000,{len(mins)},0,genome {i},g{i}.fa\r\n")
            zf.writestr("SOURMASH-MANIFEST.csv", man.getvalue(), compress_type=zipfile.ZIP_DEFLATED)
        zbytes = os.path.getsize(zpath)
        index.SketchSet.load(zpath, ksize=31, moltype="DNA")   # warm
        dts = []
        for _ in range(3):
            t0 = time.perf_counter()
            db = index.SketchSet.load(zpath, ksize=31, moltype="DNA")
            dts.append(time.perf_counter() - t0)
        dt = min(dts)
        extra["sigload_10k"] = {
            "seconds_each_of_3_loads": [round(x, 3) for x in dts],
            "signatures": len(db), "zip_bytes": zbytes, "seconds": round(dt, 3), "signatures_per_s": round(len(db) / dt, 1),
            "zip_MB_per_s": round(zbytes / dt / 1e6, 1),
            "bound": "the device inflater's pass 1 over 427 MB of gzip members (digits: literals and short matches, many symbols a byte), "
                     "then the number parser (csrc/sigjson.hip); the host reads only what lies outside the hash arrays",
            "what": "SketchSet.load of a 10,000-member zip (stored .sig.gz members + manifest) -> one CSR in HBM, no per-sketch object; "
                    "100,000 members: profiles/r06_sigload.json (tools/bench_sigload.py)"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


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
    try:
        main()
    except BaseException:
        if GIVING_UP.is_set():       # a peer left through the watchdog while this rank was inside a collective: the line is out
            os._exit(0)
        raise
