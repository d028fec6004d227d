#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X FracMinHash engine.

Metric (BASELINE.json): Gbase/s sketched, k=31, scaled=1000, DNA, seed 42.
Workload (BASELINE.json configs[1], "C2"): 10 GB of synthetic random DNA per GPU,
1,000 records of 10^7 bases, generated directly in HBM; one
"step" = one full pass of the hot path over that resident batch: the k-mer kernel
(canonicalise + MurmurHash3 + keep h <= max_hash), the device radix sort and the
unique pass, leaving the sorted unique hash vector (the sketch) in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--bases B]

N > 1 is launched by the driver with torch.distributed.run; every rank sketches
its own 10 GB slice of the stream (weak scaling), ranks exchange their hash
vectors with one RCCL all-gather after the timed region's last step so that every
rank holds the sketch of the whole input (set union is associative), and
value = (bases sketched by all ranks) / (max over ranks of the elapsed time).

Prints ONE JSON line on rank 0 with the contract fields plus
  roofline:     HBM roofline of the dominant kernel (algorithmic bytes / measured kernel time)
  cpu_baseline: the oracle (CPU restatement of the reference algorithm) on a bounded sample
  extra:        secondary metric: sketch-pairs/s on the 1,000 x 1,000 compare (config C3)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
# HBM traffic of sketch_dna_kernel<31,16,false> on the default C2 batch from the PMC passes committed in
# profiles/r01_end_pmc.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs):
# FETCH_SIZE 5,030,617 KiB, doubled as MI355X_MICROARCH.md prescribes for 16 B/lane streaming reads on
# gfx950, + WRITE_SIZE 78,545 KiB.  Only quoted when the live run uses that exact batch.
PMC_C2_BYTES = 2 * 5_030_617 * 1024 + 78_545 * 1024
PMC_C2_INPUT_BYTES = 9_990_000_999
# SQ_INSTS_VALU of the same kernel on the same batch (same file): wave-instructions per launch.  The issue model prices
# them by the kernel's static mix: 71 % of the per-k-mer instructions (multiplies, v_add3, v_alignbit, v_lshl_add_u64,
# permutes, compares) cost 4.3 cycles per wave-instruction per SIMD, 29 % (xor / and / add / lshr / bitop3 / mov) 2.45
# (profiles/r01_ubench_valu.txt) = 3.76 on average; 1,024 SIMDs at the 2.36 GHz the kernel runs at
# (GRBM_GUI_ACTIVE / 8 XCDs / kernel time).
PMC_C2_VALU_INSTS = 18_194_765_969
VALU_CYCLES_PER_INST, N_SIMDS, SHADER_HZ = 3.76, 1024, 2.36e9


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
    ap.add_argument("--no-compare", action="store_true")
    ap.add_argument("--cpu-sample", type=float, default=0.0, help="bases for the CPU baseline (0 = auto)")
    return ap.parse_args()


def main():
    args = parse()
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

    import sourmash_amd as sm
    from sourmash_amd import device as smd, parallel

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
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    total_bases = bases_per_step * world * args.steps
    value = total_bases / elapsed / 1e9

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel (sketch_dna_kernel): HIP events on the launch stream ----
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
                    "traffic": PMC_C2_BYTES if (n_bytes == PMC_C2_INPUT_BYTES and args.ksize == 31) else None,
                    "traffic_source": "profiles/r01_end_pmc.txt (PMC passes of this same command)",
                    "kernel_ms": round(kern_ms, 3), "algorithmic_bytes": alg_bytes,
                    "valu": ({"insts_per_kmer": round(PMC_C2_VALU_INSTS * 64 / bases_per_step, 1),
                              "issue_busy_frac_model": round(PMC_C2_VALU_INSTS * VALU_CYCLES_PER_INST / N_SIMDS / SHADER_HZ
                                                             / (kern_ms * 1e-3), 3),
                              "source": "profiles/r01_end_pmc.txt SQ_INSTS_VALU x 3.76 cycles (static mix x r01_ubench_valu.txt) / 1024 SIMDs / 2.36 GHz"}
                             if (n_bytes == PMC_C2_INPUT_BYTES and args.ksize == 31) else None),
                    "note": "VALU-integer bound (12 x 64-bit multiplies per k-mer), see DESIGN.md; "
                            "kernel-only Gbase/s = %.1f" % (bases_per_step / (kern_ms * 1e-3) / 1e9)}

        # ---- CPU baseline: the oracle on a bounded sample of the same stream ----
        cpu = None
        if not args.no_cpu_baseline and world == 1:         # the contract: rank 0 at N = 1 only
            import oracle
            cores = os.cpu_count() or 1
            sample = int(args.cpu_sample) if args.cpu_sample else int(min(n_bytes, 25e6 * cores, 1e9))
            host = seq[:sample].cpu().numpy()
            tc = time.perf_counter()
            ref = oracle.sketch_dna_bulk(host, args.ksize, scaled=args.scaled, nthreads=cores)
            tcpu = time.perf_counter() - tc
            sample_bases = int((host != 10).sum())
            # parity spot check of the GPU path on the very same sample
            got = sk.sketch(seq[:sample]).cpu().numpy().view(np.uint64)
            cpu = {"value": round(sample_bases / tcpu / 1e9, 4), "unit": "Gbase/s", "cores": cores, "kind": "port",
                   "sample": f"first {sample} bytes of the same synthetic stream ({sample_bases} bases), "
                             f"oracle.sketch_dna_bulk with {cores} OpenMP threads, {tcpu:.1f} s",
                   "gpu_matches_oracle_on_sample": bool(np.array_equal(got, ref))}

        # ---- secondary metrics: 1,000 x 1,000 compare (config C3) and a gather run (scaled-down C5) ----
        extra = {}
        if not args.no_compare and world == 1:              # single-GPU secondary metrics; N > 1 runs time the sketch only
            try:
                from sourmash_amd.synth import synth_sketches, synth_gather
                from sourmash_amd import parallel
                sketches = synth_sketches(1000, seed=1234)
                h, off = smd.pack_csr(sketches, device=dev)
                n = len(sketches)
                pairs = n * (n - 1) // 2
                sizes = (off[1:] - off[:-1]).cpu().numpy().astype(np.int64)
                alg = 8 * int((sizes.sum() * (n - 1)))             # sum over pairs of 8*(n_i+n_j)

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

                common, jac = smd.compare_rows(h, off)
                ms_merge = timed(lambda: smd.compare_rows(h, off, common=common, jaccard=jac))
                extra["compare_1000x1000_merge"] = {
                    "pairs_per_s": round(pairs / (ms_merge * 1e-3), 1), "ms": round(ms_merge, 3), "pairs": pairs,
                    "algorithmic_GBps": round(alg / (ms_merge * 1e-3) / 1e9, 1),
                    "kernel": "compare_tile_kernel (LDS-tiled merge walk; the general path)"}
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
                        "kernel": "bitmatrix_kernel (hashes held by many sketches as bit columns + popcount; auto-selected)"}
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
                # gather: 2e5-hash query vs 5,000 x ~1,000-hash database, threshold_bp = 50 kbp
                qh, dbh = synth_gather(n_query=200_000, n_db=5000, db_size=1000)
                gh, goff = smd.pack_csr(dbh, device=dev)
                gq = torch.from_numpy(qh.view(np.int64).copy()).to(dev)
                be = parallel.DeviceBackend(dev)
                thr_hashes = 50
                for _ in range(2):                                  # second pass: allocator / code objects warm
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    state = be.gather_state(gq, len(qh), gh, goff, len(dbh), 0)      # invert the database against the query
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    state.begin(thr_hashes, len(dbh))
                    res = state.run()                                # every round on the device
                    torch.cuda.synchronize()
                    t2 = time.perf_counter()
                extra["gather_200k_vs_5000"] = {"rounds": len(res), "index_build_ms": round((t1 - t0) * 1e3, 2),
                                                "loop_ms": round((t2 - t1) * 1e3, 2),
                                                "us_per_round": round((t2 - t1) * 1e6 / max(len(res), 1), 1),
                                                "note": "C5 at full size: profiles/r01_gather_c5.json (tools/bench_gather.py)"}
            except Exception as e:   # the headline metric must still print
                extra["error"] = repr(e)
            # ---- BASELINE configs C4 and C5 at full size on this one GPU (a few seconds; tools/ hold the property checks) ----
            try:
                del seq                                              # 10 GB back before the big matrices
                torch.cuda.empty_cache()
                sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
                import bench_gather as bg
                big = synth_sketches(10_000, seed=1234)
                bh, boff = smd.pack_csr(big, device=dev)
                bn = len(big)
                bpairs = bn * (bn - 1) // 2
                bc, bj = smd.compare_rows(bh, boff)
                ms_big = timed(lambda: smd.compare_rows(bh, boff, common=bc, jaccard=bj), reps=1)
                t0 = time.perf_counter()
                ca, ja = smd.compare_rows(bh, boff, method="auto")
                torch.cuda.synchronize()
                auto_big = (time.perf_counter() - t0) * 1e3
                extra["compare_10000x10000"] = {
                    "pairs": bpairs, "merge_ms": round(ms_big, 2), "merge_pairs_per_s": round(bpairs / (ms_big * 1e-3), 1),
                    "auto_ms": round(auto_big, 2), "auto_pairs_per_s": round(bpairs / (auto_big * 1e-3), 1),
                    "identical": bool((ca == bc).all().item() and (ja == bj).all().item()),
                    "note": "config C4 (pool-drawn sketches: the cost model picks bit columns); auto includes the index build"}
                del bc, bj, ca, ja, bh, boff
                torch.cuda.empty_cache()
                gq5, gh5, goff5 = bg.make_inputs(1_000_000, 100_000, 5000, dev)
                torch.cuda.synchronize()
                for _ in range(2):
                    t0 = time.perf_counter()
                    st5 = be.gather_state(gq5, gq5.numel(), gh5, goff5, 100_000, 0)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    st5.begin(50, 100_000)
                    res5 = st5.run()
                    torch.cuda.synchronize()
                    t2 = time.perf_counter()
                iso = [r[1] for r in res5]
                extra["gather_1M_vs_100000"] = {
                    "db_bytes": int(gh5.numel() * 8), "rounds": len(res5), "index_build_ms": round((t1 - t0) * 1e3, 2),
                    "loop_ms": round((t2 - t1) * 1e3, 2), "total_ms": round((t2 - t0) * 1e3, 2),
                    "overlaps_non_increasing": bool(all(a >= b for a, b in zip(iso, iso[1:]))), "last_overlap": iso[-1] if iso else None,
                    "note": "config C5 on one GPU, threshold_bp 50,000; full property checks: tools/bench_gather.py"}
            except Exception as e:
                extra["error_full_size"] = repr(e)

        out = {
            "metric": "Gbase/s sketched (k=31, scaled=1000)", "value": round(value, 3), "unit": "Gbase/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "C2: sketch 10 GB synthetic random-DNA per GPU (1,000 records x 1e7 bases, "
                                   "ASCII resident in HBM), k=31 scaled=1000 seed=42; kernel + radix sort + unique",
                       "bases_per_gpu": bases_per_step, "bytes_per_gpu": n_bytes, "ksize": args.ksize,
                       "scaled": args.scaled, "unique_hashes_rank0": n_unique_local,
                       "unique_hashes_job": n_unique_total, "allgather_ms": gather_ms},
            "roofline": roofline, "cpu_baseline": cpu, "extra": extra,
        }
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        # RCCL prints a version banner through C stdio, which sits in libc's buffer until exit when stdout is a file
        # or a pipe: push it out first so that the JSON is the last line of rank 0's output
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
