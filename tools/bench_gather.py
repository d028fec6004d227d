"""Gather (BASELINE config C5) at full size on one GPU: inputs generated in HBM, index build and the
min-set-cover loop timed separately, results checked through size-independent properties with an
independent kernel (the streaming overlap kernel of pair_ops.hip, which shares no code with the postings walk).

    python tools/bench_gather.py                      # C5: 1e6-hash query vs 100,000 x ~5,000
    python tools/bench_gather.py --ndb 12500          # one GPU's shard of the 8-GPU layout

Generator (device-side variant of sourmash_amd/synth.py: synth_gather; same construction, hashes drawn as
(splitmix64(x) >>> 1) mod m so that torch's signed int64 arithmetic can express it): query = nq distinct
hashes below max_hash(scaled=1000); every database sketch takes half of its hashes from the query and half
from a private stream."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sourmash_amd import parallel  # noqa: E402

from sourmash_amd.synth import synth_gather_device  # noqa: E402


def make_inputs(nq, ndb, dbsize, dev, seed=777, chunk=10_000):
    return synth_gather_device(nq, ndb, dbsize, dev, seed=seed, chunk=chunk)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nq", type=int, default=1_000_000)
    ap.add_argument("--ndb", type=int, default=100_000)
    ap.add_argument("--dbsize", type=int, default=5000)
    ap.add_argument("--threshold-bp", type=int, default=50_000)
    ap.add_argument("--stepwise", action="store_true", help="time the sharded step protocol (one rank) as well")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    be = parallel.DeviceBackend(dev)
    t0 = time.perf_counter()
    q, hashes, offsets = make_inputs(args.nq, args.ndb, args.dbsize, dev)
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t0
    nq, ndb, total = len(q), args.ndb, int(offsets[-1].item())

    def timed_build():
        torch.cuda.synchronize()
        t = time.perf_counter()
        st = be.gather_state(q, nq, hashes, offsets, ndb, 0)
        torch.cuda.synchronize()
        return st, time.perf_counter() - t

    import gc
    if os.environ.get("SMG_BENCH_GC") != "1":
        gc.collect()
        gc.freeze()                                        # the interpreter's cyclic GC stays out of the timed regions:
        gc.disable()                                       # a full pass over torch's object graph costs 40-70 ms
    st, _ = timed_build()                                  # warm (allocator, code objects)
    del st
    st, build_s = timed_build()
    thr = int(np.ceil(args.threshold_bp / 1000))
    st.begin(thr, ndb)
    torch.cuda.synchronize()
    t = time.perf_counter()
    res = st.run()
    t_back = time.perf_counter() - t
    torch.cuda.synchronize()
    run_s = time.perf_counter() - t
    idx = np.array([r[0] for r in res], dtype=np.int64)
    isect = np.array([r[1] for r in res], dtype=np.int64)

    # ---- properties that hold for the reference's greedy loop at any size --------------------------------
    checks = {}
    checks["winners_distinct"] = bool(len(set(idx.tolist())) == len(idx))
    checks["overlaps_non_increasing"] = bool((np.diff(isect) <= 0).all())
    checks["all_rounds_meet_threshold"] = bool((isect >= thr).all())
    # the union of the winners covers exactly sum(isect) query hashes
    off_h = offsets.cpu().numpy()
    covered = torch.zeros(nq, dtype=torch.bool, device=dev)
    for gi in idx.tolist():
        row = hashes[off_h[gi]:off_h[gi + 1]]
        pos = torch.searchsorted(q, row).clamp_(max=nq - 1)
        covered[pos[q[pos] == row]] = True
    checks["sum_isect_equals_covered"] = bool(int(covered.sum().item()) == int(isect.sum()))
    # remaining counters == |row ∩ uncovered query| recomputed by the streaming kernel; none reaches the threshold
    left = q[~covered].contiguous()
    recount = be.zeros((ndb,), torch.int64)
    be.overlaps(left, len(left), hashes, offsets, ndb, recount, 0)
    torch.cuda.synchronize()
    recount = recount.cpu().numpy().view(np.uint64)
    checks["final_counters_match_streaming_recount"] = bool(np.array_equal(st.counters(), recount))
    checks["stop_rule_holds"] = bool(recount.max() < max(thr, 1) or len(left) < thr or len(res) == ndb)
    # replay of round 0 and of the last round with the streaming kernel: the winner is the arg-max, ties lowest
    first = be.zeros((ndb,), torch.int64)
    be.overlaps(q, nq, hashes, offsets, ndb, first, 0)
    first = first.cpu().numpy().view(np.uint64)
    checks["round0_is_argmax_lowest_index"] = bool(len(res) == 0 or (int(np.argmax(first)) == idx[0] and int(first.max()) == isect[0]))

    # the overlap pass on its own (search / prefetch over the resident collection): |Q ∩ row| for every row
    scratch = be.zeros((ndb,), torch.int64)
    be.overlaps(q, nq, hashes, offsets, ndb, scratch, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        be.overlaps(q, nq, hashes, offsets, ndb, scratch, 0)
    e1.record()
    torch.cuda.synchronize()
    overlap_ms = e0.elapsed_time(e1) / 3
    checks["overlap_pass_equals_build_counters"] = bool(np.array_equal(scratch.cpu().numpy().view(np.uint64), first))

    out = {"config": {"query_hashes": nq, "datasets": ndb, "db_hashes": total, "db_bytes": total * 8,
                      "threshold_bp": args.threshold_bp, "scaled": 1000},
           "generate_s": round(gen_s, 3), "index_build_ms": round(build_s * 1e3, 2),
           "postings": int(be.lib.smgpu_gather_postings(st._ptr)),
           "rounds": len(res), "loop_ms": round(run_s * 1e3, 2), "loop_call_ms": round(t_back * 1e3, 2),
           "us_per_round": round(run_s * 1e6 / max(len(res), 1), 2),
           "total_ms": round((build_s + run_s) * 1e3, 2),
           "overlap_pass_ms": round(overlap_ms, 3), "overlap_pass_GBps": round(total * 8 / overlap_ms / 1e6, 1),
           "streaming_equivalent_bytes": int(8 * (nq + total) * max(len(res), 1)),
           "first": res[:3], "last": res[-3:], "checks": checks}
    if args.stepwise:
        torch.cuda.synchronize()
        t = time.perf_counter()
        res2 = parallel.gather_distributed(q, nq, hashes, offsets, ndb, 0, args.threshold_bp, 1000, be, stepwise=True)
        torch.cuda.synchronize()
        out["stepwise_total_ms"] = round((time.perf_counter() - t) * 1e3, 2)
        out["stepwise_identical"] = bool(res2 == res)
    print(json.dumps(out))
    return 0 if all(checks.values()) else 1


if __name__ == "__main__":
    sys.exit(main())
