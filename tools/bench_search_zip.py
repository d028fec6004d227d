#!/usr/bin/env python3
"""`search` / `prefetch` of one query against a zip collection of N signatures (~5,000 hashes each), through the Index
API: ZipFileLinearIndex.find loads the archive natively into one CSR in HBM and scores it in one pass (first query:
load + score; later queries: score only); the base-class walk creates a signature object per member first (what the
reference's loop does), timed on a sample.

    python tools/bench_search_zip.py [--n 10000]
"""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_load import write_zip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000)
    ap.add_argument("--sample", type=int, default=500, help="members walked one object at a time for comparison")
    args = ap.parse_args()
    import sourmash_amd as sm
    from sourmash_amd.index import Index, LinearIndex, ZipFileLinearIndex
    from sourmash_amd.search import make_containment_query
    out = {"n": args.n}
    with tempfile.TemporaryDirectory() as tmp:
        zpath = os.path.join(tmp, "coll.zip")
        write_zip(zpath, args.n)
        zidx = ZipFileLinearIndex.load(zpath).select(ksize=31, moltype="DNA")
        # the query: member 17 plus half of member 4711 % n
        it = zidx.signatures()
        sigs = [next(it) for _ in range(min(args.sample, args.n))]
        q_mh = sigs[17 % len(sigs)].minhash.to_mutable()
        q_mh.add_many(list(sigs[-1].minhash.hashes)[::2])
        query = sm.SourmashSignature(q_mh, name="query")
        t0 = time.perf_counter()
        first = list(zidx.prefetch(query, threshold_bp=50000))
        out["first_query_s"] = round(time.perf_counter() - t0, 3)
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            again = list(zidx.prefetch(query, threshold_bp=50000))
        out["next_query_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 2)
        assert [(r.score, r.signature.md5sum()) for r in again] == [(r.score, r.signature.md5sum()) for r in first]
        out["matches"] = len(first)
        out["signatures_per_s_first"] = round(args.n / out["first_query_s"], 1)
        out["signatures_per_s_next"] = round(args.n / (out["next_query_ms"] * 1e-3), 1)
        # per-object walk on a sample: objects from the archive, one CSR from the objects, then the same scoring
        t0 = time.perf_counter()
        it = zidx.signatures()
        sample = LinearIndex([next(it) for _ in range(len(sigs))])
        slow = list(Index.find(sample, make_containment_query(query.minhash, 50000), query))
        dt = time.perf_counter() - t0
        out["object_walk"] = {"sample": len(sigs), "s": round(dt, 3), "signatures_per_s": round(len(sigs) / dt, 1),
                              "matches": len(slow)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
