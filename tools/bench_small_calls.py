#!/usr/bin/env python3
"""Latency of the per-call object API (what a Python loop over records pays): MinHash.add_sequence for short and
medium sequences, count_common / jaccard of two sketches, seq_to_hashes.   python tools/bench_small_calls.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import sourmash_amd as sm
    rng = np.random.default_rng(5)
    out = {}
    for length, calls in ((150, 2000), (10_000, 1000), (1_000_000, 50)):
        seqs = [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), length)).decode() for _ in range(min(calls, 200))]
        mh = sm.MinHash(0, 31, scaled=1000)
        mh.add_sequence(seqs[0])
        t0 = time.perf_counter()
        for i in range(calls):
            mh.add_sequence(seqs[i % len(seqs)], True)
        dt = time.perf_counter() - t0
        out[f"add_sequence_{length}bp"] = {"us_per_call": round(dt / calls * 1e6, 1), "Mbase_per_s": round(length * calls / dt / 1e6, 1)}
    a, b = sm.MinHash(0, 31, scaled=1000), sm.MinHash(0, 31, scaled=1000)
    a.add_many(range(1, 10_001, 2))
    b.add_many(range(1, 10_001, 3))
    for name, fn in (("count_common", lambda: a.count_common(b)), ("jaccard", lambda: a.jaccard(b)),
                     ("contained_by", lambda: a.contained_by(b)), ("len", lambda: len(a)), ("copy", lambda: a.copy())):
        fn()
        t0 = time.perf_counter()
        for _ in range(500):
            fn()
        out[name + "_5000_hashes"] = {"us_per_call": round((time.perf_counter() - t0) / 500 * 1e6, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
