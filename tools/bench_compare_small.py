#!/usr/bin/env python3
"""Where `auto` (the cost model of csrc/capi.cpp: bitindex_build) stops choosing the general all-pairs kernel
(compare_hash_kernel): N = 16 ... 2,000 sketches of ~5,000 hashes, a collection with heavy sharing (pool 50,000) and one whose
hashes are mostly private (pool 2e6: a hash sits in n / 400 sketches).  Per row: ms of the general kernel, of a forced index (build + matrices), of auto, and
what auto took.  python tools/bench_compare_small.py -> JSON lines (GPU box)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sourmash_amd import device as smd
from sourmash_amd.synth import synth_sketches_device


def wall(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    rows = []
    dev = torch.device("cuda:0")
    for label, pool, keep in (("shared", 50_000, 10), ("private", 2_000_000, 400)):
        for n in (16, 32, 64, 128, 256, 512, 1000, 2000):
            h, off = synth_sketches_device(n, dev, seed=77, pool_size=pool, keep_one_in=keep, chunk=64)
            pairs = n * (n - 1) // 2
            c0, j0 = smd.compare_rows(h, off)
            ms_general = wall(lambda: smd.compare_rows(h, off, common=c0, jaccard=j0))

            def forced():
                idx = smd.BitIndex.build(h, off, threshold=max(1, int(n * 0.00256)), one_shot=False)   # the model's own threshold, forced
                return None if idx is None else smd.compare_rows(h, off, index=idx)
            f = forced()
            ms_index = wall(forced) if f is not None else None
            took = smd.BitIndex.build(h, off, one_shot=True)
            ms_auto = wall(lambda: smd.compare_rows(h, off, method="auto"))
            ca, ja = smd.compare_rows(h, off, method="auto")
            same = bool((ca[:n] == c0).all().item()) and bool((ja.view(torch.int64) == j0.view(torch.int64)).all().item())
            rows.append({"collection": label, "n": n, "mean_hashes": round(float(off[-1].item()) / n, 1), "pairs": pairs,
                         "general_ms": round(ms_general, 3), "index_ms_incl_build": None if ms_index is None else round(ms_index, 3),
                         "auto_ms": round(ms_auto, 3), "auto_took": "general kernel" if took is None else "index (builder %s)" % took.builder,
                         "auto_equals_general_bitwise": same,
                         "general_pairs_per_s": round(pairs / ms_general * 1e3, 1)})
            print(json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
