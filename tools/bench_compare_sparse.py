#!/usr/bin/env python3
"""The indexed compare path on a collection of unrelated sketches (every hash held by one or two sketches), where
the merge kernel's N * sum(n) steps are almost all wasted: inputs generated in HBM, index build and matrix timed,
result tied to the merge kernel's.   python tools/bench_compare_sparse.py [n] [hashes per sketch]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sourmash_amd import device as smd  # noqa: E402
from sourmash_amd.synth import splitmix63, MAX_HASH_1000  # noqa: E402


def unrelated(n, size, dev):
    "90 % private hashes, 10 % drawn from a pool so large that a pool hash is held by ~2 sketches"
    pool = n * size // 20
    d = torch.arange(n, device=dev, dtype=torch.int64)[:, None]
    col = torch.arange(size, device=dev, dtype=torch.int64)[None, :]
    priv = splitmix63((d << 33) + col + (1 << 62)) % MAX_HASH_1000 + 1
    shared = splitmix63(splitmix63((d << 32) ^ col) % pool + 12345) % MAX_HASH_1000 + 1
    return torch.where(col < size // 10, shared, priv)


def clustered(n, size, dev, cluster=10):
    """genomes in clusters of ~10 (strains of a species): 90 % of a sketch's hashes are private to the genome, 10 % come from
    its cluster's core -- every member takes the same `size // 10` core hashes, so a core hash is held by ~10 sketches and
    pairs inside a cluster share ~10 % of their hashes, pairs across clusters nothing (what `sourmash compare` sees on a
    collection of unrelated species with a few strains each, compare.py:14-64)"""
    d = torch.arange(n, device=dev, dtype=torch.int64)[:, None]
    col = torch.arange(size, device=dev, dtype=torch.int64)[None, :]
    priv = splitmix63((d << 33) + col + (1 << 62)) % MAX_HASH_1000 + 1
    core = splitmix63(((d // cluster) << 34) + col + (1 << 61)) % MAX_HASH_1000 + 1
    return torch.where(col < size // 10, core, priv)


def to_csr(x, dev):
    n = x.shape[0]
    x = torch.sort(x, dim=1).values
    keep = torch.ones_like(x, dtype=torch.bool)
    keep[:, 1:] = x[:, 1:] != x[:, :-1]
    hashes = x[keep].contiguous()
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(keep.sum(dim=1), 0)
    return hashes, offsets


def run(name, hashes, offsets, n):
    out = {"collection": name, "n": n, "pairs": n * (n - 1) // 2, "total_hashes": int(hashes.numel())}
    idx = None
    for rep in range(2):
        idx = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx = smd.BitIndex.build(hashes, offsets)
        torch.cuda.synchronize()
        t_build = time.perf_counter() - t0
    out["index"] = None if idx is None else dict(zip(("bit_columns", "matrix_increments", "threshold"), idx.stats), builder=idx.builder)
    if idx is not None:
        c, j = smd.compare_rows(hashes, offsets, index=idx)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        smd.compare_rows(hashes, offsets, common=c, jaccard=j, index=idx)
        torch.cuda.synchronize()
        t_idx = time.perf_counter() - t0
        out["indexed"] = {"build_ms": round(t_build * 1e3, 2), "matrix_ms": round(t_idx * 1e3, 2),
                          "pairs_per_s_incl_build": round(out["pairs"] / (t_build + t_idx), 1)}
    # what the one-shot entry point does: cost model -> index or merge kernel, build included
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        one = smd.BitIndex.build(hashes, offsets, one_shot=True)
        ca, ja = smd.compare_rows(hashes, offsets, index=one) if one is not None else smd.compare_rows(hashes, offsets)
        torch.cuda.synchronize()
        t_auto = time.perf_counter() - t0
    out["auto"] = {"path": "merge kernel" if one is None else "index (%s builder): %d bit columns + inverted lists, %d matrix increments"
                   % (one.builder, one.stats[0], one.stats[1]), "ms": round(t_auto * 1e3, 2),
                   "pairs_per_s": round(out["pairs"] / t_auto, 1)}
    cm, jm = smd.compare_rows(hashes, offsets)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    smd.compare_rows(hashes, offsets, common=cm, jaccard=jm)
    torch.cuda.synchronize()
    t_m = time.perf_counter() - t0
    out["merge"] = {"ms": round(t_m * 1e3, 2), "pairs_per_s": round(out["pairs"] / t_m, 1)}
    out["auto_identical_to_merge"] = bool((ca == cm).all().item() and (ja.view(torch.int64) == jm.view(torch.int64)).all().item())
    if idx is not None:
        out["identical_counts"] = bool((c == cm).all().item())
        out["identical_jaccard_bits"] = bool((j.view(torch.int64) == jm.view(torch.int64)).all().item())
    out["nonzero_offdiagonal_pairs"] = int(((cm > 0).sum().item() - n) // 2)
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    dev = torch.device("cuda", 0)
    res = []
    for name, gen in (("unrelated", unrelated), ("clustered", clustered)):
        hashes, offsets = to_csr(gen(n, size, dev), dev)
        torch.cuda.synchronize()
        res.append(run(name, hashes, offsets, n))
        del hashes, offsets
        torch.cuda.empty_cache()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
