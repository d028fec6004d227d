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


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    dev = torch.device("cuda", 0)
    # 90 % private hashes, 10 % drawn from a pool so large that a pool hash is held by ~2 sketches
    pool = n * size // 20
    d = torch.arange(n, device=dev, dtype=torch.int64)[:, None]
    col = torch.arange(size, device=dev, dtype=torch.int64)[None, :]
    priv = splitmix63((d << 33) + col + (1 << 62)) % MAX_HASH_1000 + 1
    shared = splitmix63(splitmix63((d << 32) ^ col) % pool + 12345) % MAX_HASH_1000 + 1
    x = torch.where(col < size // 10, shared, priv)
    x = torch.sort(x, dim=1).values
    keep = torch.ones_like(x, dtype=torch.bool)
    keep[:, 1:] = x[:, 1:] != x[:, :-1]
    hashes = x[keep].contiguous()
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(keep.sum(dim=1), 0)
    torch.cuda.synchronize()
    out = {"n": n, "pairs": n * (n - 1) // 2, "total_hashes": int(hashes.numel())}
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx = smd.BitIndex.build(hashes, offsets)
        torch.cuda.synchronize()
        t_build = time.perf_counter() - t0
    out["index"] = None if idx is None else dict(zip(("bit_columns", "matrix_increments", "threshold"), idx.stats))
    if idx is not None:
        c, j = smd.compare_rows(hashes, offsets, index=idx)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        smd.compare_rows(hashes, offsets, common=c, jaccard=j, index=idx)
        torch.cuda.synchronize()
        t_idx = time.perf_counter() - t0
        out["indexed"] = {"build_ms": round(t_build * 1e3, 2), "matrix_ms": round(t_idx * 1e3, 2),
                          "pairs_per_s_incl_build": round(out["pairs"] / (t_build + t_idx), 1)}
    cm, jm = smd.compare_rows(hashes, offsets)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    smd.compare_rows(hashes, offsets, common=cm, jaccard=jm)
    torch.cuda.synchronize()
    t_m = time.perf_counter() - t0
    out["merge"] = {"ms": round(t_m * 1e3, 2), "pairs_per_s": round(out["pairs"] / t_m, 1)}
    if idx is not None:
        out["identical_counts"] = bool((c == cm).all().item())
        out["identical_jaccard_bits"] = bool((j.view(torch.int64) == jm.view(torch.int64)).all().item())
        out["nonzero_offdiagonal_pairs"] = int(((cm > 0).sum().item() - n) // 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
