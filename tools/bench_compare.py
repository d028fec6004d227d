#!/usr/bin/env python3
"""Timing probe for the compare kernel on synthetic sketch collections (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sourmash_amd import device as smd
from sourmash_amd.synth import synth_sketches


def run(n, planted, pool=50_000, keep=10, reps=5):
    sk = synth_sketches(n, seed=1234, pool_size=pool, keep_one_in=keep, planted=planted)
    h, off = smd.pack_csr(sk)
    common, jac = smd.compare_rows(h, off)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        smd.compare_rows(h, off, common=common, jaccard=jac)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    # dense path: index build (sort + unique + bit rows) and the popcount matrix, timed separately
    t0 = time.perf_counter()
    idx = smd.BitIndex.build(h, off)
    torch.cuda.synchronize()
    t_build = (time.perf_counter() - t0) * 1e3
    if idx is not None:
        c2, j2 = smd.compare_rows(h, off, index=idx)
        torch.cuda.synchronize()
        same = bool((c2 == common).all().item()) and bool((j2 == jac).all().item())
        e0.record()
        for _ in range(reps):
            smd.compare_rows(h, off, common=c2, jaccard=j2, index=idx)
        e1.record()
        torch.cuda.synchronize()
        ms_b = e0.elapsed_time(e1) / reps
        print(f"    bits: U={idx.universe} build {t_build:.2f} ms + matrix {ms_b:.3f} ms  identical={same}  "
              f"{n * (n - 1) // 2 / (ms_b + t_build) / 1e3:.1f} Mpairs/s incl. build")
    else:
        print(f"    bits: too sparse (build probe {t_build:.2f} ms)")
    pairs = n * (n - 1) // 2
    sizes = np.array([len(s) for s in sk])
    steps = float(sizes.sum()) * (n - 1)        # sum over unordered pairs of (n_i + n_j)
    print(f"n={n} planted={planted} pool={pool} keep=1/{keep} mean={sizes.mean():.0f} max={sizes.max()} : {ms:.3f} ms  "
          f"{pairs / ms / 1e3:.2f} Mpairs/s  {steps / ms / 1e6:.1f} G merge-steps/s")


if __name__ == "__main__":
    run(1000, True)
    run(1000, False)
    run(2000, False)
    run(1000, False, pool=5000, keep=10)       # 500-hash sketches
    run(256, False)
    run(4000, False)
    run(1000, False, pool=5_000_000, keep=1000)   # sparse universe: U ~ 5e6
