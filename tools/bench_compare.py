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


def c4():
    """BASELINE config C4 at full size (10,000 sketches x ~5,000 hashes, 49,995,000 pairs, planted edge rows):
    both compare paths timed, results tied together by size-independent properties."""
    import json
    n = 10_000
    t0 = time.perf_counter()
    sk = synth_sketches(n, seed=1234, pool_size=50_000, keep_one_in=10, planted=True)
    gen_s = time.perf_counter() - t0
    h, off = smd.pack_csr(sk)
    sizes = torch.tensor([len(s) for s in sk], device=h.device, dtype=torch.int64)
    common, jac = smd.compare_rows(h, off)                       # merge kernel
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    smd.compare_rows(h, off, common=common, jaccard=jac)
    e1.record()
    torch.cuda.synchronize()
    ms_merge = e0.elapsed_time(e1)
    t0 = time.perf_counter()
    idx = smd.BitIndex.build(h, off)
    torch.cuda.synchronize()
    build_ms = (time.perf_counter() - t0) * 1e3
    c2, j2 = smd.compare_rows(h, off, index=idx)
    torch.cuda.synchronize()
    e0.record()
    smd.compare_rows(h, off, common=c2, jaccard=j2, index=idx)
    e1.record()
    torch.cuda.synchronize()
    ms_bits = e0.elapsed_time(e1)
    cm = common.to(torch.int64)
    checks = {
        "merge_equals_bits_counts": bool((c2 == common).all().item()),
        "merge_equals_bits_jaccard_bitwise": bool((j2.view(torch.int64) == jac.view(torch.int64)).all().item()),
        "symmetric": bool((common == common.T).all().item()),
        "diagonal_is_size": bool((torch.diagonal(cm) == sizes).all().item()),
        "common_le_min_size": bool((cm <= torch.minimum(sizes[:, None], sizes[None, :])).all().item()),
        "planted_duplicate_jaccard_1": bool(jac[0, n - 4].item() == 1.0 and cm[0, n - 4].item() == len(sk[0])),
        "planted_disjoint_row_zero": bool(cm[n - 3].sum().item() == len(sk[n - 3])),
        "planted_superset_row_is_sizes": bool((cm[n - 1, : n - 4] == sizes[: n - 4]).all().item()),
        "jaccard_is_one_divide": bool((jac == cm.double() / torch.clamp(sizes[:, None] + sizes[None, :] - cm, min=1).double()).all().item()),
    }
    pairs = n * (n - 1) // 2
    alg = float(sizes.sum().item()) * (n - 1) * 8
    out = {"config": {"n": n, "pairs": pairs, "mean_hashes": float(sizes.double().mean().item()), "csr_bytes": int(h.numel() * 8)},
           "generate_s": round(gen_s, 2),
           "merge": {"ms": round(ms_merge, 2), "pairs_per_s": round(pairs / ms_merge * 1e3, 1), "algorithmic_GBps": round(alg / ms_merge / 1e6, 1)},
           "bits": {"index_build_ms": round(build_ms, 2), "matrix_ms": round(ms_bits, 2), "universe": idx.universe,
                    "pairs_per_s_incl_build": round(pairs / (ms_bits + build_ms) * 1e3, 1)},
           "counts_checksum": int(cm.sum().item()), "checks": checks}
    print(json.dumps(out))
    return 0 if all(checks.values()) else 1


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "c4":
        sys.exit(c4())
    run(1000, True)
    run(1000, False)
    run(2000, False)
    run(1000, False, pool=5000, keep=10)       # 500-hash sketches
    run(256, False)
    run(4000, False)
    run(1000, False, pool=5_000_000, keep=1000)   # sparse universe: U ~ 5e6
