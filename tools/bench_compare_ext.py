"""Kernel rates of the batched bottom-k / abundance all-pairs launches (csrc/compare_ext.hip) on resident collections:
python tools/bench_compare_ext.py [n]   -> one JSON line"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from sourmash_amd import device as smd, parallel
    from sourmash_amd.synth import synth_sketches
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    be = parallel.DeviceBackend()
    dev = be.device
    lib, p, s = be.lib, be._p, be._s
    out = {}

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

    pairs = n * (n - 1) // 2
    # bottom-k: n sketches of num = 500 drawn from a common pool
    rng = np.random.default_rng(3)
    pool = np.unique(rng.integers(1, 2**62, 20_000, dtype=np.int64).astype(np.uint64))
    rows = [np.sort(rng.choice(pool, size=2000, replace=False))[:500] for _ in range(n)]
    h, off = smd.pack_csr(rows, device=dev)
    nums = torch.full((n,), 500, dtype=torch.int32, device=dev)
    common = torch.zeros((n, n), dtype=torch.int32, device=dev)
    union = torch.zeros((n, n), dtype=torch.int32, device=dev)
    jac = torch.zeros((n, n), dtype=torch.float64, device=dev)
    ms = timed(lambda: be.rustcall(lib.smgpu_compare_num_raw, p(h), p(off), p(nums), n, p(common), p(union), p(jac), s()))
    out["num_500"] = {"sketches": n, "hashes_per_sketch": 500, "pairs": pairs, "ms": round(ms, 3), "pairs_per_s": round(pairs / (ms * 1e-3), 1),
                      "mean_jaccard": float(jac.sum().item() - n) / (n * (n - 1))}
    # abundance: config-C3-shaped (n x ~5,000 hashes from a pool of 50,000)
    sk = synth_sketches(n, seed=1234)
    h, off = smd.pack_csr(sk, device=dev)
    ab = (h % 7 + 1) * ((h >> 3) % 11 + 1)
    tot = int(off[-1].item())               # the caller of a raw device entry point knows its array sizes
    prod = torch.zeros((n, n), dtype=torch.int64, device=dev)
    sq = torch.zeros((n,), dtype=torch.int64, device=dev)
    for narrow in (True, False):
        ms = timed(lambda: be.rustcall(lib.smgpu_compare_abund_raw_n, p(h), p(ab), p(off), n, tot, narrow, p(common), p(prod), p(sq), s()))
        out["abund_c3_%s" % ("u32" if narrow else "u64")] = {"sketches": n, "hashes": int(off[-1].item()), "pairs": pairs, "ms": round(ms, 3),
                                                            "pairs_per_s": round(pairs / (ms * 1e-3), 1),
                                                            "prod_checksum": int(prod.sum().item())}
    # abundance, related genomes: 2,000 hashes held by every sketch + 3,000 of its own (runs of 64 in every block list)
    rng = np.random.default_rng(11)
    core = np.unique(rng.integers(1, 2**54, 2000, dtype=np.int64).astype(np.uint64))
    sk = [np.unique(np.concatenate([core, rng.integers(1, 2**54, 3000, dtype=np.int64).astype(np.uint64)])) for _ in range(n)]
    h, off = smd.pack_csr(sk, device=dev)
    ab = (h % 7 + 1) * ((h >> 3) % 11 + 1)
    tot = int(off[-1].item())               # the caller of a raw device entry point knows its array sizes
    ms = timed(lambda: be.rustcall(lib.smgpu_compare_abund_raw_n, p(h), p(ab), p(off), n, tot, True, p(common), p(prod), p(sq), s()))
    out["abund_core_u32"] = {"sketches": n, "hashes": int(off[-1].item()), "pairs": pairs, "ms": round(ms, 3), "pairs_per_s": round(pairs / (ms * 1e-3), 1),
                             "prod_checksum": int(prod.sum().item()), "min_common": int((common + torch.eye(n, dtype=torch.int32, device=dev) * 10**6).min().item())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
