#!/usr/bin/env python3
"""BASELINE config C2 at FULL size, bit for bit: the 10 GB synthetic batch of bench.py (1,000 records x 1e7 bases,
seed 42) sketched on the GPU (k=31, scaled=1000) and by the CPU oracle on every host core; the two sorted u64
hash vectors must be identical.  Also checks the lower-case and N-every-89th variants of SURVEY.md section 8d on a
1e8 prefix.   python tests/check_c2_full.py  (GPU box; ~1 min of host time for the oracle).
Lives under tests/ because it calls the oracle (test infrastructure); not collected by pytest -- its size-independent
counterparts in the suite are tests/test_gpu_sketch.py and bench.py's 1e9-byte parity sample."""
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from sourmash_amd import device as smd  # noqa: E402


def main():
    n_records, record_len = 1000, 10_000_000
    n = 9_990_000_999                       # bench.py's C2 batch: 999 records x (1e7 bases + 1 separator), seed 42, start 0
    seq = smd.synth_dna(n, seed=42, record_len=record_len)
    sk = smd.DeviceSketcher(31, 1000)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = sk.sketch(seq)
    torch.cuda.synchronize()
    gpu_s = time.perf_counter() - t0
    got = got.cpu().numpy().view(np.uint64)
    host = seq.cpu().numpy()
    threads = os.cpu_count()
    t0 = time.perf_counter()
    want = oracle.sketch_dna_bulk(host, 31, scaled=1000, nthreads=threads)
    cpu_s = time.perf_counter() - t0
    out = {"bytes": int(n), "gpu_hashes": int(got.size), "oracle_hashes": int(want.size),
           "identical": bool(np.array_equal(got, want)),
           "md5_of_hash_vector": hashlib.md5(got.tobytes()).hexdigest(),
           "gpu_s": round(gpu_s, 4), "oracle_s": round(cpu_s, 2), "oracle_threads": threads,
           "oracle_Gbase_per_s": round(n / cpu_s / 1e9, 3)}
    # variants on a 1e8 prefix: lower case; N at every position i % 89 == 1 (src/core/benches/compute.rs:22-26)
    pre = host[:100_000_000].copy()
    variants = {}
    low = pre.copy()
    acgt = (low >= 65) & (low <= 90)
    low[acgt] += 32
    withn = pre.copy()
    withn[1::89] = ord("N")
    for name, buf in (("lowercase", low), ("n_every_89", withn)):
        g = sk.sketch(torch.from_numpy(buf).cuda()).cpu().numpy().view(np.uint64)
        w = oracle.sketch_dna_bulk(buf, 31, scaled=1000, nthreads=threads)
        variants[name] = {"hashes": int(g.size), "identical": bool(np.array_equal(g, w))}
    variants["lowercase"]["same_as_uppercase"] = bool(np.array_equal(
        sk.sketch(torch.from_numpy(low).cuda()).cpu().numpy(), sk.sketch(torch.from_numpy(pre).cuda()).cpu().numpy()))
    out["variants_1e8"] = variants
    print(json.dumps(out))
    return 0 if out["identical"] and all(v["identical"] for v in variants.values()) else 1


if __name__ == "__main__":
    sys.exit(main())
