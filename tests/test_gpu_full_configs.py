"""BASELINE.json configurations C2 (sketch 10 GB), C4 (compare, N = 10,000) and C5 (gather, 10^6-hash query vs
100,000 sketches) at FULL size on the GPU, each compared EXACTLY with the CPU oracle run on the host cores this
container may use (oracle.usable_cpus(): affinity capped by the cgroup quota):

  C2  identical sorted u64 hash vector for the whole 9.99e9-byte batch (+ lower-case / N-every-89th variants on 1e8)
  C4  identical u32 common matrix and bit-identical f64 Jaccard matrix, for the merge kernel and for the indexed
      (auto) path; reference semantics: src/sourmash/compare.py:14-64, src/core/src/sketch/minhash.rs:624-631,915-953
  C5  identical ordered list of (dataset index, |intersect|), for both device loops and for the sharded exchange
      protocol on one rank; reference semantics: src/sourmash/index/__init__.py:856-909, src/sourmash/search.py:15-37,
      915-919

The oracle walks what the reference walks (every pair, every dataset every round), so these tests take minutes of
host time (about 10^12 merge steps for C5); SMG_SKIP_FULL=1 skips them for quick local iterations on a GPU box."""
import hashlib
import os
import time

import numpy as np
import pytest

import oracle

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("SMG_SKIP_FULL") == "1", reason="SMG_SKIP_FULL=1")]


def test_sketch_c2_full_size_vs_oracle():
    import torch
    from sourmash_amd import device as smd
    n, record_len = 9_990_000_999, 10_000_000                    # bench.py's C2 batch: 999 records, seed 42, start 0
    seq = smd.synth_dna(n, seed=42, record_len=record_len)
    sk = smd.DeviceSketcher(31, 1000)
    got = sk.sketch(seq).cpu().numpy().view(np.uint64)
    host = seq.cpu().numpy()
    threads = oracle.usable_cpus()
    want = oracle.sketch_dna_bulk(host, 31, scaled=1000, nthreads=threads)
    assert got.size == want.size and np.array_equal(got, want)
    assert np.all(got[1:] > got[:-1])                            # sorted, unique
    # variants on a 1e8 prefix: lower case; N at every position i % 89 == 1 (benches/compute.rs:22-26)
    pre = host[:100_000_000].copy()
    low = pre.copy()
    low[(low >= 65) & (low <= 90)] += 32
    withn = pre.copy()
    withn[1::89] = ord("N")
    upper = sk.sketch(torch.from_numpy(pre).cuda()).cpu().numpy()
    for buf in (low, withn):
        g = sk.sketch(torch.from_numpy(buf).cuda()).cpu().numpy().view(np.uint64)
        assert np.array_equal(g, oracle.sketch_dna_bulk(buf, 31, scaled=1000, nthreads=threads))
    assert np.array_equal(sk.sketch(torch.from_numpy(low).cuda()).cpu().numpy(), upper)


def test_compare_c4_full_size_vs_oracle():
    import torch
    from sourmash_amd import device as smd
    from sourmash_amd.synth import synth_sketches
    n = 10_000
    sk = synth_sketches(n, seed=1234, pool_size=50_000, keep_one_in=10, planted=True)
    hh, ho = oracle.make_csr(sk)
    t0 = time.perf_counter()
    want_c, want_j = oracle.compare_all_pairs(hh, ho, nthreads=oracle.usable_cpus())
    print(f"oracle C4: {time.perf_counter() - t0:.1f} s on {oracle.usable_cpus()} threads")
    h, off = smd.pack_csr(sk)
    for method in ("merge", "auto"):
        common, jac = smd.compare_rows(h, off, method=method)
        torch.cuda.synchronize()
        got_c = common.cpu().numpy().view(np.uint32)
        assert got_c.shape == (n, n)
        assert np.array_equal(got_c, want_c), method
        assert np.array_equal(jac.cpu().numpy().view(np.uint64), want_j.view(np.uint64)), method   # f64 bit patterns
        del common, jac
    assert hashlib.sha256(want_c.tobytes()).hexdigest() == hashlib.sha256(got_c.tobytes()).hexdigest()


def test_gather_c5_full_size_vs_oracle():
    import torch
    from sourmash_amd import parallel
    from sourmash_amd.synth import synth_gather_device
    dev = torch.device("cuda", 0)
    nq, ndb, thr_bp = 1_000_000, 100_000, 50_000
    q, hashes, offsets = synth_gather_device(nq, ndb, 5000, dev)
    t0 = time.perf_counter()
    want = oracle.gather(q.cpu().numpy().view(np.uint64), hashes.cpu().numpy().view(np.uint64),
                         offsets.cpu().numpy().view(np.uint64), threshold_bp=thr_bp, scaled=1000,
                         nthreads=oracle.usable_cpus())
    print(f"oracle C5: {len(want)} rounds, {time.perf_counter() - t0:.1f} s on {oracle.usable_cpus()} threads")
    assert len(want) > 1000
    be = parallel.DeviceBackend(dev)
    got = parallel.gather_distributed(q, len(q), hashes, offsets, ndb, 0, thr_bp, 1000, be)
    assert got == want                                           # the native single-GPU loop
    stats = {}
    got = parallel.gather_distributed(q, len(q), hashes, offsets, ndb, 0, thr_bp, 1000, be, stepwise=True, stats=stats)
    assert got == want                                           # export -> load -> replay, as N ranks run it
    assert stats["exchanges"] < len(want) / 2                    # fewer collectives than rounds (two per round before)
