"""GPU parity: merge-intersection kernels (pair ops + the LDS-tiled all-pairs
compare) through the C-ABI, against the oracle and the reference's golden
compare values.  Run with -m gpu."""
import glob

import numpy as np
import pytest

import oracle
from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sm():
    import torch  # noqa: F401
    import sourmash_amd
    assert sourmash_amd.gpu_available()
    return sourmash_amd


def _load(sm, path, ksize=None):
    return list(sm.load_signatures_from_json(path, ksize=ksize))


def _omh(d):
    scaled = oracle.scaled_for_max_hash(d["max_hash"]) if d["max_hash"] else 0
    num = 0 if d["max_hash"] else d["num"]
    ab = "abundances" in d
    mh = oracle.OracleMinHash(num, d["ksize"], scaled=scaled, seed=d["seed"], track_abundance=ab)
    if ab:
        for h, a in sorted(zip(d["mins"].tolist(), d["abundances"].tolist())):
            mh.add_hash_with_abundance(h, a)
    else:
        mh.add_many(np.sort(d["mins"]))
    return mh


def test_demo_matrix_num_sketches(sm):
    # tests/test_compare.py:48-63 -- num=500 sketches: bottom-k rule on the GPU pair path
    from sourmash_amd.compare import compare_all_pairs
    sigs = [s for f in sorted(glob.glob(golden("demo", "*.sig"))) for s in _load(sm, f)]
    got = compare_all_pairs(sigs, ignore_abundance=True)
    want = np.array([
        [1.0, 0.356, 0.078, 0.086, 0.0, 0.0, 0.0],
        [0.356, 1.0, 0.072, 0.078, 0.0, 0.0, 0.0],
        [0.078, 0.072, 1.0, 0.074, 0.0, 0.0, 0.0],
        [0.086, 0.078, 0.074, 1.0, 0.0, 0.0, 0.0],
        [0.0, 0.0, 0.0, 0.0, 1.0, 0.382, 0.364],
        [0.0, 0.0, 0.0, 0.0, 0.382, 1.0, 0.386],
        [0.0, 0.0, 0.0, 0.0, 0.364, 0.386, 1.0]])
    np.testing.assert_array_equal(got, want)


def test_scaled_on_real_data(sm):
    # tests/test_jaccard.py:207-232
    a = _load(sm, golden("scaled100", "GCF_000005845.2_ASM584v2_genomic.fna.gz.sig.gz"))[0].minhash
    b = _load(sm, golden("scaled100", "GCF_000006945.1_ASM694v1_genomic.fna.gz.sig.gz"))[0].minhash
    assert round(a.similarity(b), 5) == 0.01644 == round(b.similarity(a), 5)
    a2, b2 = a.downsample(scaled=1000), b.downsample(scaled=1000)
    assert round(a2.similarity(b2), 5) == 0.01874
    a3, b3 = a2.downsample(scaled=10000), b2.downsample(scaled=10000)
    assert a3.similarity(b3) == 0.01
    assert round(a.similarity(b2, downsample=True), 5) == 0.01874   # minhash.rs:688-696
    with pytest.raises(ValueError) as e:                              # MismatchScaled -> ValueError
        a.similarity(b2)
    assert "mismatch in scaled" in str(e.value)
    assert a.count_common(b2, downsample=True) == a2.count_common(b2)
    with pytest.raises(ValueError):
        a.count_common(b2)


def test_pair_ops_vs_oracle(sm):
    da = oracle.read_sig_json(golden("pairs", "47.fa.sig"))[0]
    db = oracle.read_sig_json(golden("pairs", "63.fa.sig"))[0]
    a, b = _load(sm, golden("pairs", "47.fa.sig"))[0].minhash, _load(sm, golden("pairs", "63.fa.sig"))[0].minhash
    oa, ob = _omh(da), _omh(db)
    assert a.count_common(b) == oa.count_common(ob) == b.count_common(a)
    assert a.intersection_and_union_size(b) == oa.intersection_and_union_size(ob)
    assert a.jaccard(b) == oa.jaccard(ob) and a.similarity(b) == oa.similarity(ob)
    inter = a & b
    want = np.intersect1d(oa.mins, ob.mins)
    assert np.array_equal(inter._mins_array(), want) and inter.scaled == a.scaled
    # the float layer against its independent restatement (oracle.contained_by follows minhash.py:819-841): same bits
    assert a.contained_by(b) == oracle.contained_by(len(want), len(a), a.scaled)
    assert b.contained_by(a) == oracle.contained_by(len(want), len(b), a.scaled)
    assert a.max_containment(b) == oracle.max_containment(len(want), len(a), len(b), a.scaled)
    assert a.avg_containment(b) == oracle.avg_containment(len(want), len(a), len(b), a.scaled)
    assert a.containment_ani(b).dist == oracle.containment_to_distance_point(a.contained_by(b), a.ksize)
    e = sm.MinHash(0, 31, scaled=1000)
    assert a.count_common(e) == 0 and a.jaccard(e) == 0.0 and e.jaccard(e) == 0.0   # tests/test_minhash.py:115-132
    assert e.contained_by(a) == 0.0
    # incompatible sketches: check order ksize -> moltype -> scaled -> seed (minhash.rs:886-912)
    for other, msg in ((sm.MinHash(0, 21, scaled=1000), "different ksizes"),
                       (sm.MinHash(0, 31, scaled=1000, seed=43), "mismatch in seed")):
        with pytest.raises(ValueError) as err:
            a.count_common(other)
        assert msg in str(err.value)
    with pytest.raises(TypeError):
        a.intersection_and_union_size(sm.MinHash(0, 21, scaled=1000))


def test_angular_similarity(sm):
    da = oracle.read_sig_json(golden("pairs", "track_abund_47.fa.sig"))[0]
    db = oracle.read_sig_json(golden("pairs", "track_abund_63.fa.sig"))[0]
    a = _load(sm, golden("pairs", "track_abund_47.fa.sig"))[0].minhash
    b = _load(sm, golden("pairs", "track_abund_63.fa.sig"))[0].minhash
    oa, ob = _omh(da), _omh(db)
    assert a.track_abundance and b.track_abundance
    # integer sums on the GPU, sqrt/acos on the host with the same libm as the oracle: bit-identical
    assert a.similarity(b) == oa.similarity(ob) == a.angular_similarity(b)
    assert a.similarity(a) == 1.0
    assert a.similarity(b, ignore_abundance=True) == oa.jaccard(ob)
    with pytest.raises(TypeError):
        a.angular_similarity(b.flatten())


def _check_csr(sm, sketches):
    mhs = []
    for s in sketches:
        mh = sm.MinHash(0, 31, scaled=1000)
        mh.add_many(s)
        mhs.append(mh)
    from sourmash_amd.compare import common_matrix
    common, jac = common_matrix(mhs)
    hashes, offsets = oracle.make_csr(sketches)
    wc, wj = oracle.compare_all_pairs(hashes, offsets, nthreads=8)
    assert np.array_equal(common, wc)
    assert np.array_equal(jac.view(np.uint64), wj.view(np.uint64))       # bit-identical f64
    return common, jac


def test_compare_small_and_ragged(sm):
    from sourmash_amd.synth import synth_sketches
    rng = np.random.default_rng(1)
    sk = synth_sketches(40, pool_size=3000)
    sk.append(np.zeros(0, dtype=np.uint64))                              # empty row
    sk.append(np.sort(rng.choice(sk[-2], size=1234, replace=False)))     # subset of the big row
    sk.append(np.arange(1, 700, dtype=np.uint64))                        # dense small values
    sk.append(np.array([2**64 - 1], dtype=np.uint64))                    # max u64 (only fits scaled=1 in practice)
    sk[-1] = np.array([18446744073709552], dtype=np.uint64)              # exactly max_hash
    common, jac = _check_csr(sm, sk)
    n = len(sk)
    assert np.array_equal(common, common.T) and np.all(np.diag(jac) == 1.0)
    assert common[0, 36] == len(sk[0]) and jac[0, 36] == 1.0            # planted duplicate of row 0
    _check_csr(sm, [sk[3]])                                              # 1 x 1
    _check_csr(sm, sk[:17])                                              # one tile + 1


def test_compare_c3_full(sm):
    """BASELINE config C3: 1,000 sketches x ~5,000 hashes (+ planted rows), full matrix vs oracle."""
    from sourmash_amd.synth import synth_sketches
    common, jac = _check_csr(sm, synth_sketches(1000, seed=1234))
    assert common.shape == (1000, 1000)


def test_compare_device_rows(sm):
    import torch
    from sourmash_amd import device as smd
    from sourmash_amd.synth import synth_sketches
    sk = synth_sketches(150, pool_size=8000)
    h, off = smd.pack_csr(sk)
    full_c, full_j = smd.compare_rows(h, off)
    torch.cuda.synchronize()
    hashes, offsets = oracle.make_csr(sk)
    wc, wj = oracle.compare_all_pairs(hashes, offsets, nthreads=8)
    assert np.array_equal(full_c.cpu().numpy().view(np.uint32), wc)
    # row-block shards (the multi-GPU partition) reproduce the same rows
    for lo, hi in ((0, 37), (37, 101), (101, 150)):
        c, j = smd.compare_rows(h, off, lo, hi)
        torch.cuda.synchronize()
        assert np.array_equal(c.cpu().numpy().view(np.uint32), wc[lo:hi])
        assert np.array_equal(j.cpu().numpy().view(np.uint64), wj[lo:hi].view(np.uint64))


def test_containment_matrices(sm):
    from sourmash_amd.compare import (compare_serial_containment, compare_serial_max_containment,
                                      compare_serial_avg_containment)
    sigs = [_load(sm, golden("pairs", f))[0] for f in ("47.fa.sig", "63.fa.sig")]
    c = compare_serial_containment(sigs)
    assert c[0, 1] == sigs[1].contained_by(sigs[0]) and c[1, 0] == sigs[0].contained_by(sigs[1])
    m = compare_serial_max_containment(sigs)
    assert m[0, 1] == m[1, 0] == sigs[0].max_containment(sigs[1])
    a = compare_serial_avg_containment(sigs)
    assert a[0, 1] == sigs[0].avg_containment(sigs[1])


def _mixed_collection(n, seed=3):
    """Sketches with three kinds of hashes: a shared pool (held by ~1/4 of the sketches each: bit columns), hashes
    shared by exactly two or three sketches (inverted lists) and private ones (runs of length one)."""
    from sourmash_amd.synth import splitmix64, MAX_HASH_1000
    rng = np.random.default_rng(seed)
    pool = np.unique(splitmix64(np.arange(600, dtype=np.uint64) + np.uint64(17)) % np.uint64(MAX_HASH_1000))
    few = np.unique(splitmix64(np.arange(4000, dtype=np.uint64) + np.uint64(1 << 40)) % np.uint64(MAX_HASH_1000))
    rows = [set(pool[rng.random(len(pool)) < 0.25].tolist()) for _ in range(n)]
    for h in few.tolist():
        for r in rng.choice(n, size=int(rng.integers(2, 4)), replace=False):
            rows[int(r)].add(h)
    for i in range(n):
        priv = splitmix64(np.arange(300, dtype=np.uint64) + np.uint64((i + 1) << 44)) % np.uint64(MAX_HASH_1000)
        rows[i].update(priv.tolist())
    rows[5] = set()                                                       # an empty sketch in the middle
    return [np.array(sorted(r), dtype=np.uint64) for r in rows]


def test_compare_index_splits_frequent_and_rare_hashes(sm):
    "the indexed path (bit columns + inverted lists) against the oracle and against the merge kernel, whole and sharded"
    import torch
    from sourmash_amd import device as smd
    sk = _mixed_collection(333)
    n = len(sk)
    wc, wj = oracle.compare_all_pairs(*oracle.make_csr(sk), nthreads=8)
    h, off = smd.pack_csr(sk)
    idx = smd.BitIndex.build(h, off, threshold=8)                        # pool hashes -> bit columns, the rest -> lists
    frequent, rare_pairs, threshold = idx.stats
    assert 0 < frequent <= 600 and rare_pairs > 4000 and threshold == 8   # both parts are in play
    auto = smd.BitIndex.build(h, off)                                    # the cost model's own split gives the same matrix
    assert auto is not None
    c_auto, _ = smd.compare_rows(h, off, index=auto)
    assert np.array_equal(c_auto.cpu().numpy().view(np.uint32), wc)
    c_idx, j_idx = smd.compare_rows(h, off, index=idx)
    c_mrg, j_mrg = smd.compare_rows(h, off)
    torch.cuda.synchronize()
    assert np.array_equal(c_idx.cpu().numpy().view(np.uint32), wc)
    assert np.array_equal(c_mrg.cpu().numpy().view(np.uint32), wc)
    assert np.array_equal(j_idx.cpu().numpy().view(np.uint64), wj.view(np.uint64))
    for first, stride, count in ((0, 2, 11), (1, 2, 10), (3, 5, 4)):     # owned 16-row tiles of a sharded launch
        out = idx.compare_tiles(first, stride, count).cpu().numpy().view(np.uint32)
        for t in range(count):
            lo = (first + t * stride) * 16
            hi = min(lo + 16, n)
            assert np.array_equal(out[t * 16:t * 16 + hi - lo], wc[lo:hi]), (first, stride, t)
        # the triangle form (callers mirror afterwards): everything on or above the diagonal, whatever lies below
        up = idx.compare_tiles(first, stride, count, upper=True).cpu().numpy().view(np.uint32)
        for t in range(count):
            lo = (first + t * stride) * 16
            for r in range(lo, min(lo + 16, n)):
                assert np.array_equal(up[t * 16 + r - lo, r:], wc[r, r:]), (first, stride, r)
    # the host convenience entry point picks the same path and the same numbers
    mhs = []
    for hs in sk[:60]:
        mh = sm.MinHash(0, 31, scaled=1000)
        mh.add_many(hs)
        mhs.append(mh)
    from sourmash_amd.compare import compare_all_pairs
    got = compare_all_pairs([sm.SourmashSignature(m, name=str(i)) for i, m in enumerate(mhs)], ignore_abundance=True)
    assert np.array_equal(got.view(np.uint64), wj[:60, :60].view(np.uint64))


def test_compare_index_all_rare(sm):
    "unrelated sketches: no bit columns at all, and a collection with nothing in common"
    import torch
    from sourmash_amd import device as smd
    from sourmash_amd.synth import synth_sketches
    sk = synth_sketches(200, pool_size=400_000, keep_one_in=400, planted=False)    # every pool hash in ~0.5 sketches
    h, off = smd.pack_csr(sk)
    idx = smd.BitIndex.build(h, off, threshold=200)                      # no hash can be held by more than all sketches
    assert idx is not None and idx.stats[0] == 0 and idx.stats[1] > 0
    c, j = smd.compare_rows(h, off, index=idx)
    torch.cuda.synchronize()
    wc, wj = oracle.compare_all_pairs(*oracle.make_csr(sk), nthreads=8)
    assert np.array_equal(c.cpu().numpy().view(np.uint32), wc) and np.array_equal(j.cpu().numpy().view(np.uint64), wj.view(np.uint64))


def test_ani_matrices_golden(sm):
    "tests/test_compare.py:94-196 of the reference: Jaccard / containment / max / avg containment ANI of four genomes"
    from sourmash_amd.compare import (compare_all_pairs, compare_parallel, compare_serial, compare_serial_avg_containment,
                                      compare_serial_containment, compare_serial_max_containment)
    sigs = []
    for f in ("2.fa.sig", "2+63.fa.sig", "47.fa.sig", "63.fa.sig"):
        sigs.extend(s for s in sm.load_signatures_from_json(golden("pairs", f), ksize=31) if s.minhash.scaled)
    assert len(sigs) == 4
    for ignore_abundance in (True, False):
        j_ani = compare_serial(sigs, ignore_abundance, downsample=False, return_ani=True)
        np.testing.assert_array_almost_equal(j_ani, np.array([[1.0, 0.978, 0.0, 0.0],
                                                              [0.978, 1.0, 0.96973012, 0.99262776],
                                                              [0.0, 0.96973012, 1.0, 0.97697011],
                                                              [0.0, 0.99262776, 0.97697011, 1.0]]), decimal=3)
        assert np.array_equal(compare_parallel(sigs, ignore_abundance, downsample=False, n_jobs=2, return_ani=True), j_ani)
        assert np.array_equal(compare_all_pairs(sigs, ignore_abundance, downsample=False, n_jobs=2, return_ani=True), j_ani)
    np.testing.assert_array_almost_equal(compare_serial_containment(sigs, return_ani=True),
                                         np.array([[1, 0.966, 0.0, 0.0],
                                                   [1, 1.0, 0.97715525, 1.0],
                                                   [0.0, 0.96377054, 1.0, 0.97678608],
                                                   [0.0, 0.98667513, 0.97715525, 1.0]]), decimal=3)
    np.testing.assert_array_almost_equal(compare_serial_max_containment(sigs, return_ani=True),
                                         np.array([[1.0, 1.0, 0.0, 0.0],
                                                   [1.0, 1.0, 0.97715525, 1.0],
                                                   [0.0, 0.97715525, 1.0, 0.97715525],
                                                   [0.0, 1.0, 0.97715525, 1.0]]), decimal=3)
    np.testing.assert_array_almost_equal(compare_serial_avg_containment(sigs, return_ani=True),
                                         np.array([[1.0, 0.983, 0.0, 0.0],
                                                   [0.983, 1.0, 0.97046289, 0.99333757],
                                                   [0.0, 0.97046289, 1.0, 0.97697067],
                                                   [0.0, 0.99333757, 0.97697067, 1.0]]), decimal=3)
    # the plain (non-ANI) containment matrices against the per-pair API
    cont = compare_serial_containment(sigs)
    for i in range(4):
        for j in range(4):
            assert cont[i][j] == (1.0 if i == j else sigs[j].contained_by(sigs[i]))


def _sigs_from_arrays(sm, arrays, scaled=1000, ksize=31, abund=()):
    sigs = []
    for i, arr in enumerate(arrays):
        mh = sm.MinHash(0, ksize, scaled=scaled, track_abundance=i in abund)
        if i in abund:
            mh.set_abundances({int(h): 1 + (int(h) % 7) for h in arr})
        else:
            mh.add_many(arr)
        sigs.append(sm.SourmashSignature(mh, name=f"s{i}"))
    return sigs


def test_float_layer_matrices_bit_exact_on_c3(sm):
    """compare.py:67-187 on config C3 (1,000 sketches): the whole-array float layer == the oracle's scalar restatement
    of minhash.py:819-841,881-905,946-959 / distance_utils.py:276-283 for EVERY entry, and == the per-pair object API
    on a sample of entries.  `==` on f64, not approx."""
    import time
    from sourmash_amd.compare import (compare_serial_containment, compare_serial_max_containment,
                                      compare_serial_avg_containment)
    from sourmash_amd.synth import synth_sketches
    sk = synth_sketches(1000, seed=1234)
    sigs = _sigs_from_arrays(sm, sk)
    wc, _ = oracle.compare_all_pairs(*oracle.make_csr(sk), nthreads=oracle.usable_cpus())
    sizes = [len(s) for s in sk]
    n = len(sk)
    t0 = time.perf_counter()
    cont = compare_serial_containment(sigs)
    mx = compare_serial_max_containment(sigs)
    avg = compare_serial_avg_containment(sigs)
    cont_ani = compare_serial_containment(sigs, return_ani=True)
    mx_ani = compare_serial_max_containment(sigs, return_ani=True)
    avg_ani = compare_serial_avg_containment(sigs, return_ani=True)
    assert time.perf_counter() - t0 < 30                          # six 1000 x 1000 matrices (the per-pair loops: ~10 min)
    trusted = [s.minhash.size_is_accurate() for s in sigs]
    rng = np.random.default_rng(5)
    rows = sorted(set(rng.integers(0, n, 40).tolist()) | {0, n - 4, n - 3, n - 2, n - 1})    # incl. the planted rows
    for i in rows:
        for j in range(n):
            if i == j:
                assert cont[i, j] == mx[i, j] == avg[i, j] == cont_ani[i, j] == 1.0
                continue
            c = int(wc[i, j])
            w = oracle.contained_by(c, sizes[j], 1000)
            assert cont[i, j] == w, (i, j)
            wm = oracle.max_containment(c, sizes[j], sizes[i], 1000)
            assert mx[i, j] == wm, (i, j)
            assert avg[i, j] == oracle.avg_containment(c, sizes[j], sizes[i], 1000), (i, j)
            ok = trusted[i] and trusted[j]
            a1 = 1 - oracle.containment_to_distance_point(w, 31)
            a2 = 1 - oracle.containment_to_distance_point(oracle.contained_by(c, sizes[i], 1000), 31)
            assert cont_ani[i, j] == (a1 if ok else 0.0), (i, j)
            assert mx_ani[i, j] == ((1 - oracle.containment_to_distance_point(wm, 31)) if ok else 0.0), (i, j)
            assert avg_ani[i, j] == ((a1 + a2) / 2 if ok else 0.0), (i, j)
    # the per-pair object API (the reference's loop bodies) on a sample
    for i, j in zip(rng.integers(0, n, 60).tolist(), rng.integers(0, n, 60).tolist()):
        if i == j:
            continue
        assert cont[i, j] == sigs[j].contained_by(sigs[i])
        assert mx[i, j] == sigs[j].max_containment(sigs[i])
        assert avg[i, j] == sigs[j].avg_containment(sigs[i])
        ani = sigs[j].containment_ani(sigs[i]).ani
        assert cont_ani[i, j] == (0.0 if ani is None else ani)
        ani = sigs[j].max_containment_ani(sigs[i]).ani
        assert mx_ani[i, j] == (0.0 if ani is None else ani)


def test_mixed_scaled_and_mixed_abundance_follow_the_per_pair_rule(sm):
    """compare.py:14-187 call similarity / contained_by(downsample=True) PER PAIR: a pair is compared at the coarser scaled
    of THAT pair (not of the whole list), and angular similarity applies to exactly the pairs whose two sketches track
    abundance (minhash.rs:682-702)."""
    from sourmash_amd.compare import compare_serial, compare_serial_containment, compare_serial_max_containment
    from sourmash_amd.synth import synth_sketches
    sk = synth_sketches(6, seed=7, pool_size=30_000, keep_one_in=3, planted=False)
    sigs = _sigs_from_arrays(sm, sk[:4], scaled=1000)
    coarse = [s.minhash.downsample(scaled=2000) for s in _sigs_from_arrays(sm, sk[4:], scaled=1000)]
    sigs += [sm.SourmashSignature(mh, name="coarse") for mh in coarse]
    with pytest.raises(ValueError):
        compare_serial(sigs, True, downsample=False)                 # MismatchScaled, like the reference's first mixed pair
    from sourmash_amd.compare import compare_serial_avg_containment
    j = compare_serial(sigs, True, downsample=True)
    c = compare_serial_containment(sigs, downsample=True)
    m = compare_serial_max_containment(sigs, downsample=True)
    av = compare_serial_avg_containment(sigs, downsample=True)
    c_ani = compare_serial_containment(sigs, downsample=True, return_ani=True)
    m_ani = compare_serial_max_containment(sigs, downsample=True, return_ani=True)
    for a in range(len(sigs)):
        for b in range(len(sigs)):
            if a == b:
                continue
            assert j[a, b] == sigs[a].similarity(sigs[b], ignore_abundance=True, downsample=True)
            assert c[a, b] == sigs[b].contained_by(sigs[a], downsample=True)
            hi, lo = (a, b) if a > b else (b, a)                     # the reference's loops call the method on the higher index
            assert m[a, b] == sigs[hi].max_containment(sigs[lo], downsample=True)
            assert av[a, b] == sigs[hi].avg_containment(sigs[lo], downsample=True)
            r = sigs[b].containment_ani(sigs[a], downsample=True).ani
            assert c_ani[a, b] == (0.0 if r is None else r)
            r = sigs[hi].max_containment_ani(sigs[lo], downsample=True).ani
            assert m_ani[a, b] == (0.0 if r is None else r)
    # the (0, 1) pair is compared at scaled 1000 although the list holds scaled-2000 sketches
    assert j[0, 1] == sigs[0].jaccard(sigs[1]) != sigs[0].minhash.downsample(scaled=2000).jaccard(sigs[1].minhash.downsample(scaled=2000))
    # abundance: sketches 1 and 3 weighted, the others flat
    mixed = _sigs_from_arrays(sm, sk[:5], abund=(1, 3))
    sims = compare_serial(mixed, False)
    for a in range(5):
        for b in range(5):
            if a != b:
                assert sims[a, b] == mixed[a].similarity(mixed[b], ignore_abundance=False)
    assert sims[1, 3] == mixed[1].minhash.angular_similarity(mixed[3].minhash) != mixed[1].jaccard(mixed[3])
    assert np.array_equal(compare_serial(mixed, True), compare_serial(_sigs_from_arrays(sm, sk[:5]), True))


def test_compare_extreme_hash_values():
    """The hash-table tile kernel keeps 2^64 - 1 (possible with scaled = 1) out of its table and counts it out of band;
    0 is an ordinary key.  Ragged collection with those values planted, more rows than one tile, vs the oracle (the general
    all-pairs kernel: whole matrix and row blocks)."""
    import torch
    from sourmash_amd import device as smd
    rng = np.random.default_rng(11)
    top = np.uint64(2**64 - 1)
    pool = np.unique(rng.integers(0, 2**63, 4000, dtype=np.int64).astype(np.uint64) * np.uint64(2) + np.uint64(1))
    sk = []
    for i in range(70):
        n = int(rng.integers(0, 900)) if i % 7 else int(rng.integers(0, 3))
        row = set(rng.choice(pool, size=min(n, len(pool)), replace=False).tolist())
        if i % 3 == 0:
            row.add(int(top))
        if i % 4 == 0:
            row.add(0)
        sk.append(np.array(sorted(row), dtype=np.uint64))
    sk[5] = np.array([int(top)], dtype=np.uint64)
    sk[6] = np.array([0], dtype=np.uint64)
    sk[8] = np.unique(np.concatenate([pool, np.array([0, int(top)], dtype=np.uint64)]))      # 4,000+ hashes: many rounds
    wc, wj = oracle.compare_all_pairs(*oracle.make_csr(sk), nthreads=4)
    h, off = smd.pack_csr(sk)
    c, j = smd.compare_rows(h, off)
    torch.cuda.synchronize()
    assert np.array_equal(c.cpu().numpy().view(np.uint32), wc)
    assert np.array_equal(j.cpu().numpy().view(np.uint64), wj.view(np.uint64))
    for lo, hi in ((0, 16), (16, 70), (3, 41)):                    # row blocks (the unsymmetric launch)
        cb, _ = smd.compare_rows(h, off, lo, hi)
        torch.cuda.synchronize()
        assert np.array_equal(cb.cpu().numpy().view(np.uint32), wc[lo:hi])


def _slot_of_table_kernel(v, log_slots=12):
    "the even slot compare.hip's pair_slot starts a hash's probe sequence on (white box: the collision cases below aim at it)"
    x = (v & np.uint64(0xffffffff)) ^ (v >> np.uint64(32))
    x = x ^ (x >> np.uint64(15))
    return (x & np.uint64((1 << (log_slots - 1)) - 1)) << np.uint64(1)


def test_compare_table_kernel_counters_and_probe_chains():
    """The corners of the hash-table tile kernel's round (csrc/compare.hip), against the oracle's per-pair walk
    (minhash.rs:915-953):
    * 48 identical sketches -- every lookup finds all 16 rows, so every lane's 4-bit counters go up every round and are
      flushed at exactly 15, many times over (2,000 hashes = 32 rounds; and 120 hashes: no flush before the last round);
    * rows that differ in one hash each at the front / the back / the 64th and 65th place (round boundaries);
    * hashes that all start their probe sequence on the SAME slot (chains hundreds of slots long, wrapping the table),
      alone and mixed with ordinary ones, shared by some rows and not by others;
    * a full tile next to a ragged one (50 rows, 37 columns' worth) so row and column padding is exercised."""
    import torch
    from sourmash_amd import device as smd
    rng = np.random.default_rng(77)

    def check(sk, what):
        sk = [np.unique(np.asarray(r, dtype=np.uint64)) for r in sk]
        wc, wj = oracle.compare_all_pairs(*oracle.make_csr(sk), nthreads=4)
        h, off = smd.pack_csr(sk)
        c, j = smd.compare_rows(h, off)                                    # no index: the tile kernel
        torch.cuda.synchronize()
        assert np.array_equal(c.cpu().numpy().view(np.uint32), wc), what
        assert np.array_equal(j.cpu().numpy().view(np.uint64), wj.view(np.uint64)), what

    base = np.unique(rng.integers(1, 2**62, 2000, dtype=np.int64).astype(np.uint64))
    check([base] * 48, "identical, 32 rounds")
    check([base[:120]] * 50, "identical, 2 rounds, ragged tile")
    check([base[:960]] * 16 + [base[:961]] * 16 + [base[1:960]] * 18, "identical up to the ends, 15 rounds + 1")
    variants = []
    for i in range(50):
        r = base.copy()
        r = np.delete(r, [i, 63 + (i % 3), 1999 - i])                    # holes at the front, around the first round's end, at the back
        if i % 5 == 0:
            r = np.append(r, np.uint64(2**64 - 1))                       # the empty-slot key as a hash
        if i % 7 == 0:
            r = np.append(r, np.uint64(0))
        variants.append(r)
    check(variants, "one-hash differences")
    # hashes whose probe sequences all start on one slot
    cand = rng.integers(1, 2**63, 6_000_000, dtype=np.int64).astype(np.uint64)
    same = np.unique(cand[_slot_of_table_kernel(cand) == np.uint64(2 * 1000)])
    assert len(same) >= 2000
    same = same[:2000]
    check([same[:700]] * 20 + [same[200:1500]] * 20 + [same[::2]] * 10, "one slot: chains of hundreds of slots")
    near_end = np.unique(cand[_slot_of_table_kernel(cand) >= np.uint64(4096 - 6)])[:1500]      # chains that wrap around the table
    mixed = []
    for i in range(48):
        part = rng.choice(near_end, size=int(rng.integers(0, 1200)), replace=False)
        rest = rng.choice(base, size=int(rng.integers(0, 1500)), replace=False)
        mixed.append(np.concatenate([part, rest, same[: (i * 37) % 900]]))
    check(mixed, "wrapping chains mixed with ordinary hashes")


def _index_builder_cases():
    """Both builders of the compare index must give the oracle's matrix: the sort-free one (csrc/dictindex.hip: buckets of
    the hash space, LDS tables) and the sort it falls back to (csrc/sparse_pairs.hip).  Run in a subprocess per builder."""
    import torch
    from sourmash_amd import device as smd
    from sourmash_amd.synth import synth_sketches
    rng = np.random.default_rng(23)
    top = np.uint64(2**64 - 1)
    pool = np.unique(rng.integers(0, 2**63, 3000, dtype=np.int64).astype(np.uint64) * np.uint64(2) + np.uint64(1))
    sk = []
    for i in range(150):                                          # more than one 128-row chunk of pass 2
        n = int(rng.integers(0, 700)) if i % 7 else int(rng.integers(0, 3))
        row = set(rng.choice(pool, size=n, replace=False).tolist())
        if i % 3 == 0:
            row.add(int(top))                                     # the tables' empty marker is a legal hash (scaled = 1)
        if i % 4 == 0:
            row.add(0)
        sk.append(np.array(sorted(row), dtype=np.uint64))
    sk[5] = np.array([int(top)], dtype=np.uint64)
    sk[6] = np.array([0], dtype=np.uint64)
    sk[9] = np.array([], dtype=np.uint64)
    wc, wj = oracle.compare_all_pairs(*oracle.make_csr(sk), nthreads=4)
    h, off = smd.pack_csr(sk)
    for threshold in (1, 8, 60, 10_000):                          # 2^64 - 1 (50 holders) frequent or rare; all rare
        idx = smd.BitIndex.build(h, off, threshold=threshold)
        assert idx is not None
        c, j = smd.compare_rows(h, off, index=idx)
        torch.cuda.synchronize()
        assert np.array_equal(c.cpu().numpy().view(np.uint32), wc), threshold
        assert np.array_equal(j.cpu().numpy().view(np.uint64), wj.view(np.uint64)), threshold
        cb, _ = smd.compare_rows(h, off, 16, 112, index=idx)      # a row block: every column, no mirror
        torch.cuda.synchronize()
        assert np.array_equal(cb.cpu().numpy().view(np.uint32), wc[16:112]), threshold
    # a pool collection (every hash frequent) and one whose distinct hashes overflow the buckets' tables (-> the sort)
    for sk2, thr in ((synth_sketches(300, pool_size=20_000, keep_one_in=8, planted=True), 0),
                     (synth_sketches(260, pool_size=3_000_000, keep_one_in=600, planted=False), 300)):
        wc2, _ = oracle.compare_all_pairs(*oracle.make_csr(sk2), nthreads=8)
        h2, off2 = smd.pack_csr(sk2)
        idx2 = smd.BitIndex.build(h2, off2, threshold=thr or None)
        assert idx2 is not None
        c2, _ = smd.compare_rows(h2, off2, index=idx2)
        torch.cuda.synchronize()
        assert np.array_equal(c2.cpu().numpy().view(np.uint32), wc2)
        if thr:
            assert idx2.universe > 512 * 1024                     # more distinct hashes than the tables hold: the fallback ran
    # corners of the bucketing: hashes below the number of buckets, one sketch, two identical sketches, only empty sketches,
    # one crowded bucket (2,000 hashes that differ in their low bits only) next to a spread-out rest
    crowded = np.uint64(1) << np.uint64(40)
    corner_sets = [
        [np.arange(0, 300, 3, dtype=np.uint64), np.arange(0, 300, 2, dtype=np.uint64), np.array([7], dtype=np.uint64)],
        [np.unique(rng.integers(0, 2**62, 500, dtype=np.int64).astype(np.uint64))],
        [np.arange(10, 4000, 7, dtype=np.uint64) * np.uint64(2**40)] * 2,
        [np.array([], dtype=np.uint64)] * 3,
        [np.concatenate([crowded + np.arange(2000, dtype=np.uint64), np.uint64(2**50) + np.arange(5, dtype=np.uint64) * np.uint64(2**45)]),
         np.concatenate([crowded + np.arange(0, 2000, 2, dtype=np.uint64), np.array([2**55], dtype=np.uint64)])] * 3,
    ]
    for sk3 in corner_sets:
        sk3 = [np.unique(r) for r in sk3]
        wc3, wj3 = oracle.compare_all_pairs(*oracle.make_csr(sk3), nthreads=2)
        h3, off3 = smd.pack_csr(sk3)
        for threshold in (1, 2, 1000):
            idx3 = smd.BitIndex.build(h3, off3, threshold=threshold)
            if idx3 is None:                                      # nothing to index (no hashes at all)
                assert sum(len(r) for r in sk3) == 0
                continue
            c3, j3 = smd.compare_rows(h3, off3, index=idx3)
            torch.cuda.synchronize()
            assert np.array_equal(c3.cpu().numpy().view(np.uint32), wc3), (len(sk3), threshold)
            assert np.array_equal(j3.cpu().numpy().view(np.uint64), wj3.view(np.uint64)), (len(sk3), threshold)


@pytest.mark.parametrize("builder", ["dict", "sort"])
def test_compare_index_builders_agree(builder):
    import os, subprocess, sys
    from conftest import ROOT
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_compare as t\nt._index_builder_cases()\nprint('ok')\n" % (ROOT, os.path.join(ROOT, "tests")))
    env = dict(os.environ)
    if builder == "sort":
        env["SMG_COMPARE_INDEX"] = "sort"
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), (p.stdout[-1500:], p.stderr[-3000:])


def test_compare_index_random_collections(sm):
    """Random shapes through every compare path -- general kernel, indexed path with the threshold the cost model picks and
    with forced ones (the sort-free builder's row chunks of 128, bucket ranges, carried words, rare lists), whole matrix
    (triangle + mirror) and a row block (all columns) -- against the oracle, bit for bit."""
    import torch
    from sourmash_amd import device as smd
    rng = np.random.default_rng(2024)
    for case in range(24):
        n = int(rng.choice([1, 2, 3, 15, 16, 17, 63, 127, 128, 129, 200, 300, 400]))
        top_bits = int(rng.choice([9, 20, 40, 54, 63, 64]))        # hash range: from fewer values than buckets to all 64 bits
        pool_size = int(rng.choice([50, 700, 5000, 40_000]))
        hi = (1 << top_bits) - 1
        pool = np.unique(rng.integers(0, hi, size=pool_size, dtype=np.uint64, endpoint=True))
        sk = []
        for _ in range(n):
            size = int(rng.integers(0, min(len(pool), 900) + 1))
            row = rng.choice(pool, size=size, replace=False)
            if rng.random() < 0.3:                                  # private hashes: rare ones next to the shared pool
                row = np.concatenate([row, rng.integers(0, hi, size=int(rng.integers(1, 50)), dtype=np.uint64, endpoint=True)])
            sk.append(np.unique(row.astype(np.uint64)))
        wc, wj = oracle.compare_all_pairs(*oracle.make_csr(sk), nthreads=4)
        h, off = smd.pack_csr(sk)
        c, j = smd.compare_rows(h, off)
        torch.cuda.synchronize()
        assert np.array_equal(c.cpu().numpy().view(np.uint32), wc), ("general", case, n)
        if sum(len(r) for r in sk) == 0:
            continue
        for threshold in (None, 1, int(rng.integers(2, 40))):
            idx = smd.BitIndex.build(h, off, threshold=threshold)
            if idx is None:                                         # the cost model preferred the general kernel
                assert threshold is None
                continue
            c2, j2 = smd.compare_rows(h, off, index=idx)
            torch.cuda.synchronize()
            assert np.array_equal(c2.cpu().numpy().view(np.uint32), wc), ("indexed", case, n, threshold)
            assert np.array_equal(j2.cpu().numpy().view(np.uint64), wj.view(np.uint64)), ("indexed", case, n, threshold)
            if n > 16:
                lo = 16 * int(rng.integers(0, n // 16))
                hi_row = int(rng.integers(lo + 1, n + 1))
                cb, _ = smd.compare_rows(h, off, lo, hi_row, index=idx)
                torch.cuda.synchronize()
                assert np.array_equal(cb.cpu().numpy().view(np.uint32), wc[lo:hi_row]), ("block", case, n, threshold, lo, hi_row)


# ---- batched launches for what used to be a per-pair loop: bottom-k, abundance, mixed scaled, Jaccard ANI ----------------
def _oracle_sketch(hashes, ksize=31, num=0, scaled=0, abunds=None):
    mh = oracle.OracleMinHash(num, ksize, scaled=scaled, track_abundance=abunds is not None)
    if abunds is not None:
        for h, a in zip(hashes, abunds):
            mh.add_hash_with_abundance(int(h), int(a))
    else:
        mh.add_many(np.asarray(hashes, dtype=np.uint64))
    return mh


def _bottom_k_collection(n, num, seed, pool_size=20_000, draw=2000):
    "n bottom-k sketches over a common pool: sketch i = the `num` smallest of its own random draw (fewer when the draw is small)"
    rng = np.random.default_rng(seed)
    pool = np.unique(rng.integers(1, 2**62, pool_size, dtype=np.int64).astype(np.uint64))
    out = []
    for i in range(n):
        k = draw if i % 17 else int(rng.integers(0, num))              # some sketches hold fewer than num hashes
        out.append(np.sort(rng.choice(pool, size=min(k, len(pool)), replace=False))[:num])
    out[3] = out[2].copy()                                             # identical sketches
    out[5] = np.zeros(0, dtype=np.uint64)                              # an empty one
    out[7] = out[6][: num // 3].copy()                                 # a prefix of another
    return out


def test_num_all_pairs_batched_vs_oracle(sm):
    """Bottom-k collections in ONE launch (csrc/compare_ext.hip): |A ∩ B ∩ merged|, |merged| and Jaccard of every pair
    == the oracle's literal restatement of minhash.rs:593-631 (merge both into a sketch truncated to num, intersect),
    f64 bit for bit; `num` is the lower-indexed sketch's where they differ.  The reference's golden 7 x 7 demo matrix
    (tests/test_compare.py:48-63) through the same path, counts included."""
    from sourmash_amd.compare import compare_all_pairs, num_matrix
    sigs = [s for f in sorted(glob.glob(golden("demo", "*.sig"))) for s in _load(sm, f)]
    jac, common, union = num_matrix([s.minhash for s in sigs], want_counts=True)
    omh = [_omh(oracle.read_sig_json(f)[0]) for f in sorted(glob.glob(golden("demo", "*.sig")))]
    for i in range(7):
        for j in range(i + 1, 7):
            assert (int(common[i, j]), int(union[i, j])) == omh[i].intersection_and_union_size(omh[j])
            assert common[j, i] == common[i, j] and union[j, i] == union[i, j]
    assert np.array_equal(jac, compare_all_pairs(sigs, ignore_abundance=True)) and jac[0, 1] == 0.356
    # config-C3-shaped: 1,000 sketches of num = 500 (ragged: short, empty, identical, prefix sketches planted)
    arrays = _bottom_k_collection(1000, 500, seed=3)
    mhs = []
    for i, a in enumerate(arrays):
        mh = sm.MinHash(500 if i % 5 else 300, 31)                     # two different num values in one list
        mh.add_many(a)
        mhs.append(mh)
    want = oracle.similarity_matrix([_oracle_sketch(np.asarray(mh._mins_array()), num=mh.num) for mh in mhs], ignore_abundance=True,
                                    nthreads=oracle.usable_cpus())
    got = compare_all_pairs([sm.SourmashSignature(mh, name=str(i)) for i, mh in enumerate(mhs)], ignore_abundance=False)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert got[2, 3] == 1.0 and (got[5] == np.eye(1000)[5]).all() and 0 < got[10, 11] < 1
    # per-pair API and batch agree (the pair kernel of pair_ops.hip is the round-1 form of the same rule)
    for i, j in ((0, 1), (6, 7), (2, 3), (17, 34), (4, 5), (10, 995)):
        assert got[i, j] == mhs[i].similarity(mhs[j])


def test_angular_all_pairs_batched_vs_oracle(sm):
    """Abundance-tracking collections in ONE launch: the u64 sums of the common hashes' abundance products and the sums of
    squares (csrc/compare_ext.hip), sqrt / acos on the host -- == the oracle's restatement of minhash.rs:635-680 for every
    pair, f64 bit for bit; abundances beyond 32 bits take the wide multiply."""
    from sourmash_amd.compare import compare_all_pairs, angular_matrix
    from sourmash_amd.synth import synth_sketches
    sk = synth_sketches(1000, seed=99, pool_size=20_000, keep_one_in=8)         # ~2,500 hashes each, planted edge rows
    def abund(h):
        return 1 + (h % np.uint64(7)) * (h % np.uint64(11))
    mhs, omhs = [], []
    for i, a in enumerate(sk):
        mh = sm.MinHash(0, 31, scaled=1000, track_abundance=True)
        ab = abund(a)
        mh.set_abundances(dict(zip(a.tolist(), ab.tolist())))
        mhs.append(mh)
        omhs.append(_oracle_sketch(a, scaled=1000, abunds=ab))
    want = oracle.similarity_matrix(omhs, ignore_abundance=False, nthreads=oracle.usable_cpus())
    got = compare_all_pairs([sm.SourmashSignature(mh, name=str(i)) for i, mh in enumerate(mhs)], ignore_abundance=False)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert np.array_equal(got, angular_matrix(mhs))
    for i, j in ((0, 1), (5, 700), (998, 999)):
        assert got[i, j] == mhs[i].angular_similarity(mhs[j]) == mhs[j].similarity(mhs[i])
    # ignore_abundance: the flat Jaccard matrix of the same hashes
    flat = compare_all_pairs([sm.SourmashSignature(mh, name=str(i)) for i, mh in enumerate(mhs[:200])], ignore_abundance=True)
    wc, wj = oracle.compare_all_pairs(*oracle.make_csr(sk[:200]), nthreads=4)
    assert np.array_equal(flat.view(np.uint64), wj.view(np.uint64))
    # abundances that do not fit 32 bits: products wrap in u64 like the reference's (release build) arithmetic
    big, obig = [], []
    for i, a in enumerate(sk[:40]):
        ab = (abund(a) << np.uint64(29 + i % 5)) + np.uint64(i)
        mh = sm.MinHash(0, 31, scaled=1000, track_abundance=True)
        mh.set_abundances(dict(zip(a.tolist(), ab.tolist())))
        big.append(mh)
        obig.append(_oracle_sketch(a, scaled=1000, abunds=ab))
    assert max(int(v) for v in big[4].hashes.values()) >= 2**32
    w2 = oracle.similarity_matrix(obig, ignore_abundance=False, nthreads=4)
    assert np.array_equal(angular_matrix(big).view(np.uint64), w2.view(np.uint64))


def test_mixed_scaled_and_jaccard_ani_batched_vs_oracle(sm):
    """compare.py:14-64 on a list with three scaled values: every pair at ITS coarser scaled (minhash.rs:688-696), served
    by one launch per scaled value; and return_ani: jaccard_ani (minhash.py:749-785) on whole arrays.  Both == the oracle
    pair by pair, f64 bit for bit (the ANI through the oracle's restatement of distance_utils.py:349-407, itself pinned to
    the reference module's outputs in tests/test_distance_utils.py)."""
    from sourmash_amd.compare import compare_serial
    from sourmash_amd.synth import synth_sketches
    sk = synth_sketches(120, seed=5, pool_size=40_000, keep_one_in=4, planted=False)      # ~10,000 hashes at scaled 1000
    sigs, omhs = [], []
    for i, a in enumerate(sk):
        s = (1000, 2000, 4000)[i % 3]
        mh = sm.MinHash(0, 31, scaled=1000)
        mh.add_many(a)
        if s != 1000:
            mh = mh.downsample(scaled=s)
        sigs.append(sm.SourmashSignature(mh, name=str(i)))
        omhs.append(_oracle_sketch(np.asarray(mh._mins_array()), scaled=s))
    with pytest.raises(ValueError) as e:
        compare_serial(sigs, True)
    assert "mismatch in scaled" in str(e.value)
    got = compare_serial(sigs, True, downsample=True)
    want = oracle.similarity_matrix(omhs, ignore_abundance=True, downsample=True, nthreads=oracle.usable_cpus())
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert got[0, 3] == sigs[0].jaccard(sigs[3]) and got[0, 1] == sigs[0].minhash.downsample(scaled=2000).jaccard(sigs[1].minhash)
    # Jaccard ANI, mixed scaled
    ani = compare_serial(sigs, True, downsample=True, return_ani=True)
    trusted = [s.minhash.size_is_accurate() for s in sigs]
    for i in range(len(sigs)):
        for j in range(i + 1, len(sigs)):
            s = max(sigs[i].minhash.scaled, sigs[j].minhash.scaled)
            a, b = omhs[i].downsample_scaled(s), omhs[j].downsample_scaled(s)
            jac = a.similarity(b, ignore_abundance=True)
            n_kmers = round((len(a) + len(b)) / 2 * s)
            dist, err = oracle.jaccard_to_distance(jac, 31, n_kmers)
            w = 0.0 if (err > 1e-4 or not (trusted[i] and trusted[j])) else 1 - dist
            assert ani[i, j] == w == ani[j, i], (i, j)
    assert (np.diag(ani) == 1.0).all() and (ani > 0.9).sum() > 100
    # and the same entries through the object API on a sample
    for i, j in ((0, 1), (2, 119), (50, 51), (7, 8)):
        r = sigs[i].jaccard_ani(sigs[j], downsample=True).ani
        assert ani[i, j] == (0.0 if r is None else r)
    # one scaled value: a single launch + array arithmetic
    same = [s for i, s in enumerate(sigs) if i % 3 == 0]
    a1 = compare_serial(same, True, return_ani=True)
    for i, j in ((0, 1), (5, 30), (38, 39)):
        r = same[i].jaccard_ani(same[j]).ani
        assert a1[i, j] == (0.0 if r is None else r)


def test_avg_containment_ani_mixed_scaled_asks_the_downsampled_sketches(sm):
    """compare.py:141-176 with return_ani: the avg form goes through FracMinHashComparison, whose containment_ani calls ask
    size_is_accurate() of the sketches ALREADY downsampled to the pair's scaled (sketchcomparison.py:53-70,143-170) -- a sketch
    of ~250 hashes at scaled 1000 is trusted as given and not at scaled 4000 (~62 hashes); the containment / max forms ask
    the sketches as given (minhash.py:877-878,938-939).  Batched matrices == the per-pair object API, entry by entry."""
    from sourmash_amd.compare import compare_serial_avg_containment, compare_serial_containment, compare_serial_max_containment
    from sourmash_amd.sketchcomparison import FracMinHashComparison
    from sourmash_amd.synth import synth_sketches
    big = synth_sketches(12, seed=9, pool_size=40_000, keep_one_in=4, planted=False)       # ~10,000 hashes at scaled 1000
    sigs = []
    for i, a in enumerate(big):
        mh = sm.MinHash(0, 31, scaled=1000)
        mh.add_many(a if i % 2 == 0 else a[::40])                  # odd: ~250 hashes: trusted at scaled 1000 (p = 0.9986), not as ~62 at 4000 (0.89)
        if i % 3 == 2:
            mh = mh.downsample(scaled=4000)
        sigs.append(sm.SourmashSignature(mh, name=str(i)))
    as_given = [s.minhash.size_is_accurate() for s in sigs]
    at_4000 = [s.minhash.downsample(scaled=4000).size_is_accurate() for s in sigs]
    assert any(g and not d for g, d in zip(as_given, at_4000))       # the case the test is about
    avg = compare_serial_avg_containment(sigs, downsample=True, return_ani=True)
    n_masked_by_downsampling = 0
    for i in range(len(sigs)):
        for j in range(i + 1, len(sigs)):
            cmp = FracMinHashComparison(sigs[j].minhash, sigs[i].minhash)
            r = cmp.avg_containment_ani
            assert avg[i, j] == avg[j, i] == (0.0 if r is None else r), (i, j)
            if r is None and as_given[i] and as_given[j]:
                n_masked_by_downsampling += 1
    assert n_masked_by_downsampling > 0
    con = compare_serial_containment(sigs, downsample=True, return_ani=True)
    mx = compare_serial_max_containment(sigs, downsample=True, return_ani=True)
    for i in range(len(sigs)):
        for j in range(len(sigs)):
            if i == j:
                continue
            r = sigs[j].containment_ani(sigs[i], downsample=True).ani
            assert con[i, j] == (0.0 if r is None else r), (i, j)
            r = sigs[j].max_containment_ani(sigs[i], downsample=True).ani
            assert mx[i, j] == (0.0 if r is None else r), (i, j)


def test_abundance_join_and_walk_agree_on_ragged_collections(sm):
    """The abundance sums come from joins of per-block hash-sorted lists (csrc/abund_pairs.hip) -- or, for collections of 2^32 elements
    and with SMG_COMPARE_ABUND=walk, from the per-pair walk (csrc/compare_ext.hip).  Both against the oracle on a collection with
    empty sketches, one-hash sketches, a sketch holding every hash of the pool, duplicates, more sketches than one 64-sketch block
    and not a multiple of it, and on one with a core of hashes held by EVERY sketch (runs of exactly 64 entries per block, cut by
    the 4,096-entry staging area; 64-bit abundances whose products wrap), and on one whose hashes crowd into one hash slice of the
    list build (round 6: more entries than the slice merge holds in LDS -- ranked against the rows in memory); every hash-slice count
    the join can be cut into (SMG_ABUND_SLICES)."""
    import os, subprocess, sys
    from conftest import ROOT
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_compare as t\nt._abundance_ragged()\nt._abundance_core()\nt._abundance_skewed()\nprint('ok')\n" % (ROOT, os.path.join(ROOT, "tests")))
    for extra in ({}, {"SMG_COMPARE_ABUND": "walk"}, {"SMG_ABUND_SLICES": "1"}, {"SMG_ABUND_SLICES": "3"}, {"SMG_ABUND_SLICES": "16"}):
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **extra))
        assert p.returncode == 0 and p.stdout.strip().endswith("ok"), (extra, p.stdout[-1500:], p.stderr[-1500:])


def _abundance_ragged():
    import torch  # noqa: F401
    import sourmash_amd as sm
    from sourmash_amd.compare import angular_matrix
    rng = np.random.default_rng(31)
    pool = np.unique(rng.integers(1, 2**63, size=6000, dtype=np.uint64))
    rows = []
    for i in range(150):
        size = int(rng.choice([0, 1, 2, 40, 700, 3000]))
        rows.append(np.sort(rng.choice(pool, size=min(size, len(pool)), replace=False)))
    rows[7] = pool.copy()                                          # holds every hash: meets everybody everywhere
    rows[8] = rows[9].copy()                                       # duplicates
    rows[64] = pool[::2].copy()                                    # first sketch of the second block
    rows[149] = pool[-3:].copy()
    mhs, omhs = [], []
    for i, a in enumerate(rows):
        ab = (a % np.uint64(13)) * (a % np.uint64(5)) + np.uint64(1 + i % 3)
        mh = sm.MinHash(0, 31, scaled=1, track_abundance=True)
        if len(a):
            mh.set_abundances(dict(zip(a.tolist(), ab.tolist())))
        mhs.append(mh)
        omhs.append(_oracle_sketch(a, scaled=1, abunds=ab))
    want = oracle.similarity_matrix(omhs, ignore_abundance=False, nthreads=oracle.usable_cpus())
    got = angular_matrix(mhs)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


def _abundance_skewed():
    import torch  # noqa: F401
    import sourmash_amd as sm
    from sourmash_amd.compare import angular_matrix
    rng = np.random.default_rng(33)
    pool = np.unique(rng.integers(1, 20_000, size=5000, dtype=np.uint64))          # everything far below the one large hash
    rows = [np.sort(rng.choice(pool, size=int(rng.integers(100, 200)), replace=False)) for _ in range(70)]
    rows[3] = np.array([2**63 + 12345], dtype=np.uint64)                           # stretches the hash range: the others share slice 0
    rows[69] = np.concatenate([rows[69], np.array([2**63 + 12345], dtype=np.uint64)])
    mhs, omhs = [], []
    for i, a in enumerate(rows):
        ab = (a % np.uint64(9)) + np.uint64(1 + i % 4)
        if i % 5 == 0:
            ab = ab << np.uint64(31)                                               # the wide form's lists as well
        mh = sm.MinHash(0, 31, scaled=1, track_abundance=True)
        mh.set_abundances(dict(zip(a.tolist(), ab.tolist())))
        mhs.append(mh)
        omhs.append(_oracle_sketch(a, scaled=1, abunds=ab))
    want = oracle.similarity_matrix(omhs, ignore_abundance=False, nthreads=oracle.usable_cpus())
    got = angular_matrix(mhs)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    narrow = angular_matrix(mhs[1:5] + mhs[6:10] + mhs[11:15] + mhs[16:20] + mhs[21:25] + mhs[26:30] + mhs[31:35] + mhs[36:40] + mhs[41:45] +
                            mhs[46:50] + mhs[51:55] + mhs[56:60] + mhs[61:65] + mhs[66:70])
    keep = [i for i in range(70) if i % 5]
    assert np.array_equal(narrow.view(np.uint64), want[np.ix_(keep, keep)].view(np.uint64))


def _abundance_core():
    import torch  # noqa: F401
    import sourmash_amd as sm
    from sourmash_amd.compare import angular_matrix
    rng = np.random.default_rng(32)
    core = np.unique(rng.integers(1, 2**63, size=300, dtype=np.uint64))
    pool = np.unique(rng.integers(1, 2**63, size=5000, dtype=np.uint64))
    mhs, omhs = [], []
    for i in range(130):                                           # two full blocks and one of two sketches
        a = np.unique(np.concatenate([core, rng.choice(pool, size=500, replace=False)]))
        if i % 3 == 0:                                             # products and sums that wrap 2^64 (minhash.rs:635-680, release build)
            ab = (a | np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) | np.uint64(1)
        else:
            ab = (a % np.uint64(1000)) + np.uint64(1)
        mh = sm.MinHash(0, 31, scaled=1, track_abundance=True)
        mh.set_abundances(dict(zip(a.tolist(), ab.tolist())))
        mhs.append(mh)
        omhs.append(_oracle_sketch(a, scaled=1, abunds=ab))
    want = oracle.similarity_matrix(omhs, ignore_abundance=False, nthreads=oracle.usable_cpus())
    got = angular_matrix(mhs)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
