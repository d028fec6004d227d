"""The MinHash object API on the GPU path, following the reference's own test suite
(tests/test_minhash.py -- cited per case).  These are the behaviours a user of `sourmash.MinHash`
relies on; values are the reference's.  Run with -m gpu."""
import math
import pickle

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sm():
    import torch  # noqa: F401
    import sourmash_amd
    assert sourmash_amd.gpu_available()
    return sourmash_amd


@pytest.fixture(params=[True, False])
def track_abundance(request):
    return request.param


def _scaled_for(sm, max_hash):
    from sourmash_amd.minhash import _get_scaled_for_max_hash
    return _get_scaled_for_max_hash(max_hash)


# ---- basic sketching ---------------------------------------------------------------------------------
def test_basic_dna(sm, track_abundance):
    # test_minhash.py:98-112
    mh = sm.MinHash(1, 4, track_abundance=track_abundance)
    assert mh.moltype == "DNA"
    mh.add_sequence("ATGC")
    a = list(mh.hashes)
    mh.add_sequence("GCAT")                 # reverse complement -> same canonical k-mer
    b = list(mh.hashes)
    assert a == b == [12415348535738636339] and len(b) == 1


def test_div_zero_and_requirements(sm, track_abundance):
    # :115-160
    mh = sm.MinHash(1, 4, track_abundance=track_abundance)
    mh2 = mh.copy_and_clear()
    mh.add_sequence("ATGC")
    assert mh.similarity(mh2) == 0 and mh2.similarity(mh) == 0
    s1 = sm.MinHash(0, 4, scaled=1, track_abundance=track_abundance)
    s2 = s1.copy_and_clear()
    s1.add_sequence("ATGC")
    assert s1.contained_by(s2) == 0 and s2.contained_by(s1) == 0
    n1, n2 = sm.MinHash(1, 4), sm.MinHash(1, 4)
    n1.add_sequence("ATGC")
    for a, b in ((n1, n2), (n1, s1), (s1, n1)):
        with pytest.raises(TypeError) as e:
            a.contained_by(b)
        assert "Error: can only calculate containment for scaled MinHashes" in str(e.value)


def test_bytes_and_long_seqs(sm, track_abundance):
    # :163-203
    a, b = sm.MinHash(1, 4, track_abundance=track_abundance), sm.MinHash(1, 4, track_abundance=track_abundance)
    a.add_sequence("ATGC")
    b.add_sequence(b"ATGC")
    assert list(a.hashes) == list(b.hashes)
    mh = sm.MinHash(0, 21, scaled=10, track_abundance=track_abundance)            # every k-mer spans an N
    seq = "ACGTN" * 100000
    assert mh.seq_to_hashes(seq, force=True) == []
    mh.add_sequence(seq, force=True)
    assert len(mh.hashes) == 0


def test_seq_to_hashes_and_kmers(sm, track_abundance):
    # :206-219, 265-282, 2544-2627
    mh = sm.MinHash(0, 4, scaled=1, track_abundance=track_abundance)
    seq = "ATGAGAGACGATAGACAGATGACC"
    mh.add_sequence(seq)
    assert set(mh.hashes) == set(mh.seq_to_hashes(seq))
    bad = "ATGAGAGACGATAGACAGATGACN"
    hs = mh.seq_to_hashes(bad, force=True, bad_kmers_as_zeroes=True)
    assert len(hs) == len(bad) - 4 + 1 and hs[-1] == 0 and all(hs[:-1])
    with pytest.raises(ValueError):
        mh.seq_to_hashes(bad, bad_kmers_as_zeroes=True)
    m21 = sm.MinHash(0, 21, scaled=1)
    s = "TGCCGCCCAGCACCGGGTGACTAGGTTGAGCCATGATTAACCTGCAATGA"
    pairs = list(m21.kmers_and_hashes(s))
    assert len(pairs) == len(s) - 21 + 1
    for kmer, h in pairs:
        single = sm.MinHash(0, 21, scaled=1)
        single.add_sequence(kmer)
        assert list(single.hashes) == [h] and kmer in s
    badseq = "NTGCGAGTGTTGAAGTTCGGCGGTACATCAGTGGC"
    with pytest.raises(ValueError) as e:
        list(sm.MinHash(0, 31, scaled=1).kmers_and_hashes(badseq))
    assert "invalid DNA character in input k-mer: NTGCGAGTGT" in str(e.value)
    forced = list(sm.MinHash(0, 31, scaled=1).kmers_and_hashes(badseq, force=True))
    assert forced[0][1] is None and all(h is not None for _, h in forced[1:])


def test_dna_bad_input(sm, track_abundance):
    # :711-765
    mh = sm.MinHash(1, 4, track_abundance=track_abundance)
    with pytest.raises(ValueError) as e:
        mh.add_sequence("ATGR")
    assert "invalid DNA character in input k-mer: ATGR" in str(e.value)
    with pytest.raises(ValueError):
        sm.MinHash(1, 4, track_abundance=track_abundance).add_sequence("ATGCN")      # second k-mer is bad
    f = sm.MinHash(100, 4, track_abundance=track_abundance)
    assert len(f) == 0
    f.add_sequence("ATGN", True)
    assert len(f) == 0
    f.add_sequence("AATGN", True)
    assert len(f) == 1
    f.add_sequence("AATG", True)
    assert len(f) == 1                                                                 # same k-mer (rc of CATT)
    a, b = sm.MinHash(3, 4, track_abundance=track_abundance), sm.MinHash(3, 4, track_abundance=track_abundance)
    a.add_sequence("TGCCGCCCAGCACCGGGTGACTAGG".lower())
    b.add_sequence("TGCCGCCCAGCACCGGGTGACTAGG")
    assert list(a.hashes) == list(b.hashes)
    short = sm.MinHash(1, 31)
    short.add_sequence("ACGT")
    assert len(short) == 0                                                             # :1232-1236


# ---- scaled / num plumbing ----------------------------------------------------------------------------
def test_scaled_and_constructor_rules(sm, track_abundance):
    # :475-545
    scaled = _scaled_for(sm, 35)
    mh = sm.MinHash(0, 4, track_abundance=track_abundance, scaled=scaled)
    assert mh._max_hash == 35
    for h in (10, 20, 30):
        mh.add_hash(h)
    mh.add_hash(40)
    mh.add_hash(36)
    assert sorted(mh.hashes) == [10, 20, 30]
    with pytest.raises(ValueError):
        sm.MinHash(0, 4, track_abundance=track_abundance)
    with pytest.raises(ValueError):
        sm.MinHash(0, 4, max_hash=35, scaled=5)
    with pytest.raises(ValueError):
        sm.MinHash(2, 4, scaled=2)
    with pytest.raises(ValueError) as e:
        sm.MinHash(2, 4).downsample(scaled=100000000)
    assert "cannot downsample a num MinHash using scaled" in str(e.value)
    from sourmash_amd.minhash import _get_max_hash_for_scaled
    assert _scaled_for(sm, _get_max_hash_for_scaled(100000)) == 100000
    assert sm.MinHash(0, 4, scaled=1000).scaled == 1000                               # :1644-1651
    assert sm.MinHash(500, 4).scaled == 0


def test_size_limit_and_len(sm, track_abundance):
    # :458-472, 806-835
    mh = sm.MinHash(3, 4, track_abundance=track_abundance)
    for h in (10, 20, 30):
        mh.add_hash(h)
    assert sorted(mh.hashes) == [10, 20, 30]
    mh.add_hash(5)
    assert sorted(mh.hashes) == [5, 10, 20]
    a = sm.MinHash(20, 10, track_abundance=track_abundance)
    for i in range(0, 40, 2):
        a.add_hash(i)
    assert len(a) == 20 and sorted(a.hashes) == list(range(0, 40, 2))
    a.add_hash(9227159859419181011)                                                   # > C long
    b = sm.MinHash(25, 10, track_abundance=track_abundance)
    b.add_hash(9227159859419181011)
    assert 9227159859419181011 in b.hashes


# ---- similarity values ---------------------------------------------------------------------------------
def test_jaccard_and_angular_values(sm):
    # :548-640
    s50, s100 = _scaled_for(sm, 50), _scaled_for(sm, 100)
    a, b = sm.MinHash(0, 20, scaled=s50), sm.MinHash(0, 20, scaled=s50)
    a.add_many([1, 3, 5, 8])
    b.add_many([1, 3, 5, 6, 8, 10])
    assert a.similarity(b) == 4.0 / 6.0
    c = sm.MinHash(0, 20, scaled=s100)
    c.add_many([1, 3, 5, 6, 8, 10, 70])
    a2 = sm.MinHash(0, 20, scaled=s50)
    a2.add_many([1, 3, 5, 8, 70])                       # 70 > max_hash 50: not kept
    assert a2.similarity(c, downsample=True) == 4.0 / 6.0
    aa, ab = sm.MinHash(0, 20, scaled=s50, track_abundance=True), sm.MinHash(0, 20, scaled=s50, track_abundance=True)
    aa.set_abundances({1: 5, 3: 3, 5: 2, 8: 2})
    ab.set_abundances({1: 3, 3: 2, 5: 1, 6: 1, 8: 1, 10: 1})
    assert round(aa.similarity(ab), 4) == round(1 - 2 * math.acos(0.9356) / math.pi, 4) == 0.7703
    ba, bb = sm.MinHash(0, 20, scaled=s100, track_abundance=True), sm.MinHash(0, 20, scaled=s100, track_abundance=True)
    ba.set_abundances({1: 5, 3: 3, 5: 2, 8: 2, 70: 70})
    bb.set_abundances({1: 3, 3: 2, 5: 1, 6: 1, 8: 1, 10: 1, 70: 70})
    assert round(ba.similarity(bb), 4) == 0.9728
    assert ba.similarity(bb, ignore_abundance=True) == 5.0 / 7.0
    ca = sm.MinHash(0, 20, scaled=s50, track_abundance=True)
    ca.set_abundances({1: 5, 3: 3, 5: 2, 8: 2, 70: 70})
    assert round(ca.similarity(bb, downsample=True), 4) == 0.7703
    assert ca.similarity(bb, downsample=True, ignore_abundance=True) == 4.0 / 6.0
    for x, y in ((ca, bb), (a, c)):
        with pytest.raises(ValueError) as e:
            x.similarity(y, ignore_abundance=True)
        assert "mismatch in scaled; comparison fail" in str(e.value)
    flat = sm.MinHash(0, 20, scaled=s50)
    flat.add_many([1, 3])
    with pytest.raises(TypeError) as e:
        aa.angular_similarity(flat)
    assert "requires both sketches to track hash abundance" in str(e.value)


def test_similarity_of_sequences(sm, track_abundance):
    # :768-784
    a, b = sm.MinHash(20, 10, track_abundance=track_abundance), sm.MinHash(20, 10, track_abundance=track_abundance)
    a.add_sequence("TGCCGCCCAGCACCGGGTGACTAGGTTGAGCCATGATTAACCTGCAATGA")
    b.add_sequence("TGCCGCCCAGCACCGGGTGACTAGGTTGAGCCATGATTAACCTGCAATGA")
    assert round(a.similarity(b), 3) == 1.0 == round(b.similarity(a), 3)
    b.add_sequence("TGCCGCCCAGCACCGGGTGACTAGGTTGAGCCATGATTAACCTGCAATGA")
    assert round(a.similarity(b), 3) == 1.0
    b.add_sequence("GATTGGTGCACACTTAACTGGGTGCCGCGCTGGTGCTGATCCATGAAGTT")
    assert a.similarity(b) >= 0.3 and b.similarity(a) >= 0.3


def test_count_common_and_errors(sm, track_abundance):
    # :845-903
    a, b = sm.MinHash(20, 10, track_abundance=track_abundance), sm.MinHash(20, 10, track_abundance=track_abundance)
    for i in range(0, 40, 2):
        a.add_hash(i)
    for i in range(0, 80, 4):
        b.add_hash(i)
    assert a.count_common(b) == 10 == b.count_common(a)
    for x, y in ((sm.MinHash(20, 5), sm.MinHash(20, 6)),
                 (sm.MinHash(20, 5, seed=1), sm.MinHash(20, 5, seed=2)),
                 (sm.MinHash(0, 5, scaled=_scaled_for(sm, 1)), sm.MinHash(0, 5, scaled=_scaled_for(sm, 2)))):
        with pytest.raises(ValueError):
            x.count_common(y)
    with pytest.raises(TypeError):
        a.count_common(set())
    with pytest.raises(ValueError):
        a.downsample(num=30)
    small = sm.MinHash(10, 10, track_abundance=track_abundance)                      # :911-934
    for i in range(0, 80, 4):
        small.add_hash(i)
    assert a.count_common(small) == 10
    with pytest.raises(TypeError):
        a.jaccard(small)
    a10 = a.downsample(num=10)
    assert a10.jaccard(small) == 0.5 == small.jaccard(a10)


# ---- merging / concatenation ----------------------------------------------------------------------------
def test_merge_family(sm, track_abundance):
    # :937-1120
    with pytest.raises(TypeError):
        sm.MinHash(20, 10).merge(set())
    a, b = sm.MinHash(100, 10, track_abundance=track_abundance), sm.MinHash(100, 10, track_abundance=track_abundance)
    for i in range(0, 40, 2):
        a.add_hash(i)
    for i in range(0, 80, 4):
        b.add_hash(i)
    c, d = a.__copy__(), b.__copy__()
    c.merge(b)
    d.merge(a)
    assert sorted(c.hashes.items()) == sorted(d.hashes.items()) and round(c.similarity(d), 3) == 1.0
    assert len(c) == 30                                                              # :1003-1018 (distinct union)
    e = sm.MinHash(0, 10, scaled=1, track_abundance=track_abundance)
    f = e.__copy__()
    f.merge(e)
    assert len(f) == 0
    g = sm.MinHash(20, 10, track_abundance=track_abundance)
    h = sm.MinHash(10, 10, track_abundance=track_abundance)
    for i in range(0, 40, 2):
        g.add_hash(i)
    for i in range(0, 80, 4):
        h.add_hash(i)
    g2, h2 = g.__copy__(), h.__copy__()
    g2.merge(h)                                                                      # asymmetric num: :1036-1068
    h2.merge(g)
    assert len(g2) == 20 and len(h2) == 10
    assert sorted(h2.hashes) == list(range(0, 20, 2))
    i2 = g.__copy__()
    i2 += h
    assert sorted(i2.hashes) == sorted(g2.hashes)
    for x, y in ((sm.MinHash(20, 5), sm.MinHash(20, 6)), (sm.MinHash(20, 5, seed=1), sm.MinHash(20, 5, seed=2)),
                 (sm.MinHash(0, 5, scaled=_scaled_for(sm, 5)), sm.MinHash(0, 5, scaled=_scaled_for(sm, 10)))):
        with pytest.raises(ValueError):
            x.merge(y)
        with pytest.raises(ValueError):
            x += y
        with pytest.raises(ValueError):
            x.similarity(y)


def test_addition_and_or(sm):
    # :2071-2160, 2372-2400
    with pytest.raises(TypeError) as e:
        sm.MinHash(10, 21) + sm.MinHash(20, 21)
    assert "incompatible num values: self=10 other=20" in str(e.value)
    m1, m2 = sm.MinHash(10, 21, track_abundance=True), sm.MinHash(10, 21, track_abundance=True)
    m1.set_abundances({0: 1})
    m2.set_abundances({0: 3})
    m3 = m1 + m2
    assert dict(m3.hashes) == {0: 4} and dict(m1.hashes) == {0: 1}
    m1 += m2
    assert dict(m1.hashes) == {0: 4} and dict(m2.hashes) == {0: 3}
    n1, n2 = sm.MinHash(10, 21), sm.MinHash(10, 21)
    n1.add_hash(0)
    n2.add_hash(0)
    assert dict((n1 + n2).hashes) == {0: 1}
    p, q = sm.MinHash(0, 21, scaled=1), sm.MinHash(0, 21, scaled=1)
    p.add_many([0, 1, 5])
    q.add_many([0, 2, 7])
    assert (p + q) == (q + p) == (p | q)
    assert sorted((p | q).hashes) == [0, 1, 2, 5, 7]


def test_intersections(sm):
    # :2172-2330
    a, b = sm.MinHash(0, 21, scaled=1), sm.MinHash(0, 21, scaled=1)
    a.add_many([0, 1])
    b.add_many([0, 2])
    i = a.intersection(b)
    assert len(i) == 1 and 0 in i.hashes and (a & b) == i
    with pytest.raises(TypeError) as e:
        sm.MinHash(0, 21, scaled=1, track_abundance=True).intersection(sm.MinHash(0, 21, scaled=1, track_abundance=True))
    assert str(e.value) == "can only intersect flat MinHash objects"
    with pytest.raises(ValueError) as e:
        sm.MinHash(500, 21).intersection(sm.MinHash(500, 31))
    assert str(e.value) == "different ksizes cannot be compared"
    with pytest.raises(TypeError) as e:
        a.intersection(set())
    assert str(e.value) == "can only intersect MinHash objects"
    n1, n2 = sm.MinHash(20, 21), sm.MinHash(20, 21)
    for k in range(100):
        n1.add_hash(k)
    for k in range(0, 100, 2):
        n2.add_hash(k)
    assert len(n1) == len(n2) == 20
    assert n1.intersection(n2) == n2.intersection(n1)
    assert all(k in n1.hashes and k in n2.hashes for k in n1.intersection(n2).hashes)
    assert n1.intersection_and_union_size(n2) == (10, 20)
    s1, s2 = sm.MinHash(0, 21, scaled=100), sm.MinHash(0, 21, scaled=100)
    for k in range(100):
        s1.add_hash(k)
    for k in range(0, 200, 2):
        s2.add_hash(k)
    assert s1.intersection(s2) == s2.intersection(s1)
    assert s1.intersection_and_union_size(s2) == (50, 150)
    with pytest.raises(TypeError) as e:
        sm.MinHash(0, 21, scaled=1).intersection_and_union_size(sm.MinHash(0, 31, scaled=1))
    assert "incompatible MinHash objects" in str(e.value)


# ---- abundance bookkeeping ------------------------------------------------------------------------------
def test_abundance_family(sm):
    # :1267-1520
    a, b = sm.MinHash(20, 5, track_abundance=True), sm.MinHash(20, 5, track_abundance=False)
    a.add_sequence("AAAAA")
    a.add_sequence("AAAAA")
    assert dict(a.hashes) == {2110480117637990133: 2}
    b.add_sequence("AAAAA")
    b.add_sequence("GGGGG")
    assert a.count_common(b) == 1 == b.count_common(a)
    assert sorted(b.hashes) == [2110480117637990133, 10798773792509008305]
    c = sm.MinHash(20, 5, track_abundance=True)
    c.add_hash_with_abundance(10, 1)
    assert dict(c.hashes) == {10: 1}
    c.add_hash_with_abundance(20, 2)
    c.add_hash_with_abundance(10, 2)
    assert dict(c.hashes) == {10: 3, 20: 2}
    with pytest.raises(RuntimeError) as e:
        sm.MinHash(20, 5).add_hash_with_abundance(10, 1)
    assert "track_abundance=True when constructing" in e.value.args[0]
    c.clear()
    assert dict(c.hashes) == {}
    with pytest.raises(RuntimeError):
        sm.MinHash(20, 10).set_abundances({1: 3, 2: 4})
    d = sm.MinHash(20, 5, track_abundance=True)
    d.add_hash(10)
    d.set_abundances({20: 2})                               # default clear=True
    assert dict(d.hashes) == {20: 2}
    d.set_abundances({10: 1, 20: 3}, clear=False)
    assert dict(d.hashes) == {10: 1, 20: 5}
    d.set_abundances({20: 0}, clear=False)                  # abundance 0 removes
    assert dict(d.hashes) == {10: 1}
    with pytest.raises(ValueError):
        d.set_abundances({1: -1})
    e1 = sm.MinHash(2, 10, track_abundance=True)
    e1.set_abundances({1: 3, 2: 4, 3: 5})                  # num caps it
    assert dict(e1.hashes) == {1: 3, 2: 4}
    f = sm.MinHash(1, 4, track_abundance=True)
    f.add_hash(10)
    f.track_abundance = False
    assert not f.track_abundance
    with pytest.raises(RuntimeError):
        f.track_abundance = True                            # not empty
    g = sm.MinHash(5, 4)
    g.track_abundance = True
    g.set_abundances({1: 5})
    assert dict(g.hashes) == {1: 5}
    big = 2**63 - 1
    h = sm.MinHash(5, 4, track_abundance=True)
    h.set_abundances({7: big})
    assert dict(h.hashes) == {7: big}
    i = sm.MinHash(0, 4, scaled=1, track_abundance=True)
    i.set_abundances({1: 2, 5: 4, 9: 6})
    assert (i.sum_abundances, i.mean_abundance, i.median_abundance) == (12, 4.0, 4.0)
    assert round(i.std_abundance, 4) == 1.633


def test_add_remove_many_flatten_inflate(sm, track_abundance):
    # :1700-1850
    a = sm.MinHash(0, 10, scaled=1, track_abundance=track_abundance)
    a.add_many(list(range(0, 100, 2)))
    assert len(a) == 50 and all(c % 2 == 0 for c in a.hashes)
    a.remove_many(list(range(0, 100, 3)))
    assert len(a) == 33 and all(c % 6 != 0 for c in a.hashes)
    r = sm.MinHash(0, 10, scaled=1)
    r.add_many(range(0, 100, 3))
    b = sm.MinHash(0, 10, scaled=1, track_abundance=track_abundance)
    b.add_many(range(0, 100, 2))
    b.remove_many(r)
    assert sorted(b.hashes) == sorted(a.hashes)
    b.add_many(r)
    assert len(b) == 33 + 34
    with pytest.raises(RuntimeError):
        b.hashes[5] = 2
    m = sm.MinHash(0, 10, scaled=1, track_abundance=True)
    m.set_abundances({10: 2, 20: 3, 30: 4})
    flat = m.flatten()
    assert not flat.track_abundance and sorted(flat.hashes) == [10, 20, 30] and flat.flatten() is flat
    sub = sm.MinHash(0, 10, scaled=1)
    sub.add_many([10, 30, 40])
    inflated = sub.inflate(m)
    assert dict(inflated.hashes) == {10: 2, 30: 4}                                   # 40 not in m: dropped
    with pytest.raises(ValueError):
        m.inflate(m)
    k = sm.MinHash(0, 4, scaled=1, track_abundance=track_abundance)
    k.add_kmer("ATGC")
    assert len(k) == 1
    with pytest.raises(ValueError) as e:
        k.add_kmer("ATGCG")
    assert "kmer to add is not 4 in length" in str(e.value)


# ---- copies, pickles, frozen ------------------------------------------------------------------------------
def test_copy_pickle_frozen(sm, track_abundance):
    # :787-803, 1560-1640, 2460-2540
    a = sm.MinHash(20, 21, track_abundance=track_abundance)
    a.add_hash(5)
    b = a.__copy__()
    assert a == b
    a.add_hash(6)
    assert a != b
    for mh in (sm.MinHash(0, 6, scaled=_scaled_for(sm, 20)), sm.MinHash(0, 6, scaled=1000, track_abundance=track_abundance),
               sm.MinHash(500, 21, seed=7, track_abundance=track_abundance)):
        for k in range(0, 100, 2):
            mh.add_hash(k)
        back = pickle.loads(pickle.dumps(mh))
        assert back == mh and back.ksize == mh.ksize and back.seed == mh.seed and back.scaled == mh.scaled
        assert back.track_abundance == mh.track_abundance and back.num == mh.num
    c = a.copy_and_clear()
    assert len(c) == 0 and c.ksize == a.ksize and c.track_abundance == a.track_abundance and c.num == a.num
    s = sm.MinHash(0, 10, scaled=_scaled_for(sm, 20), track_abundance=track_abundance)
    for k in range(0, 40, 2):
        s.add_hash(k)
    sc = s.copy_and_clear()
    assert len(sc) == 0 and sc._max_hash == s._max_hash == 20 and sc.scaled == s.scaled
    fz = a.to_frozen()
    assert fz == a and isinstance(fz, sm.FrozenMinHash)
    for op in (lambda: fz.add_hash(1), lambda: fz.add_sequence("ATGC"), lambda: fz.clear(), lambda: fz.merge(a),
               lambda: fz.add_many([1]), lambda: fz.remove_many([1])):
        with pytest.raises(TypeError):
            op()
    mut = fz.to_mutable()
    mut.add_hash(99)
    assert mut != fz and not isinstance(mut, sm.FrozenMinHash)
    assert fz.to_frozen() is fz and fz.copy() is fz
    a.into_frozen()
    assert isinstance(a, sm.FrozenMinHash)


def test_downsample_rules(sm, track_abundance):
    # :1965-2010
    n = sm.MinHash(10, 21, track_abundance=track_abundance)
    for k in range(20):
        n.add_hash(k)
    d = n.downsample(num=5)
    assert sorted(d.hashes) == [0, 1, 2, 3, 4] and d.num == 5
    s = sm.MinHash(0, 21, scaled=_scaled_for(sm, 50), track_abundance=track_abundance)
    for k in (5, 10, 15, 20, 25, 30, 35, 40, 45, 50):
        s.add_hash(k)
    coarser = s.downsample(scaled=_scaled_for(sm, 25))
    assert sorted(coarser.hashes) == [5, 10, 15, 20, 25]
    for kw in ({}, {"num": 5, "scaled": 5}):
        with pytest.raises(ValueError):
            s.downsample(**kw)
    with pytest.raises(ValueError):
        s.downsample(num=5)
    with pytest.raises(ValueError):
        coarser.downsample(scaled=_scaled_for(sm, 50))


# ---- containment family and ANI ---------------------------------------------------------------------------
def test_containment_family(sm):
    # :2403-2460, 2630-2660
    from sourmash_amd.minhash import MinHash
    a, b = MinHash(0, 21, scaled=1), MinHash(0, 21, scaled=1)
    a.add_many((1, 2, 3, 4))
    b.add_many((1, 2, 3, 4, 5, 6))
    assert a.contained_by(b) == 1.0 and b.contained_by(a) == 4 / 6
    assert a.max_containment(b) == 1.0 == b.max_containment(a)
    assert a.avg_containment(b) == (1.0 + 4 / 6) / 2 == b.avg_containment(a)
    e = MinHash(0, 21, scaled=1)
    assert a.max_containment(e) == 0 == e.max_containment(a) and a.avg_containment(e) == 0
    c, d = MinHash(0, 21, scaled=1), MinHash(0, 21, scaled=1)
    c.add_many((1, 2, 3, 4))
    d.add_many((1, 2, 5, 6))
    assert c.contained_by(d) == 0.5 == c.max_containment(d) == c.avg_containment(d)
    x, y = MinHash(1, 21), MinHash(1, 21)
    for fn in ("contained_by", "max_containment", "avg_containment", "containment_ani", "jaccard_ani"):
        with pytest.raises(TypeError):
            getattr(x, fn)(y)
    assert MinHash(0, 21, scaled=1000).unique_dataset_hashes == 0
    u = MinHash(0, 21, scaled=100)
    u.add_many(range(50))
    assert u.unique_dataset_hashes == 5000
    with pytest.raises(TypeError):
        x.unique_dataset_hashes


def test_ani_on_real_sketches(sm):
    # the pipeline of :2700-2860 on fixtures we carry (47.fa / 63.fa, k=31 scaled=1000):
    # counts from the GPU, float layer on the host; cross-checked against the formulas directly.
    from conftest import golden
    from sourmash_amd.distance_utils import containment_to_distance, jaccard_to_distance
    a = sm.load_one_signature_from_json(golden("pairs", "47.fa.sig")).minhash
    b = sm.load_one_signature_from_json(golden("pairs", "63.fa.sig")).minhash
    c = a.contained_by(b)
    res = a.containment_ani(b, estimate_ci=True)
    want = containment_to_distance(c, 31, 1000, n_unique_kmers=len(a) * 1000, estimate_ci=True)
    assert (res.dist, res.dist_low, res.dist_high, res.p_nothing_in_common) == \
           (want.dist, want.dist_low, want.dist_high, want.p_nothing_in_common)
    assert 0.9 < res.ani < 1.0 and res.ani_low < res.ani < res.ani_high
    assert a.containment_ani(b, containment=c).ani == res.ani
    j = a.jaccard(b)
    jres = a.jaccard_ani(b)
    assert jres.dist == jaccard_to_distance(j, 31, 1000, n_unique_kmers=round((len(a) + len(b)) / 2 * 1000)).dist
    mres = a.max_containment_ani(b)
    assert mres.ani == max(a.containment_ani(b).ani, b.containment_ani(a).ani)
    assert a.avg_containment_ani(b) == (a.containment_ani(b).ani + b.containment_ani(a).ani) / 2
    a100 = a.downsample(scaled=2000)
    assert a100.containment_ani(b, downsample=True).ani == a100.containment_ani(b.downsample(scaled=2000)).ani
    assert a.size_is_accurate() and not sm.MinHash(0, 31, scaled=1000, mins=[1, 2, 3]).size_is_accurate()
    tiny = sm.MinHash(0, 31, scaled=1000, mins=[1, 2, 3])
    assert tiny.containment_ani(tiny).ani is None                                    # size estimate not trustworthy


def test_device_mirrors_follow_every_mutation(sm):
    """The per-pair entry points keep a device copy of each sketch between calls (device_ctx.hpp: mirror_of, keyed by the
    sketch's content generation).  Whatever changes a sketch -- add / remove / merge / clear / abundances / queued
    sequence / downsample, on the object itself or on a copy -- the next count_common / jaccard / angular answers for the
    NEW content, checked against the oracle's walk (minhash.rs:539-558,593-631)."""
    import numpy as np
    import oracle
    rng = np.random.default_rng(12)
    universe = rng.integers(1, 2**50, size=20_000, dtype=np.uint64)

    def fresh(idx, abund=False):
        mh = sm.MinHash(0, 31, scaled=1, track_abundance=abund)
        mh.add_many([int(x) for x in universe[idx]])
        return mh

    def check(a, b):
        ha, hb = np.array(sorted(a.hashes), dtype=np.uint64), np.array(sorted(b.hashes), dtype=np.uint64)
        c, u = oracle.intersection_size(ha, hb)
        assert a.count_common(b) == c and b.count_common(a) == c
        assert a.jaccard(b) == (c / u if u else 0.0)

    a, b = fresh(slice(0, 6000)), fresh(slice(3000, 9000))
    check(a, b)
    check(a, b)                                                    # second call: both mirrors are warm
    a.add_hash(int(universe[8000])); check(a, b)
    a.add_many([int(x) for x in universe[8500:8600]]); check(a, b)
    a.remove_many([int(x) for x in universe[3000:3500]]); check(a, b)
    b.merge(fresh(slice(0, 100))); check(a, b)
    c = a.copy()                                                   # the copy shares a's content (and its mirror) ...
    check(c, b)
    c.add_many([int(x) for x in universe[9000:9500]])              # ... until one of them changes
    check(c, b); check(a, b); check(a, c)
    d = a.downsample(scaled=4); e = b.downsample(scaled=4)
    check(d, e)
    a.clear(); check(a, b)
    a.add_many([int(x) for x in universe[100:200]]); check(a, b)
    q = sm.MinHash(0, 21, scaled=1)
    q2 = sm.MinHash(0, 21, scaled=1)
    seq = "ACGTTGCAAGCTTGCATCGATCGGATCGATTAGCTAGCTAGGATCGATCGATTAGC" * 3
    q.add_sequence(seq); q2.add_sequence(seq[:80])
    assert q.count_common(q2) == len(q2)
    q2.add_sequence(seq[40:])                                      # queued records settle on the next access: new content
    assert q.count_common(q2) == len(q2) == len(q)
    # abundances live in the mirror too: changing them changes the angular similarity
    x, y = fresh(slice(0, 3000), abund=True), fresh(slice(1000, 4000), abund=True)
    s0 = x.angular_similarity(y)
    assert s0 == x.angular_similarity(y)
    x.set_abundances({int(h): 7 for h in universe[1000:1500]}, clear=False)
    s1 = x.angular_similarity(y)
    hx = dict(x.hashes); hy = dict(y.hashes)
    prod = sum(v * hy[k] for k, v in hx.items() if k in hy)
    na, nb = math.sqrt(sum(v * v for v in hx.values())), math.sqrt(sum(v * v for v in hy.values()))
    assert s1 != s0 and s1 == 1.0 - 2.0 * math.acos(min(prod / (na * nb), 1.0)) / math.pi
    # many distinct sketches: more than the cache keeps would still be right (entries are evicted, least recently used first)
    many = [fresh(slice(i * 37, i * 37 + 500)) for i in range(40)]
    for i in range(0, 40, 3):
        check(many[i], many[(i * 7 + 1) % 40])


def test_native_copies_share_a_generation_without_sharing_a_fate(sm):
    """Copies made inside the library (signature_set_mh / signature_first_mh copy-construct, so the copy keeps the
    original's content generation and finds its device mirror) must not lose that mirror to the ORIGINAL's next change in
    the middle of a call: f = sig.minhash; mh changes; f.count_common(mh) first takes f's (shared) mirror, then sees mh let go
    of that very generation (device_ctx.hpp: drop_mirror parks the blocks until the call's kernels are on the stream).
    Counts against the oracle's walk (minhash.rs:539-558)."""
    import numpy as np
    import oracle
    rng = np.random.default_rng(99)
    universe = rng.integers(1, 2**50, size=12_000, dtype=np.uint64)

    def want(a, b):
        ha, hb = np.array(sorted(a.hashes), dtype=np.uint64), np.array(sorted(b.hashes), dtype=np.uint64)
        return oracle.intersection_size(ha, hb)[0]

    for with_abund in (False, True):
        mh = sm.MinHash(0, 31, scaled=1, track_abundance=with_abund)
        mh.add_many([int(x) for x in universe[:5000]])
        other = sm.MinHash(0, 31, scaled=1, track_abundance=with_abund)
        other.add_many([int(x) for x in universe[2500:7000]])
        assert mh.count_common(other) == want(mh, other)           # mh is mirrored now
        sig = sm.SourmashSignature(mh)
        f = sig.minhash                                            # native copy: same generation as mh
        mh.remove_many([int(x) for x in universe[:2000]])          # mh is no superset of f any more ...
        mh.add_many([int(x) for x in universe[8000:9000]])         # ... and not a subset either
        c = want(f, mh)
        assert c == 3000
        assert f.count_common(mh) == c                             # f's mirror first, then mh drops "its" old generation
        assert mh.count_common(f) == c
        g = sig.minhash
        mh.add_many([int(x) for x in universe[9000:9100]])
        assert g.jaccard(mh) == c / (len(g) + len(mh) - c)
        if with_abund:
            h = sig.minhash
            mh.add_many([int(x) for x in universe[9100:9200]])
            assert h.angular_similarity(mh) == mh.angular_similarity(h)
