"""The SourmashSignature object API on the GPU path, following the reference's tests/test_signature.py (cited per
case): copies and frozen-ness, equality, names, JSON round trips, containment helpers.  Run with -m gpu."""
import gzip

import pytest

from conftest import golden

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


def _at(sm, track_abundance, **kw):
    e = sm.MinHash(n=1, ksize=20, track_abundance=track_abundance, **kw)
    e.add_kmer("AT" * 10)
    return e


def test_copies_and_frozen(sm, track_abundance):
    # :20-57, 157-166
    e = _at(sm, track_abundance)
    assert e.copy() == e
    sig1 = sm.SourmashSignature(e, name="foo")
    sig2 = sig1.copy()
    assert sig1 == sig2 and isinstance(sig1.minhash, sm.FrozenMinHash)
    for sig in (sig2, sig1):
        with pytest.raises(TypeError) as err:
            sig.minhash.add_hash(5)
        assert "FrozenMinHash does not support modification" in str(err.value)
    sig1.minhash = sig1.minhash.to_mutable()                   # a mutable sketch handed in is stored frozen again
    assert sig1.copy() == sig1 and isinstance(sig1.to_frozen().minhash, sm.FrozenMinHash)


def test_equality_names_and_hash(sm, track_abundance):
    # :59-136, 258-287
    e, f = _at(sm, track_abundance), _at(sm, track_abundance)
    assert e == f
    assert sm.SourmashSignature(e, name="foo") != sm.SourmashSignature(f, name="bar")
    a, b = sm.SourmashSignature(e, name="foo", filename="a"), sm.SourmashSignature(f, name="foo", filename="b")
    assert a != b and b != a
    c, d = sm.SourmashSignature(e, name="foo"), sm.SourmashSignature(f, filename="b")
    assert c != d and d != c
    assert len({sm.SourmashSignature(e)}) == 1
    sig = sm.SourmashSignature(e)
    assert repr(sig) == "SourmashSignature('', 59502a74)"
    sig._name = "fizbar"
    assert repr(sig) == "SourmashSignature('fizbar', 59502a74)"
    empty = sm.MinHash(n=1, ksize=20, track_abundance=track_abundance)
    assert str(sm.SourmashSignature(empty, name="foo")) == "foo"
    assert str(sm.SourmashSignature(empty, filename="foo.txt")) == "foo.txt"
    assert str(sm.SourmashSignature(empty, name="foo", filename="foo.txt")) == "foo"
    anon = sm.SourmashSignature(empty)
    assert str(anon) == anon.md5sum()[:8]
    five = sm.MinHash(n=1, ksize=20, track_abundance=track_abundance)
    five.add_hash(5)
    assert sm.SourmashSignature(five).md5sum() == "eae27d77ca20db309e056e3d2dcd7d69"


def test_json_round_trips(sm, track_abundance, tmp_path):
    # :138-245, 289-400
    sig = sm.SourmashSignature(_at(sm, track_abundance))
    js = sm.save_signatures_to_json([sig])
    assert isinstance(js, bytes) and b"\n" not in js
    sig2 = list(sm.load_signatures_from_json(js))[0]
    assert sig.similarity(sig2) == 1.0 == sig2.similarity(sig)
    assert isinstance(sig2, sm.FrozenSourmashSignature) and not isinstance(sig, sm.FrozenSourmashSignature)
    assert isinstance(sig2.minhash, sm.FrozenMinHash)
    assert list(sm.load_signatures_from_json(js, ksize="20"))[0].similarity(sig) == 1.0       # ksize given as text
    empty = sm.SourmashSignature(sm.MinHash(n=1, ksize=20, track_abundance=track_abundance))
    assert list(sm.load_signatures_from_json(sm.save_signatures_to_json([empty])))[0].similarity(empty) == 0
    sc = sm.MinHash(n=0, ksize=20, track_abundance=track_abundance, max_hash=10)
    sc.add_hash(5)
    back = list(sm.load_signatures_from_json(sm.save_signatures_to_json([sm.SourmashSignature(sc)])))[0]
    assert back.minhash.scaled == sc.scaled and back.similarity(sm.SourmashSignature(sc)) == 1.0
    seeded = sm.MinHash(n=1, ksize=20, track_abundance=track_abundance, seed=10)
    seeded.add_hash(5)
    back = list(sm.load_signatures_from_json(sm.save_signatures_to_json([sm.SourmashSignature(seeded)])))[0]
    assert back.minhash.seed == 10
    # several signatures, minified, compressed, file handles
    s1 = sm.SourmashSignature(sm.MinHash(n=1, ksize=20, track_abundance=track_abundance), name="foo")
    s2 = sm.SourmashSignature(sm.MinHash(n=1, ksize=25, track_abundance=track_abundance), name="bar baz")
    both = sm.save_signatures_to_json([s1, s2])
    y = list(sm.load_signatures_from_json(both))
    assert len(y) == 2 and s1 in y and s2 in y and s1 != s2 and {s.name for s in y} == {"foo", "bar baz"}
    with pytest.raises(ValueError):
        sm.load_one_signature_from_json(sm.save_signatures_to_json([]))
    with pytest.raises(ValueError):
        sm.load_one_signature_from_json(both)
    assert sm.load_one_signature_from_json(sm.save_signatures_to_json([s1])) == s1
    packed = sm.save_signatures_to_json([s1], compression=5)
    assert packed[:2] == b"\x1f\x8b" and sm.load_one_signature_from_json(packed) == s1
    assert gzip.decompress(packed) == sm.save_signatures_to_json([s1])
    path = tmp_path / "1.sig"
    with open(path, "wb") as fp:
        assert sm.save_signatures_to_json([sig], fp) is None
    with open(path, "w") as fp:
        sm.save_signatures_to_json([sig], fp)                                              # text-mode handle works too
    assert sm.load_one_signature_from_json(str(path)) == sig
    assert sm.load_one_signature_from_json(path) == sig                                    # a path object
    with pytest.raises(Exception):
        list(sm.load_signatures_from_json(tmp_path / "dne.sig", do_raise=True))
    assert list(sm.load_signatures_from_json(tmp_path / "dne.sig")) == []
    multi = golden("ecoli", "GCF_000005845.2_ASM584v2_genomic.fna.gz.sig")
    minified = sm.save_signatures_to_json(sm.load_signatures_from_json(multi))
    assert b"\n" not in minified and len(list(sm.load_signatures_from_json(minified))) == 3


def test_similarity_downsample_and_bad_dna(sm, track_abundance):
    # :223-256
    e = sm.MinHash(n=0, ksize=20, track_abundance=track_abundance, max_hash=2**63)
    f = sm.MinHash(n=0, ksize=20, track_abundance=track_abundance, max_hash=2**2)
    for h in (1, 5):
        e.add_hash(h)
        f.add_hash(h)                                   # 5 exceeds f's max_hash
    assert len(e.hashes) == 2 and len(f.hashes) == 1
    ee, ff = sm.SourmashSignature(e), sm.SourmashSignature(f)
    with pytest.raises(ValueError) as err:
        ee.similarity(ff)
    assert "mismatch in scaled; comparison fail" in str(err.value)
    assert round(ee.similarity(ff, downsample=True), 1) == 1.0
    sig = sm.SourmashSignature(sm.MinHash(n=1, ksize=21))
    with pytest.raises(ValueError) as err:
        sig.add_sequence("N" * 21, force=False)
    assert "invalid DNA character in input k-mer: NNNNNNNNNNNNNNNNNNNNN" in str(err.value)


def test_containment_helpers_and_ani(sm):
    # :402-650 (field semantics; the ANI numbers themselves are pinned in test_distance_utils.py)
    def sig(*hashes):
        mh = sm.MinHash(0, 21, scaled=1)
        mh.add_many(hashes)
        return sm.SourmashSignature(mh)
    a, b, empty = sig(1, 2, 3, 4), sig(1, 5), sig()
    assert (a.contained_by(b), b.contained_by(a), a.max_containment(b), b.max_containment(a)) == (1 / 4, 1 / 2, 1 / 2, 1 / 2)
    assert a.contained_by(empty) == empty.contained_by(a) == a.max_containment(empty) == empty.max_containment(a) == 0
    same = sig(1, 2, 3, 4)
    assert a.contained_by(same) == same.contained_by(a) == a.max_containment(same) == 1
    assert a.avg_containment(b) == (1 / 4 + 1 / 2) / 2 == b.avg_containment(a)
    s47 = sm.load_one_signature_from_json(golden("pairs", "47.fa.sig"))
    s63 = sm.load_one_signature_from_json(golden("pairs", "63.fa.sig"))
    c = s47.containment_ani(s63, estimate_ci=True)
    assert c.ani == s47.minhash.containment_ani(s63.minhash).ani and c.ani_low < c.ani < c.ani_high
    assert s47.containment_ani(s63, containment=s47.contained_by(s63)).ani == c.ani
    assert s47.max_containment_ani(s63).ani == max(s47.containment_ani(s63).ani, s63.containment_ani(s47).ani)
    assert s47.avg_containment_ani(s63) == (s47.containment_ani(s63).ani + s63.containment_ani(s47).ani) / 2
    j = s47.jaccard_ani(s63)
    assert j.ani == s47.minhash.jaccard_ani(s63.minhash).ani and s47.jaccard_ani(s63, jaccard=s47.jaccard(s63)).ani == j.ani
    coarse = s47.minhash.downsample(scaled=2000)
    assert sm.SourmashSignature(coarse).containment_ani(s63, downsample=True).ani == \
        coarse.containment_ani(s63.minhash.downsample(scaled=2000)).ani


def test_frozen_signature_updates(sm, track_abundance):
    # :652-682
    e = _at(sm, track_abundance)
    ss = sm.SourmashSignature(e, name="foo").to_frozen()
    with pytest.raises(ValueError):
        ss.name = "foo2"
    with pytest.raises(ValueError):
        ss.minhash = e.copy_and_clear()
    with ss.update() as ss2:
        ss2.name = "foo2"
    assert ss2.name == "foo2" and isinstance(ss2, sm.FrozenSourmashSignature)
    assert ss.to_frozen() is ss and ss.to_mutable().name == "foo"
