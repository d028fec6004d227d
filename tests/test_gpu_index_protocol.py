"""The Index protocol on LinearIndex over the GPU kernels: the three-signature cases of the reference's
tests/test_index_protocol.py:199-510 (search thresholds, containment, select, prefetch, best_containment with
thresholds).  Run with -m gpu."""
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sm():
    import torch  # noqa: F401
    import sourmash_amd
    assert sourmash_amd.gpu_available()
    return sourmash_amd


@pytest.fixture()
def three(sm):
    return [sm.load_one_signature_from_json(golden("pairs", "2.fa.sig"), ksize=31),
            sm.load_one_signature_from_json(golden("pairs", "47.fa.sig")),
            sm.load_one_signature_from_json(golden("pairs", "63.fa.sig"))]


def _linear(three, tmp_path):
    from sourmash_amd.index import LinearIndex
    lidx = LinearIndex(filename="three-sigs")
    for ss in three:
        lidx.insert(ss)
    return lidx


# the Index classes in scope: LinearIndex, as a list and loaded from a JSON file.  (Zip / Multi /
# manifest-backed indices are the reference's control plane; collections on disk go through SketchSet.load, which has its
# own suite in tests/test_gpu_collection.py.)
def _linear_from_file(three, tmp_path):
    from sourmash_amd.index import LinearIndex
    path = str(tmp_path / "three.sig")
    _linear(three, tmp_path).save(path)
    return LinearIndex.load(path)


@pytest.fixture(params=[_linear, _linear_from_file], ids=lambda f: f.__name__.lstrip("_"))
def index_obj(request, three, tmp_path):
    return request.param(three, tmp_path)


def test_search_thresholds(index_obj, three):
    # :203-267
    ss2, ss47, ss63 = three
    sr = index_obj.search(ss2, threshold=1.0)
    assert len(sr) == 1 and sr[0].signature.minhash == ss2.minhash and sr[0].score == 1.0
    for q, other in ((ss47, ss63), (ss63, ss47)):
        sr = sorted(index_obj.search(q, threshold=0.1), key=lambda x: -x[0])
        assert len(sr) == 2 and sr[0].signature.minhash == q.minhash and sr[0].score == 1.0
        assert sr[1].signature.minhash == other.minhash and round(sr[1].score, 2) == 0.32
    sr = index_obj.search(ss63, threshold=0.8)
    assert len(sr) == 1 and sr[0].signature.minhash == ss63.minhash and sr[0].score == 1.0
    sr = sorted(index_obj.search(ss63, do_containment=True, threshold=0.1), key=lambda x: -x[0])
    assert len(sr) == 2 and sr[0].signature.minhash == ss63.minhash and sr[0].score == 1.0
    assert sr[1].signature.minhash == ss47.minhash and round(sr[1].score, 2) == 0.48
    with pytest.raises(TypeError):
        index_obj.search(ss63)                                          # a threshold is mandatory


def test_container_protocol_and_select(sm, index_obj, three):
    # :269-360
    from sourmash_amd.index import LinearIndex
    md5s = {ss.md5sum() for ss in three}
    assert {ss.md5sum() for ss in index_obj.signatures()} == md5s
    assert {ss.md5sum() for ss, loc in index_obj.signatures_with_location()} == md5s
    assert len(index_obj) == 3 and bool(index_obj) and str(index_obj.location)
    assert index_obj.manifest is None
    idx = index_obj.select(ksize=31, moltype="DNA", abund=False, containment=True, scaled=1000, num=0, picklist=None)
    assert len(idx) == 3 and {ss.md5sum() for ss in idx.signatures()} == md5s
    for bad in ({"ksize": "31"}, {"ksize": 31.1}, {"moltype": "dna"}, {"moltype": "foo"}, {"scaled": 1000.1}, {"num": 1000.1},
                {"abund": 1}, {"plausible_extra_parameter": 5}):
        with pytest.raises(ValueError):
            index_obj.select(**bad)
    nada = index_obj.select(ksize=21)
    assert len(nada) == 0 and list(nada.signatures()) == [] and not nada
    assert len(index_obj.select(num=500)) == 0 and len(index_obj.select(abund=True)) == 0
    if isinstance(index_obj, LinearIndex):
        with pytest.raises(ValueError):
            index_obj.select(containment=True)                          # per-signature selection: containment needs a scaled value
    assert len(LinearIndex([])) == 0


def test_prefetch_and_best_containment(sm, index_obj, three):
    # :362-510
    ss2, ss47, ss63 = three
    res = list(index_obj.prefetch(ss2, threshold_bp=0))
    assert len(res) == 1 and res[0].signature.minhash == ss2.minhash
    res = sorted(index_obj.prefetch(ss47, threshold_bp=0), key=lambda r: -r.score)     # walk order is the container's
    assert len(res) == 2 and res[0].signature.minhash == ss47.minhash and res[1].signature.minhash == ss63.minhash
    for q in (ss2, ss47):
        match = index_obj.best_containment(q)
        assert match and match.score == 1.0 and match.signature.minhash == q.minhash
    mins = sorted(ss2.minhash.hashes)
    new_mh = ss2.minhash.copy_and_clear()
    with pytest.raises(ValueError):
        index_obj.best_containment(sm.SourmashSignature(new_mh))       # empty query
    new_mh.add_hash(mins.pop())
    containment, match_sig, name = index_obj.best_containment(sm.SourmashSignature(new_mh))
    assert containment == 1.0 and match_sig.minhash == ss2.minhash
    with pytest.raises(ValueError):
        index_obj.best_containment(sm.SourmashSignature(new_mh), threshold_bp=5000)    # 1 hash = 1000 bp < 5000
    for _ in range(3):
        new_mh.add_hash(mins.pop())
    assert len(new_mh) == 4
    assert index_obj.best_containment(sm.SourmashSignature(new_mh)).score == 1.0
    with pytest.raises(ValueError):
        index_obj.best_containment(sm.SourmashSignature(new_mh), threshold_bp=5000)    # 4000 bp still short
    for _ in range(21):
        new_mh.add_hash(mins.pop())
    assert len(new_mh) == 25
    containment, match_sig, name = index_obj.best_containment(sm.SourmashSignature(new_mh), threshold_bp=5000)
    assert containment == 1.0 and match_sig.minhash == ss2.minhash


def test_gather_over_the_index(sm, index_obj, three):
    "counter_gather + GatherDatabases over the same three signatures: 47+63 is covered by 47 and the rest of 63"
    from sourmash_amd.search import GatherDatabases
    ss2, ss47, ss63 = three
    query = sm.load_one_signature_from_json(golden("pairs", "47+63.fa.sig"), ksize=31)
    rows = list(GatherDatabases(query, [index_obj.counter_gather(query, 0)], threshold_bp=0))
    assert [r.match.md5sum() for r in rows] == [ss63.md5sum(), ss47.md5sum()]
    assert rows[0].f_match == 1.0 and rows[0].unique_intersect_bp == len(ss63.minhash) * 1000
    assert rows[1].unique_intersect_bp == (len(ss47.minhash) - ss47.minhash.count_common(ss63.minhash)) * 1000
    assert rows[1].remaining_bp == 0 and round(sum(r.f_unique_to_query for r in rows), 6) == 1.0
