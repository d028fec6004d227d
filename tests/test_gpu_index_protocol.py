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


def _lazy(three, tmp_path):
    from sourmash_amd.index import LazyLinearIndex
    return LazyLinearIndex(_linear(three, tmp_path))


def _zipfile(three, tmp_path):
    from sourmash_amd.index import ZipFileLinearIndex
    from sourmash_amd.save_load import SaveSignaturesToLocation
    loc = str(tmp_path / "index.zip")
    with SaveSignaturesToLocation(loc) as save:
        save.add_many(three)
    return ZipFileLinearIndex.load(loc)


def _zipfile_no_manifest(three, tmp_path):
    from sourmash_amd.index import ZipFileLinearIndex
    return ZipFileLinearIndex.load(_zipfile(three, tmp_path).location, use_manifest=False)


def _multi(three, tmp_path):
    from sourmash_amd.index import LinearIndex, MultiIndex
    return MultiIndex.load([LinearIndex(three, filename="three-sigs")], [None], None)


def _directory(three, tmp_path):
    from sourmash_amd.save_load import SaveSignaturesToLocation, load_file_as_index
    loc = str(tmp_path / "sigs") + "/"
    with SaveSignaturesToLocation(loc) as save:
        save.add_many(three)
    return load_file_as_index(loc)


def _standalone_manifest(three, tmp_path):
    from sourmash_amd.index import StandaloneManifestIndex
    from sourmash_amd.manifest import CollectionManifest
    names = ("2.fa.sig", "47.fa.sig", "63.fa.sig")
    mf = CollectionManifest.create_manifest(((ss, golden("pairs", n)) for ss, n in zip(three, names)), include_signature=False)
    mf.write_to_filename(str(tmp_path / "mf.csv"))
    return StandaloneManifestIndex.load(str(tmp_path / "mf.csv"))


# every Index class answers the same protocol (the reference parametrises its tests the same way, :166-196)
@pytest.fixture(params=[_linear, _lazy, _zipfile, _zipfile_no_manifest, _multi, _directory, _standalone_manifest],
                ids=lambda f: f.__name__.lstrip("_"))
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
    from sourmash_amd.manifest import BaseCollectionManifest
    assert len(index_obj) == 3 and bool(index_obj) and str(index_obj.location)
    assert index_obj.manifest is None or isinstance(index_obj.manifest, BaseCollectionManifest)
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


def test_zipfile_search_goes_through_the_bulk_loader(sm, tmp_path):
    """A zip collection is parsed natively into one CSR in HBM and scored in one pass; signatures are materialised
    for the matches only.  Same results, same order as the per-signature walk of the base class."""
    import glob
    from sourmash_amd.index import Index, ZipFileLinearIndex
    from sourmash_amd.save_load import SaveSignaturesToLocation
    from sourmash_amd.search import make_containment_query, make_jaccard_search_query
    loc = str(tmp_path / "genomes.zip")
    with SaveSignaturesToLocation(loc) as save:
        for path in sorted(glob.glob(golden("gather", "GCF_*.sig"))):
            save.add_many(sm.load_signatures_from_json(path))
    query = sm.load_one_signature_from_json(golden("gather", "combined.sig"))
    zidx = ZipFileLinearIndex.load(loc).select(ksize=query.minhash.ksize, moltype="DNA")
    assert len(zidx) == 12 and zidx._bulk_cache is None

    def rows(results):
        return [(r.score, r.signature.md5sum(), r.signature.name, r.location) for r in results]
    for make in (lambda: make_containment_query(query.minhash, 0), lambda: make_containment_query(query.minhash, 50000),
                 lambda: make_jaccard_search_query(threshold=0.05), lambda: make_jaccard_search_query(do_containment=True, threshold=0.1),
                 lambda: make_jaccard_search_query(do_max_containment=True, threshold=0.0, best_only=True)):
        fast = rows(zidx.find(make(), query))
        assert zidx._bulk_cache is not None                              # the native path ran
        assert fast == rows(Index.find(zidx, make(), query)) and (fast or make().threshold > 0)
    assert len(list(zidx.prefetch(query, threshold_bp=0))) == 12
    best = zidx.best_containment(query)
    assert best.signature.name.startswith("NC_003198.1") and round(best.score * len(query.minhash)) == 487
    # a picklist narrows the manifest; the CSR still holds everything, the walk skips what was not selected
    from sourmash_amd.picklist import SignaturePicklist
    pl = SignaturePicklist("identprefix")
    pl.init(["NC_003198", "NC_000853"])
    two = zidx.select(picklist=pl)
    assert sorted(r.signature.name.split(".")[0] for r in two.prefetch(query, threshold_bp=0)) == ["NC_000853", "NC_003198"]
    # protein sketches take the same path (manifest ksize is in residues, like MinHash.ksize)
    coarse = ZipFileLinearIndex.load(golden("zips", "all.zip")).select(moltype="protein", ksize=19)
    q = next(iter(coarse.signatures()))
    res = coarse.search(q, threshold=0.9)
    assert len(res) == 1 and res[0].score == 1.0 and res[0].signature == q and coarse._bulk_cache is not None


def test_zipfile_counter_gather_from_the_resident_collection(sm, tmp_path):
    """counter_gather on a zip collection: candidates come from the overlap pass over the CSR already in HBM and are
    gathered into their own CSR on the device; signatures are read from the archive only when a round returns them.
    Same rows as the object route, which is the reference's golden gather (tests/test_index_protocol.py:1057-1097)."""
    import glob
    from sourmash_amd.index import Index, ZipFileLinearIndex, _ArchiveCounterGather
    from sourmash_amd.save_load import SaveSignaturesToLocation
    from sourmash_amd.search import GatherDatabases
    loc = str(tmp_path / "genomes.zip")
    with SaveSignaturesToLocation(loc) as save:
        for path in sorted(glob.glob(golden("gather", "GCF_*.sig"))):
            save.add_many(sm.load_signatures_from_json(path))
    query = sm.load_one_signature_from_json(golden("gather", "combined.sig"))
    zidx = ZipFileLinearIndex.load(loc).select(ksize=query.minhash.ksize, moltype="DNA")

    def run(counter):
        return [(r.match.name.split()[0], r.unique_intersect_bp // query.minhash.scaled, r.f_match, r.remaining_bp)
                for r in GatherDatabases(query, [counter], threshold_bp=0)]
    fast_counter = zidx.counter_gather(query, 0)
    assert isinstance(fast_counter, _ArchiveCounterGather) and len(fast_counter.siglist) == 12
    slow_counter = Index.counter_gather(zidx, query, 0)
    assert fast_counter.counter == slow_counter.counter
    ov = sorted(slow_counter.counter.values(), reverse=True)
    assert len(ov) == 12
    assert fast_counter.union_found == slow_counter.union_found
    fast, slow = run(fast_counter), run(slow_counter)
    assert fast == slow
    assert [(n, c) for n, c, _, _ in fast] == [
        ("NC_003198.1", 487), ("NC_000853.1", 192), ("NC_011978.1", 169), ("NC_002163.1", 157), ("NC_003197.2", 152),
        ("NC_009486.1", 92), ("NC_006905.1", 76), ("NC_011080.1", 59), ("NC_011274.1", 42), ("NC_006511.1", 31),
        ("NC_011294.1", 7), ("NC_004631.1", 2)]
    # the whole loop in one native call, and a prefetch threshold that keeps 5 candidates
    assert [c for _, c in zidx.counter_gather(query, 0).gather_all()] == [c for _, c, _, _ in fast]
    cut_bp = ov[4] * query.minhash.scaled                              # keeps the rows overlapping at least as much as the 5th
    few = zidx.counter_gather(query, cut_bp)
    assert len(few.siglist) == len(Index.counter_gather(zidx, query, cut_bp).siglist) == sum(v >= ov[4] for v in ov) < 12
    with pytest.raises(ValueError):
        few.add(query)
    # nothing in common: an empty counter that answers the protocol
    mh = query.minhash.copy_and_clear()
    mh.add_many([1, 2, 3])
    empty = zidx.counter_gather(sm.SourmashSignature(mh), 0)
    assert len(empty.siglist) == 0 and empty.peek(mh) == [] and not empty.counter
