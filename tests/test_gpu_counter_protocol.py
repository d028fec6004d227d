"""CounterGather's add / peek / consume protocol on the GPU counters, case by case as the reference's
tests/test_index_protocol.py:700-1312 lays it out (contrived overlaps, thresholds, mixed scaled, abundance,
identical matches, misuse).  Run with -m gpu."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sm():
    import torch  # noqa: F401
    import sourmash_amd
    assert sourmash_amd.gpu_available()
    return sourmash_amd


def _query(sm, hashes=range(0, 20), **kw):
    mh = sm.MinHash(n=0, ksize=31, scaled=1, **kw)
    mh.add_many(hashes)
    return mh


def _consume_all(query_mh, counter, threshold_bp=0):
    "the reference's driver (:743-764): peek, consume, shrink the query by the intersection"
    results, last = [], None
    query_mh = query_mh.to_mutable()
    while True:
        result = counter.peek(query_mh, threshold_bp=threshold_bp)
        if not result:
            break
        sr, intersect_mh = result
        assert last is None or len(intersect_mh) <= last
        last = len(intersect_mh)
        counter.consume(intersect_mh)
        query_mh.remove_many(intersect_mh.hashes)
        results.append((sr, len(intersect_mh)))
    return results


CASES = {
    # name: (match ranges, transform of the match sketch, transform of the query sketch, threshold_bp, expected)
    "1_disjoint": ([(0, 10), (10, 15), (15, 17)], None, None, 0, [("match1", 10), ("match2", 5), ("match3", 2)]),
    "1b_overlapping": ([(0, 10), (7, 15), (13, 17)], None, None, 0, [("match1", 10), ("match2", 5), ("match3", 2)]),
    "1c_threshold": ([(0, 10), (7, 15), (13, 17)], None, None, 3, [("match1", 10), ("match2", 5)]),
    "1d_match_scaled": ([(0, 10), (7, 15), (13, 17)], "scaled", None, 0, [("match1", 10), ("match2", 5), ("match3", 2)]),
    "1d_query_scaled": ([(0, 10), (7, 15), (13, 17)], "scaled", "scaled100", 0, [("match1", 10), ("match2", 5), ("match3", 2)]),
    "1e_abund_query": ([(0, 10), (7, 15), (13, 17)], None, "abund", 0, [("match1", 10), ("match2", 5), ("match3", 2)]),
    "1f_abund_match": ([(0, 10), (7, 15), (13, 17)], "abund", None, 0, [("match1", 10), ("match2", 5), ("match3", 2)]),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_contrived_gathers(sm, case):
    # :766-1055
    from sourmash_amd.index import CounterGather
    ranges, match_kind, query_kind, threshold_bp, expected = CASES[case]
    query_mh = _query(sm, track_abundance=query_kind == "abund")
    matches = []
    for i, (lo, hi) in enumerate(ranges):
        if match_kind == "scaled":
            mh = query_mh.copy_and_clear().downsample(scaled=10 * (i + 1)).to_mutable()
        elif match_kind == "abund":
            mh = sm.MinHash(n=0, ksize=31, scaled=1, track_abundance=True)
        else:
            mh = query_mh.copy_and_clear().flatten().to_mutable() if query_kind == "abund" else query_mh.copy_and_clear()
        mh.add_many(range(lo, hi))
        matches.append(sm.SourmashSignature(mh, name=f"match{i + 1}"))
    if query_kind == "scaled100":
        query_mh = query_mh.downsample(scaled=100)
    query_ss = sm.SourmashSignature(query_mh.flatten() if match_kind == "abund" else query_mh, name="query")
    counter = CounterGather(query_ss)
    for ss in matches:
        counter.add(ss)
    siglist = list(counter.signatures())                                    # :709-741
    assert len(siglist) == 3 and all(ss in siglist for ss in matches)
    results = _consume_all(query_ss.minhash.flatten(), counter, threshold_bp=threshold_bp)
    assert [(sr.signature.name, n) for sr, n in results] == expected


def test_exact_and_identical_matches(sm):
    # :1098-1142
    from sourmash_amd.index import CounterGather
    query_ss = sm.SourmashSignature(_query(sm), name="query")
    counter = CounterGather(query_ss)
    counter.add(query_ss, location="somewhere over the rainbow")
    (sr, n), = _consume_all(query_ss.minhash, counter)
    assert sr.score == 1.0 and sr.signature == query_ss and sr.location == "somewhere over the rainbow" and n == 20
    counter = CounterGather(query_ss)
    match_mh = _query(sm, range(5, 15))
    for name in ("match1", "match2", "match3"):                             # same sketch under three names: one result
        counter.add(sm.SourmashSignature(match_mh, name=name), location=name)
    (sr, n), = _consume_all(query_ss.minhash, counter)
    assert sr.score == 0.5 and n == 10 and sr.location in ("match1", "match2", "match3")


def test_misuse_and_empty_cases(sm):
    # :1144-1312
    from sourmash_amd.index import CounterGather
    query_ss = sm.SourmashSignature(_query(sm), name="query")
    for first in ("peek", "consume"):                                       # no adds once the protocol has started
        counter = CounterGather(query_ss)
        counter.add(query_ss, location="x")
        getattr(counter, first)(query_ss.minhash)
        with pytest.raises(ValueError):
            counter.add(query_ss, location="try again")
    counter = CounterGather(query_ss)
    counter.add(query_ss)
    counter.consume(query_ss.minhash.copy_and_clear())                      # an empty intersect is a no-op
    assert _consume_all(query_ss.minhash.copy_and_clear(), counter) == []  # ... and so is an empty current query
    outside = query_ss.minhash.copy_and_clear()
    outside.add_many(range(20, 30))
    with pytest.raises(ValueError):
        counter.peek(outside)                                               # not a subset of the original query
    empty_q = sm.SourmashSignature(sm.MinHash(n=0, ksize=31, scaled=1), name="query")
    counter = CounterGather(empty_q)
    assert counter.peek(empty_q.minhash) == []                              # empty counter
    counter = CounterGather(empty_q)
    counter.add(sm.SourmashSignature(_query(sm, range(0, 10)), name="m"), require_overlap=False)
    assert counter.peek(empty_q.minhash) == []                              # empty initial query
    num_q = sm.MinHash(n=500, ksize=31)
    num_q.add_many(range(0, 10))
    with pytest.raises(ValueError):
        CounterGather(sm.SourmashSignature(num_q, name="query"))            # gather needs scaled
    counter = CounterGather(query_ss)
    num_m = sm.MinHash(n=500, ksize=31)
    num_m.add_many(range(0, 20))
    with pytest.raises((ValueError, TypeError)):
        counter.add(sm.SourmashSignature(num_m, name="nm"))
    ten = sm.SourmashSignature(_query(sm, range(0, 10)), name="query")
    counter = CounterGather(ten)
    with pytest.raises(ValueError):
        counter.add(sm.SourmashSignature(_query(sm, range(10, 20)), name="match1"))   # no overlap
    assert counter.peek(ten.minhash) == []
    counter = CounterGather(query_ss)
    counter.add(sm.SourmashSignature(_query(sm, range(0, 10)), name="match1"))
    assert counter.peek(query_ss.minhash, threshold_bp=30 * query_ss.minhash.scaled) == []   # unattainable threshold
