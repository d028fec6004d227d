"""Result rows of search / prefetch / gather: abundance-weighted gather
statistics against the numbers the reference's CLI tests print (tests/test_sourmash.py:6386-6600), and the
SearchResult / PrefetchResult / GatherResult field semantics of tests/test_search.py:257-760.  Run with -m gpu."""
import csv
import io

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sm():
    import torch  # noqa: F401
    import sourmash_amd
    assert sourmash_amd.gpu_available()
    return sourmash_amd


def _load(sm, *parts, ksize=None, moltype=None):
    return sm.load_one_signature_from_json(golden(*parts), ksize=ksize, select_moltype=moltype)


def _abund_gather(sm, query_name, **kw):
    from sourmash_amd.index import LinearIndex
    from sourmash_amd.search import GatherDatabases
    query = _load(sm, "gather-abund", query_name)
    db = LinearIndex([_load(sm, "gather-abund", f"genome-s{i}.fa.gz.sig") for i in (10, 11, 12)])
    counter = db.counter_gather(query, 0)
    return query, list(GatherDatabases(query, [counter], threshold_bp=0, **kw))


def _printed(r):
    "the three numeric columns `sourmash gather` prints (commands.py:1013-1024)"
    avg = None if r.average_abund is None else f"{r.average_abund:.1f}"
    return f"{r.f_unique_weighted * 100:.1f}%", f"{r.f_match * 100:.1f}%", avg


def test_gather_abund_1_to_1(sm):
    # tests/test_sourmash.py:6386-6429
    query, rows = _abund_gather(sm, "reads-s10-s11.sig")
    assert [r.match.filename.split("/")[-1] for r in rows] == ["genome-s10.fa.gz", "genome-s11.fa.gz"]
    assert [_printed(r) for r in rows] == [("49.6%", "78.5%", "1.8"), ("50.4%", "80.0%", "1.9")]
    assert round(rows[-1].sum_weighted_found / rows[-1].total_weighted_hashes, 3) == 1.0
    assert round(sum(r.f_unique_to_query for r in rows), 3) == 1.0


def test_gather_abund_10_to_1_and_csv_invariants(sm):
    # tests/test_sourmash.py:6432-6539
    query, rows = _abund_gather(sm, "reads-s10x10-s11.sig")
    assert [_printed(r) for r in rows] == [("91.0%", "100.0%", "14.5"), ("9.0%", "80.0%", "1.9")]
    buf = io.StringIO()
    w = None
    for r in rows:
        w = w or r.init_dictwriter(buf)
        r.write(w)
    table = list(csv.DictReader(io.StringIO(buf.getvalue())))
    assert list(table[0].keys()) == rows[0].gather_write_cols
    overlaps = [float(t["intersect_bp"]) for t in table]
    avg = [float(t["average_abund"]) for t in table]
    f_weighted = [float(t["f_unique_weighted"]) for t in table]
    prod = [o * a for o, a in zip(overlaps, avg)]
    for p, f in zip(prod, f_weighted):
        assert p / sum(prod) == f
    qmh = query.minhash
    assert sum(float(t["unique_intersect_bp"]) for t in table) + float(table[-1]["remaining_bp"]) == len(qmh) * qmh.scaled
    running = 0
    for n, t in enumerate(table):
        assert int(t["gather_result_rank"]) == n
        running += float(t["n_unique_weighted_found"])
        assert float(t["sum_weighted_found"]) == running
        assert float(t["total_weighted_hashes"]) == float(table[0]["total_weighted_hashes"]) == sum(qmh.hashes.values())
        assert t["query_abundance"] == "True" and len(t["query_md5"]) == 8 and len(t["md5"]) == 32
    assert float(table[-1]["sum_weighted_found"]) == float(table[-1]["total_weighted_hashes"])


def test_gather_abund_ignored(sm):
    # tests/test_sourmash.py:6542-6600
    query, rows = _abund_gather(sm, "reads-s10x10-s11.sig", ignore_abundance=True)
    assert [_printed(r)[:2] for r in rows] == [("57.2%", "100.0%"), ("42.8%", "80.0%")]
    for r in rows:
        d = r.gatherresultdict
        assert "average_abund" not in d and "median_abund" not in d and "std_abund" not in d     # written as ''
        assert d["query_abundance"] is False
        assert r.f_unique_weighted == r.f_unique_to_query


def test_search_and_prefetch_rows(sm):
    # tests/test_search.py:257-296, 377-432
    from sourmash_amd.search import PrefetchResult, SearchResult, SearchType
    ss47 = _load(sm, "pairs", "47.fa.sig", ksize=31)
    ss4763 = _load(sm, "pairs", "47+63.fa.sig", ksize=31).to_mutable()
    ss4763.filename = "somewhere/47+63.fa.sig"
    scaled = ss47.minhash.scaled
    res = SearchResult(ss47, ss4763, cmp_scaled=scaled, similarity=ss47.contained_by(ss4763))
    assert (res.query_name, res.match_name) == (ss47.name, ss4763.name)
    assert res.query_scaled == res.match_scaled == res.cmp_scaled == 1000
    assert (res.ksize, res.moltype, res.query_filename, res.match_filename) == (31, "DNA", "47.fa", ss4763.filename)
    assert (res.query_md5, res.match_md5, res.md5, res.name, res.filename) == \
        (ss47.md5sum(), ss4763.md5sum(), ss4763.md5sum(), ss4763.name, ss4763.filename)
    q_ani, m_ani = ss47.containment_ani(ss4763), ss4763.containment_ani(ss47)
    assert res.cmp.avg_containment_ani == np.mean([q_ani.ani, m_ani.ani])
    assert res.resultdict["query_md5"] == ss47.md5sum()[:8] and "ani" not in res.resultdict    # no searchtype: no ANI
    cres = SearchResult(ss47, ss4763, similarity=ss47.contained_by(ss4763), searchtype=SearchType.CONTAINMENT,
                        estimate_ani_ci=True)
    ci = ss47.containment_ani(ss4763, estimate_ci=True)
    assert (cres.ani, cres.ani_low, cres.ani_high) == (ci.ani, ci.ani_low, ci.ani_high)
    assert cres.write_cols == SearchResult.search_write_cols_ci
    jres = SearchResult(ss47, ss4763, similarity=ss47.jaccard(ss4763), searchtype=SearchType.JACCARD)
    assert jres.ani == ss47.jaccard_ani(ss4763).ani
    mres = SearchResult(ss47, ss4763, similarity=ss47.max_containment(ss4763), searchtype=SearchType.MAX_CONTAINMENT)
    assert mres.ani == ss47.max_containment_ani(ss4763).ani
    with pytest.raises(ValueError) as e:                                      # :364-374
        SearchResult(ss47, ss4763, cmp_scaled=scaled)
    assert "Must provide 'similarity' for SearchResult" in str(e.value)

    pf = PrefetchResult(ss47, ss4763, cmp_scaled=scaled)
    assert pf.intersect_bp == len(ss47.minhash.intersection(ss4763.minhash)) * scaled
    assert pf.jaccard == ss4763.jaccard(ss47) and pf.max_containment == ss4763.max_containment(ss47)
    assert pf.f_match_query == ss47.contained_by(ss4763) and pf.f_query_match == ss4763.contained_by(ss47)
    assert (pf.query_bp, pf.match_bp) == (len(ss47.minhash) * scaled, len(ss4763.minhash) * scaled)
    assert (pf.query_n_hashes, pf.match_n_hashes) == (len(ss47.minhash), len(ss4763.minhash))
    assert (pf.query_containment_ani, pf.match_containment_ani) == (q_ani.ani, m_ani.ani)
    assert pf.max_containment_ani == max(q_ani.ani, m_ani.ani) and pf.average_containment_ani == np.mean([q_ani.ani, m_ani.ani])
    assert pf.potential_false_negative is False
    d = pf.prefetchresultdict
    assert set(d) <= set(PrefetchResult.prefetch_write_cols) and d["scaled"] == 1000 and len(d["match_md5"]) == 8
    # num sketches cannot make prefetch rows (:435-447)
    num = _load(sm, "num", "genome-s10.fa.gz.sig", ksize=21, moltype="DNA")
    with pytest.raises(TypeError) as e:
        PrefetchResult(num, ss4763, cmp_scaled=scaled)
    assert "prefetch and gather results must be between scaled signatures" in str(e.value)


def test_gather_row_inputs_and_fields(sm):
    # tests/test_search.py:450-760
    from sourmash_amd.search import GatherResult, PrefetchResult
    ss47 = _load(sm, "pairs", "track_abund_47.fa.sig", ksize=31)
    ss4763 = _load(sm, "pairs", "47+63.fa.sig", ksize=31)
    scaled = ss47.minhash.scaled
    intersect_mh = ss47.minhash.flatten().intersection(ss4763.minhash)
    remaining = ss4763.minhash.to_mutable()
    remaining.remove_many(intersect_mh)
    abunds = ss47.minhash.hashes
    kw = dict(cmp_scaled=scaled, gather_querymh=remaining, gather_result_rank=1, total_weighted_hashes=1000,
              orig_query_len=len(ss47.minhash), orig_query_abunds=abunds)
    res = GatherResult(ss47, ss4763, **kw)
    assert res.query_abundance == ss47.minhash.track_abundance and res.match_abundance == ss4763.minhash.track_abundance
    assert res.query_bp == len(ss47.minhash) * scaled == ss47.minhash.unique_dataset_hashes
    assert res.match_bp == ss4763.minhash.unique_dataset_hashes
    assert res.query_filename == "podar-ref/47.fa" and res.intersect_bp == len(intersect_mh) * scaled
    assert res.max_containment == ss4763.max_containment(ss47)
    assert PrefetchResult(ss47, ss4763, cmp_scaled=scaled).prefetchresultdict == res.prefetchresultdict
    q_ani, m_ani = ss47.containment_ani(ss4763), ss4763.containment_ani(ss47)
    assert (res.query_containment_ani, res.match_containment_ani) == (q_ani.ani, m_ani.ani)
    assert res.gatherresultdict["intersect_bp"] == res.intersect_bp
    ci = GatherResult(ss47, ss4763, estimate_ani_ci=True, **kw)
    m_ci = ss4763.containment_ani(ss47, estimate_ci=True)
    assert (ci.match_containment_ani_low, ci.match_containment_ani_high) == (m_ci.ani_low, m_ci.ani_high)
    assert ci.gatherresultdict["match_containment_ani_low"] == m_ci.ani_low
    assert PrefetchResult(ss47, ss4763, cmp_scaled=scaled, estimate_ani_ci=True).prefetchresultdict == ci.prefetchresultdict
    for drop, text in (("cmp_scaled", "must provide comparison scaled value ('cmp_scaled')"),
                       ("gather_querymh", "must provide current gather sketch (remaining hashes)"),
                       ("gather_result_rank", "must provide 'gather_result_rank'"),
                       ("total_weighted_hashes", "must provide sum of all abundances ('total_weighted_hashes')"),
                       ("orig_query_abunds", "must provide original query abundances ('orig_query_abunds')")):
        bad = dict(kw)
        bad[drop] = None
        with pytest.raises(ValueError) as e:
            GatherResult(ss47, ss4763, **bad)
        assert text in str(e.value)
    with pytest.raises(ValueError):
        GatherResult(ss47, ss4763, **dict(kw, total_weighted_hashes=0))


def test_search_databases_drivers(sm):
    from sourmash_amd.index import LinearIndex
    from sourmash_amd.search import prefetch_database, search_databases_with_abund_query, search_databases_with_flat_query
    sigs = [_load(sm, "pairs", n, ksize=31) for n in ("47.fa.sig", "63.fa.sig", "47+63.fa.sig")]
    db1, db2 = LinearIndex(sigs[:2]), LinearIndex(sigs[1:])                 # 63 is in both: reported once
    rows = search_databases_with_flat_query(sigs[0], [db1, db2], threshold=0.0)
    assert [r.name for r in rows][0] == sigs[0].name and len(rows) == 3
    assert [r.similarity for r in rows] == sorted((r.similarity for r in rows), reverse=True)
    assert rows[1].similarity == sigs[0].jaccard(sigs[2]) and rows[1].ani == sigs[0].minhash.jaccard_ani(sigs[2].minhash).ani
    crow = search_databases_with_flat_query(sigs[0], [db2], threshold=0.0, do_containment=True, estimate_ani_ci=True)
    assert crow[0].similarity == sigs[0].contained_by(sigs[2]) == 1.0 and crow[0].write_cols[-2:] == ["ani_low", "ani_high"]
    pre = list(prefetch_database(sigs[0], db2, threshold_bp=1_000_000))
    assert [p.match_name for p in pre] == [sigs[1].name, sigs[2].name] or [p.match_name for p in pre] == [sigs[2].name, sigs[1].name]
    assert all(p.intersect_bp >= 1_000_000 for p in pre)
    a47, a63 = _load(sm, "pairs", "track_abund_47.fa.sig"), _load(sm, "pairs", "track_abund_63.fa.sig")
    arows = search_databases_with_abund_query(a47, [LinearIndex([a47, a63])], threshold=0.0)
    assert arows[0].similarity == 1.0 and arows[1].similarity == a47.minhash.angular_similarity(a63.minhash)
    with pytest.raises(TypeError):
        search_databases_with_abund_query(a47, [LinearIndex([a63])], do_containment=True)
