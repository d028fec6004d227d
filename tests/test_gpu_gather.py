"""GPU parity: search / prefetch / gather (CounterGather on device counters) against the
oracle and the reference's golden gather results.  Run with -m gpu."""
import glob

import numpy as np
import pytest

import oracle
from conftest import golden

pytestmark = pytest.mark.gpu

GOLDEN_GATHER = [("NC_003198.1", 487), ("NC_000853.1", 192), ("NC_011978.1", 169), ("NC_002163.1", 157),
                 ("NC_003197.2", 152), ("NC_009486.1", 92), ("NC_006905.1", 76), ("NC_011080.1", 59),
                 ("NC_011274.1", 42), ("NC_006511.1", 31), ("NC_011294.1", 7), ("NC_004631.1", 2)]


@pytest.fixture(scope="module")
def sm():
    import torch  # noqa: F401
    import sourmash_amd
    assert sourmash_amd.gpu_available()
    return sourmash_amd


def _sig(sm, hashes, name, scaled=1, ksize=21):
    mh = sm.MinHash(0, ksize, scaled=scaled)
    mh.add_many(hashes)
    return sm.SourmashSignature(mh, name=name)


def _gather_db(sm):
    files = sorted(glob.glob(golden("gather", "GCF_*.sig")))
    return [sm.load_one_signature_from_json(f, ksize=21) for f in files]


def test_golden_gather_counter_protocol(sm):
    # tests/test_index_protocol.py:1057-1097: CounterGather over the 12 genomes, (name, |intersect|) per round
    from sourmash_amd.index import CounterGather
    query = sm.load_one_signature_from_json(golden("gather", "combined.sig"), ksize=21)
    counter = CounterGather(query)
    for ss in _gather_db(sm):
        counter.add(ss)
    got = []
    query_mh = query.minhash.to_mutable()
    while True:
        res = counter.peek(query_mh)
        if not res:
            break
        sr, isect = res
        got.append((sr.signature.name.split()[0], len(isect)))
        counter.consume(isect)
        query_mh.remove_many(sr.signature.minhash)
    assert got == GOLDEN_GATHER


def test_golden_gather_databases_and_stats(sm):
    from sourmash_amd.index import LinearIndex
    from sourmash_amd.search import GatherDatabases
    query = sm.load_one_signature_from_json(golden("gather", "combined.sig"), ksize=21)
    db = LinearIndex(_gather_db(sm))
    counter = db.counter_gather(query, 0)
    results = list(GatherDatabases(query, [counter], threshold_bp=0))
    assert [(r.name.split()[0], r.n_intersect) for r in results] == GOLDEN_GATHER
    assert [r.gather_result_rank for r in results] == list(range(12))
    # tests/test_sourmash.py:4546-4614 (three Thermotoga genomes)
    names = ("GCF_000016785.1_ASM1678v1", "GCF_000018945.1_ASM1894v1", "GCF_000008545.1_ASM854v1")
    three = LinearIndex([sm.load_one_signature_from_json(golden("gather", n + "_genomic.fna.gz.sig"), ksize=21)
                         for n in names])
    rows = list(GatherDatabases(query, [three.counter_gather(query, 0)], threshold_bp=0))
    assert [r.name.split()[0] for r in rows] == ["NC_000853.1", "NC_011978.1", "NC_009486.1"]
    assert rows[0].f_match == 1.0 and round(rows[0].f_unique_to_query, 5) == round(0.13096862, 5)
    assert (rows[0].unique_intersect_bp, rows[0].remaining_bp) == (1920000, 12740000)
    assert round(rows[1].f_match, 5) == round(0.898936170212766, 5) and round(rows[1].f_unique_to_query, 5) == round(0.115279, 5)
    assert (rows[1].unique_intersect_bp, rows[1].remaining_bp) == (1690000, 11050000)
    assert round(rows[2].f_match, 5) == round(0.4842105, 5) and round(rows[2].f_unique_to_query, 5) == round(0.0627557, 5)
    assert (rows[2].unique_intersect_bp, rows[2].remaining_bp) == (920000, 10130000)


def test_counter_trace_ties_and_threshold(sm):
    # tests/test_index.py:1581-1679: query 0..19; matches 0-9 / 7-14 / 13-16 -> counters 10,8,4 -> 5,4 -> 2
    from sourmash_amd.index import CounterGather
    query = _sig(sm, range(0, 20), "q")
    m = [_sig(sm, range(0, 10), "a"), _sig(sm, range(7, 15), "b"), _sig(sm, range(13, 17), "c")]
    cg = CounterGather(query)
    for ss in m:
        cg.add(ss)
    assert sorted(cg.counter.values(), reverse=True) == [10, 8, 4]
    qmh = query.minhash.to_mutable()
    sr, isect = cg.peek(qmh)
    assert sr.signature.name == "a" and len(isect) == 10 and sr.score == 0.5
    cg.consume(isect)
    assert sorted(cg.counter.values(), reverse=True) == [5, 4]
    qmh.remove_many(sr.signature.minhash)
    sr, isect = cg.peek(qmh)
    assert sr.signature.name == "b" and len(isect) == 5
    cg.consume(isect)
    qmh.remove_many(sr.signature.minhash)
    assert list(cg.counter.values()) == [2]
    sr, isect = cg.peek(qmh)
    assert sr.signature.name == "c" and len(isect) == 2
    cg.consume(isect)
    assert cg.counter == {}
    with pytest.raises(ValueError):
        cg.add(m[0])                                   # no adds after peek/consume (:778-779)
    # ties go to the first inserted (Counter.most_common is stable)
    cg = CounterGather(query)
    cg.add_many([_sig(sm, range(10, 15), "x"), _sig(sm, range(0, 5), "y"), _sig(sm, range(0, 3), "z")])
    assert cg.gather_all() == [(_sig(sm, range(10, 15), "x").md5sum(), 5), (_sig(sm, range(0, 5), "y").md5sum(), 5)]
    # threshold_bp / scaled = minimum hashes (search.py:15-37); unattainable -> no result
    for thr, want in ((5, 2), (6, 1), (11, 0), (21, 0)):
        cg = CounterGather(query)
        cg.add_many(m)
        assert len(cg.gather_all(threshold_bp=thr)) == want, thr
    with pytest.raises(ValueError):
        CounterGather(query).add(_sig(sm, range(100, 110), "nope"))   # require_overlap
    with pytest.raises(ValueError):
        cg2 = CounterGather(query)
        cg2.add(m[0])
        cg2.peek(_sig(sm, range(15, 30), "not-subset").minhash)


def test_search_and_prefetch_vs_oracle(sm):
    from sourmash_amd.index import LinearIndex
    from sourmash_amd.synth import synth_sketches
    sk = synth_sketches(120, pool_size=6000)
    sigs = [_sig(sm, s, f"s{i}", scaled=1000, ksize=31) for i, s in enumerate(sk)]
    db = LinearIndex(sigs)
    q = sigs[5]
    hashes, offsets = oracle.make_csr(sk)
    wc, wj = oracle.compare_all_pairs(hashes, offsets, nthreads=4)
    res = db.search(q, threshold=0.06)
    want = sorted([(wj[5, j], j) for j in range(len(sk)) if wj[5, j] >= 0.06 and wj[5, j] > 0], key=lambda t: -t[0])
    assert [r.score for r in res] == [w[0] for w in want]
    assert {r.signature.name for r in res} == {f"s{j}" for _, j in want}
    cres = db.search(q, threshold=0.1, do_containment=True)
    wantc = {j for j in range(len(sk)) if len(sk[5]) and wc[5, j] / len(sk[5]) >= 0.1}
    assert {r.signature.name for r in cres} == {f"s{j}" for j in wantc}
    pre = list(db.prefetch(q, threshold_bp=60 * 1000))
    assert {r.signature.name for r in pre} == {f"s{j}" for j in range(len(sk)) if wc[5, j] >= 60}
    with pytest.raises(ValueError):
        list(LinearIndex([]).prefetch(q, 0))
    assert db.best_containment(q).signature.name in ("s5", "s116")          # itself or its planted duplicate


@pytest.mark.parametrize("build", ["atomic", "ranges", "ranges-direct"])
def test_synthetic_gather_vs_oracle(sm, build, monkeypatch):
    # the builders of the inverted index (csrc/gather_build.hip: one atomic per element / range-partitioned with the
    # histogram in LDS, postings filled through the two-level partition or by direct stores) -- the size heuristic
    # would pick "atomic" for a database this small
    monkeypatch.setenv("SMG_GATHER_BUILD", build.split("-")[0])
    monkeypatch.setenv("SMG_GATHER_FILL", build.split("-")[1] if "-" in build else "staged")
    from sourmash_amd.index import CounterGather
    from sourmash_amd.synth import synth_gather
    qh, dbh = synth_gather(n_query=60_000, n_db=1500, db_size=600)
    query = _sig(sm, qh, "q", scaled=1000, ksize=31)
    sigs = [_sig(sm, h, f"d{i}", scaled=1000, ksize=31) for i, h in enumerate(dbh)]
    hashes, offsets = oracle.make_csr(dbh)
    for thr_bp in (0, 50_000, 200_000):
        want = oracle.gather(qh, hashes, offsets, threshold_bp=thr_bp, scaled=1000)
        cg = CounterGather(query)
        cg.add_many(sigs)
        got = cg.gather_all(threshold_bp=thr_bp)
        md5_to_idx = {s.md5sum(): i for i, s in enumerate(sigs)}
        assert [(md5_to_idx[m], n) for m, n in got] == want, thr_bp
        assert len(want) > 10 or thr_bp == 200_000


@pytest.mark.parametrize("fill,pass1", [("staged", "ranges"), ("staged", "stage"), ("direct", "ranges")])
def test_range_builder_postings_are_exact(fill, pass1, monkeypatch):
    """The postings themselves (not only the gather they drive): after the range-partitioned build every counter equals
    |Q ∩ row|, and consuming the whole query through the postings brings every counter to exactly zero -- which holds
    iff every (query hash, row) pair sits in exactly one posting.  Ragged rows, an empty row, rows outside the query,
    a query whose last range is a few lists long."""
    import torch
    from sourmash_amd import device as smd, parallel
    from sourmash_amd.synth import synth_gather
    monkeypatch.setenv("SMG_GATHER_BUILD", "ranges")
    monkeypatch.setenv("SMG_GATHER_FILL", fill)
    monkeypatch.setenv("SMG_GATHER_PASS1", pass1)                 # pass 1 by lookups in L2 / by the lean streaming kernel staging the
                                                                  # postings itself; a fallback is an error
    qh, dbh = synth_gather(n_query=3 * 32768 + 77, n_db=700, db_size=900)
    dbh[3] = np.zeros(0, dtype=np.uint64)
    dbh[4] = np.array([1, 2, 3], dtype=np.uint64)
    dbh[5] = qh[::7].copy()                                       # a long row made of query hashes only
    dbh[6] = qh[-40:].copy()                                      # only the last (short) range
    dbh[7] = np.concatenate([qh[5:50], np.array([2**64 - 1], dtype=np.uint64)])   # the hash the walk's filler value looks like
    dbh[8] = qh[2000:2300].copy()                                 # 300 consecutive query hashes: a row's part of one range > a visit
    be = parallel.DeviceBackend()
    h, off = smd.pack_csr(dbh)
    q = torch.from_numpy(qh.view(np.int64).copy()).cuda()
    st = be.gather_state(q, len(qh), h, off, len(dbh), 0)
    want = np.array([oracle.intersection_size(qh, d)[0] for d in dbh], dtype=np.uint64)
    assert np.array_equal(st.counters(), want)
    assert int(be.lib.smgpu_gather_postings(st._ptr)) == int(want.sum())
    # the greedy loop = the oracle's, and when it has run to the end every counter is back to |row ∩ uncovered| = 0
    st.begin(0, len(dbh))
    got = st.run()
    assert got == oracle.gather(qh, *oracle.make_csr(dbh), threshold_bp=0, scaled=1000, nthreads=8)
    assert not st.counters().any()


def test_index_build_of_many_rows_takes_the_lean_pass(monkeypatch):
    """From 64 rows per CU up the builder's pass 1 is the lean streaming kernel by default (gather_build.hip: gather_build_body): 17,000
    ragged rows, the result the oracle's, every counter zero at the end (every posting in exactly one list), and the same
    again with pass 1 by lookups."""
    import torch
    from sourmash_amd import device as smd, parallel
    from sourmash_amd.synth import synth_gather
    qh, dbh = synth_gather(n_query=150_000, n_db=17_000, db_size=70)
    dbh[0] = np.zeros(0, dtype=np.uint64)
    dbh[1] = qh[::5].copy()                                       # 30,000 hashes, all in the query
    dbh[16_999] = np.concatenate([qh[-3:], np.array([2**64 - 1], dtype=np.uint64)])
    be = parallel.DeviceBackend()
    h, off = smd.pack_csr(dbh)
    q = torch.from_numpy(qh.view(np.int64).copy()).cuda()
    want = np.array([oracle.intersection_size(qh, d)[0] for d in dbh], dtype=np.uint64)
    ref = oracle.gather(qh, *oracle.make_csr(dbh), threshold_bp=0, scaled=1000, nthreads=8)
    for pass1 in (None, "stage", "ranges"):                        # the default must be the staging form: it is forced as well, and the lookups
        if pass1:
            monkeypatch.setenv("SMG_GATHER_PASS1", pass1)
        st = be.gather_state(q, len(qh), h, off, len(dbh), 0)
        assert np.array_equal(st.counters(), want), pass1
        assert int(be.lib.smgpu_gather_postings(st._ptr)) == int(want.sum())
        st.begin(0, len(dbh))
        assert st.run() == ref, pass1
        assert not st.counters().any()
    # eighty neighbouring rows that all hold the same 256 consecutive query hashes: more postings in one window than the staging
    # form has room for in LDS -- it must notice and the builder fall back (forced, it refuses)
    monkeypatch.delenv("SMG_GATHER_PASS1")
    for d in range(4000, 4080):
        dbh[d] = np.unique(np.concatenate([dbh[d], qh[70_000:70_256]]))
    h, off = smd.pack_csr(dbh)
    want = np.array([oracle.intersection_size(qh, d)[0] for d in dbh], dtype=np.uint64)
    st = be.gather_state(q, len(qh), h, off, len(dbh), 0)
    assert np.array_equal(st.counters(), want)
    assert int(be.lib.smgpu_gather_postings(st._ptr)) == int(want.sum())
    st.begin(0, len(dbh))
    assert st.run() == oracle.gather(qh, *oracle.make_csr(dbh), threshold_bp=0, scaled=1000, nthreads=8)
    assert not st.counters().any()
    monkeypatch.setenv("SMG_GATHER_PASS1", "stage")
    with pytest.raises(Exception):
        be.gather_state(q, len(qh), h, off, len(dbh), 0)


@pytest.mark.parametrize("n_query", [700, 8193, 8450, 16_384 + 255])
def test_lean_index_build_with_short_and_awkward_queries(n_query, monkeypatch):
    """The staging form of pass 1 cuts the query into ranges of W positions (a multiple of 256, at most 8,192): a query of one
    partial window, one of a full range plus a single position, one whose last window is a few lists, one that ends one list
    short of a window -- 17,000 short rows each; forced, so a fallback would be an error."""
    import torch
    from sourmash_amd import device as smd, parallel
    from sourmash_amd.synth import synth_gather
    monkeypatch.setenv("SMG_GATHER_PASS1", "stage")
    qh, dbh = synth_gather(n_query=n_query, n_db=17_000, db_size=24)
    dbh[5] = qh.copy()                                            # the whole query as a row
    dbh[6] = qh[-1:].copy()
    dbh[7] = qh[:1].copy()
    be = parallel.DeviceBackend()
    h, off = smd.pack_csr(dbh)
    q = torch.from_numpy(qh.view(np.int64).copy()).cuda()
    want = np.array([oracle.intersection_size(qh, d)[0] for d in dbh], dtype=np.uint64)
    st = be.gather_state(q, len(qh), h, off, len(dbh), 0)
    assert np.array_equal(st.counters(), want)
    assert int(be.lib.smgpu_gather_postings(st._ptr)) == int(want.sum())
    st.begin(0, len(dbh))
    assert st.run() == oracle.gather(qh, *oracle.make_csr(dbh), threshold_bp=0, scaled=1000, nthreads=8)
    assert not st.counters().any()


@pytest.mark.parametrize("form", ["wide", "stream", "rows"])
def test_overlaps_of_a_large_query_take_the_range_partitioned_pass(form):
    """smgpu_overlap_raw with a large query over >= 4096 rows (overlap.hip: overlap_ranges_launch), both ops, vs the oracle: the
    lean walk (SMG_OVERLAP=wide: a fallback is an error), the 16-lane streaming form (stream: likewise) and the one-wave-per-row
    kernel that takes whatever fits neither (rows).  The switch is read once per process, so each form runs in its own interpreter."""
    import os, subprocess, sys
    from conftest import ROOT
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_gather as t\nt._overlaps_large_query()\nprint('ok')\n" % (ROOT, os.path.join(ROOT, "tests")))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, SMG_OVERLAP=form))
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), (p.stdout[-1500:], p.stderr[-1500:])


def _overlaps_large_query():
    import torch
    from sourmash_amd import device as smd, parallel
    from sourmash_amd.synth import synth_gather
    qh, dbh = synth_gather(n_query=140_000, n_db=4200, db_size=90)
    dbh[10] = np.zeros(0, dtype=np.uint64)
    dbh[11] = qh[1000:3000].copy()                                # a run of 2,000 consecutive query hashes: long slices
    dbh[12] = np.unique(np.concatenate([qh[::3], np.array([1, 2, 2**64 - 1], dtype=np.uint64)]))   # 46k hashes, below / above the query
    be = parallel.DeviceBackend()
    h, off = smd.pack_csr(dbh)
    q = torch.from_numpy(qh.view(np.int64).copy()).cuda()
    cnt = be.zeros((len(dbh),), torch.int64)
    be.overlaps(q, len(qh), h, off, len(dbh), cnt, 0)
    want = np.array([oracle.intersection_size(qh, d)[0] for d in dbh], dtype=np.int64)
    assert np.array_equal(cnt.cpu().numpy(), want)
    part = qh[::2].copy()                                         # consume half of the query: saturating subtraction
    pq = torch.from_numpy(part.view(np.int64).copy()).cuda()
    cnt[7] = 1                                                    # a counter someone lowered: must stop at 0
    be.overlaps(pq, len(part), h, off, len(dbh), cnt, 1)
    sub = np.array([oracle.intersection_size(part, d)[0] for d in dbh], dtype=np.int64)
    want[7] = 1
    assert np.array_equal(cnt.cpu().numpy(), np.maximum(want - sub, 0))


def test_overlaps_of_a_crowded_query_fall_back_to_the_row_kernel():
    """A query crowded into a sliver of the hash space: 100,000 of its 140,000 hashes are consecutive integers, so one bucket of the
    first-level table -- one range of either streaming form -- holds far more query hashes than their LDS slices have room for.
    overlap_ranges_launch reports that, and the one-wave-per-row kernel answers (pair_ops.hip: overlap_vector_launch): the counts of
    numpy's |Q ∩ row|, no error, default settings."""
    import torch
    from sourmash_amd import device as smd, parallel
    rng = np.random.default_rng(12)
    dense = np.arange(100_000, dtype=np.uint64) + np.uint64(1 << 40)
    spread = np.unique(rng.integers(0, 1 << 54, size=40_000, dtype=np.uint64))
    qh = np.unique(np.concatenate([dense, spread]))
    dbh = []
    for d in range(4300):
        own = rng.integers(0, 1 << 54, size=40, dtype=np.uint64)
        take = rng.choice(qh, size=int(rng.integers(0, 60)), replace=False)
        dbh.append(np.unique(np.concatenate([own, take])))
    dbh[7] = dense[500:900].copy()
    dbh[8] = np.zeros(0, dtype=np.uint64)
    be = parallel.DeviceBackend()
    h, off = smd.pack_csr(dbh)
    q = torch.from_numpy(qh.view(np.int64).copy()).cuda()
    cnt = be.zeros((len(dbh),), torch.int64)
    be.overlaps(q, len(qh), h, off, len(dbh), cnt, 0)
    want = np.array([np.isin(d, qh).sum() for d in dbh], dtype=np.int64)
    assert np.array_equal(cnt.cpu().numpy(), want) and want[7] == 400


def test_overlaps_wide_form_with_more_rows_than_one_round_of_workgroups():
    """103,000 rows: more than 400 per CU, so the wide form's grid is cut into full rounds (gather.hip: overlap_ranges_launch);
    and the same collection with 3 rows per workgroup (thousands of workgroups, most waves without a row).  Counts against
    numpy's |Q ∩ row| (linear.rs:52-113 per row)."""
    import os, subprocess, sys
    from conftest import ROOT
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_gather as t\nt._overlaps_many_rows()\nprint('ok')\n" % (ROOT, os.path.join(ROOT, "tests")))
    for extra in ({}, {"SMG_OVERLAP_ROWS": "3"}):
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, SMG_OVERLAP="wide", **extra))
        assert p.returncode == 0 and p.stdout.strip().endswith("ok"), (extra, p.stdout[-1500:], p.stderr[-1500:])


def _overlaps_many_rows():
    import torch
    from sourmash_amd import device as smd, parallel
    from sourmash_amd.synth import synth_gather
    qh, dbh = synth_gather(n_query=140_000, n_db=103_000, db_size=40)
    dbh[5] = np.zeros(0, dtype=np.uint64)
    dbh[102_999] = qh[500:900].copy()                             # the last row of the last workgroup: a long slice
    be = parallel.DeviceBackend()
    h, off = smd.pack_csr(dbh)
    q = torch.from_numpy(qh.view(np.int64).copy()).cuda()
    cnt = be.zeros((len(dbh),), torch.int64)
    be.overlaps(q, len(qh), h, off, len(dbh), cnt, 0)
    flat = np.concatenate(dbh)
    ends = np.cumsum([len(d) for d in dbh])
    hits = np.concatenate([[0], np.cumsum(np.isin(flat, qh).astype(np.int64))])
    want = hits[ends] - hits[np.concatenate([[0], ends[:-1]])]
    assert np.array_equal(cnt.cpu().numpy(), want)


def test_gather_random_shapes_every_builder(monkeypatch):
    """Random query / database shapes -- one row, empty rows, rows outside the query, duplicates of the winner, hash ranges
    from a few hundred values to all 64 bits, queries shorter than one range and a few ranges long -- through both builders
    of the index with every fill, the native loop and the exchange protocol on one rank, against the oracle's ordered picks
    and against |Q ∩ row| for the counters and the overlap pass."""
    import torch
    from sourmash_amd import device as smd, parallel
    rng = np.random.default_rng(77)
    be = parallel.DeviceBackend()
    for case in range(14):
        top_bits = int(rng.choice([10, 24, 54, 64]))
        hi = (1 << top_bits) - 1
        nq = int(rng.choice([1, 40, 3000, 40_000, 70_000]))
        ndb = int(rng.choice([1, 2, 17, 300, 900]))
        qh = np.unique(rng.integers(0, hi, size=nq, dtype=np.uint64, endpoint=True))
        dbh = []
        for _ in range(ndb):
            size = int(rng.integers(0, 700))
            from_q = rng.choice(qh, size=min(len(qh), int(size * rng.random())), replace=False)
            own = rng.integers(0, hi, size=size, dtype=np.uint64, endpoint=True)
            dbh.append(np.unique(np.concatenate([from_q, own]).astype(np.uint64)))
        if ndb > 5:
            dbh[1] = dbh[0].copy()                                    # a tie: the lower index wins
            dbh[2] = np.zeros(0, dtype=np.uint64)
        h, off = smd.pack_csr(dbh)
        fh, foff = oracle.make_csr(dbh)
        q = torch.from_numpy(qh.view(np.int64).copy()).cuda()
        want_cnt = np.array([oracle.intersection_size(qh, d)[0] for d in dbh], dtype=np.uint64)
        cnt = be.zeros((ndb,), torch.int64)
        be.overlaps(q, len(qh), h, off, ndb, cnt, 0)
        torch.cuda.synchronize()
        assert np.array_equal(cnt.cpu().numpy().view(np.uint64), want_cnt), ("overlaps", case)
        thr_bp = int(rng.choice([0, 3000, 60_000]))
        want = oracle.gather(qh, fh, foff, threshold_bp=thr_bp, scaled=1000)
        for build, fill in (("atomic", "staged"), ("ranges", "staged"), ("ranges", "direct")):
            monkeypatch.setenv("SMG_GATHER_BUILD", build)
            monkeypatch.setenv("SMG_GATHER_FILL", fill)
            st = be.gather_state(q, len(qh), h, off, ndb, 0)
            assert np.array_equal(st.counters(), want_cnt), ("counters", case, build, fill)
            del st
            got = parallel.gather_distributed(q, len(qh), h, off, ndb, 0, thr_bp, 1000, be)
            assert got == want, ("loop", case, build, fill, thr_bp)
        got = parallel.gather_distributed(q, len(qh), h, off, ndb, 0, thr_bp, 1000, be, stepwise=True)
        assert got == want, ("exchange protocol", case, thr_bp)


def test_rebegin_with_more_rounds_after_graph_replay():
    """A gather state that replayed its rounds as a captured hipGraph, armed again with a larger round cap: the result
    arrays are reallocated, and the graph (whose nodes hold the old pointers) has to go with them (round-2 advisor finding:
    use-after-free).  SMG_GATHER_GRAPH is read once per process, so this runs in its own interpreter."""
    import os, subprocess, sys
    from conftest import ROOT
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_gather as t\nt._rebegin_after_graph()\nprint('ok')\n" % (ROOT, os.path.join(ROOT, "tests")))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, SMG_GATHER_GRAPH="1"))
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), (p.stdout[-1500:], p.stderr[-1500:])


def _rebegin_after_graph():
    import torch
    from sourmash_amd import device as smd, parallel
    from sourmash_amd.synth import synth_gather
    qh, dbh = synth_gather(n_query=50_000, n_db=900, db_size=500)
    be = parallel.DeviceBackend()
    h, off = smd.pack_csr(dbh)
    q = torch.from_numpy(qh.view(np.int64).copy()).cuda()
    want = oracle.gather(qh, *oracle.make_csr(dbh), threshold_bp=0, scaled=1000)
    assert len(want) > 70
    st = be.gather_state(q, len(qh), h, off, len(dbh), 0)
    st.begin(0, 5)                                                # small cap: 5 rounds through the graph, then stop
    assert st.run() == want[:5]
    st.begin(0, len(dbh))                                         # larger cap: new result arrays, the loop carries on
    got = st.run()
    assert got == want[5:], (len(got), len(want))
    # the arena served the rebuilt arrays; a second state of the same shape makes no driver call at all
    a0 = smd.arena_stats()
    st2 = be.gather_state(q, len(qh), h, off, len(dbh), 0)
    st2.begin(0, 5)
    assert st2.run() == want[:5]
    del st2
    st3 = be.gather_state(q, len(qh), h, off, len(dbh), 0)
    st3.begin(0, len(dbh))
    assert st3.run() == want
    a1 = smd.arena_stats()
    s = st3.stats()
    assert s["build_driver_allocs"] == 0 and s["build_syncs"] == 2, s
    assert a1["reuse_hits"] > a0["reuse_hits"]


def test_persistent_loop_equals_two_kernel_rounds():
    """The resident one-launch loop (gather.hip: gather_loop_kernel; taken when the staged range builder made the index) and
    the two-kernel rounds (SMG_GATHER_LOOP=scan) give the oracle's ordered picks, the same final counters, and a state a
    second begin + run carries on from.  The switch is read once per process: one interpreter per loop form."""
    import os, subprocess, sys, json
    from conftest import ROOT
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_gather as t\nt._loop_forms()\n" % (ROOT, os.path.join(ROOT, "tests")))
    outs = {}
    for form in ("persistent", "scan"):
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, SMG_GATHER_LOOP=form, SMG_GATHER_BUILD="ranges"))
        assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
        outs[form] = json.loads(p.stdout.strip().splitlines()[-1])
    assert outs["persistent"] == outs["scan"]


def _loop_forms():
    import json
    import torch
    from sourmash_amd import device as smd, parallel
    from sourmash_amd.synth import synth_gather
    be = parallel.DeviceBackend()
    digest = []
    for nq, ndb, size, base in ((70_000, 3000, 400, 0), (40_000, 300, 900, 1000), (33_000, 5, 2000, 7), (100_000, 9000, 120, 0)):
        qh, dbh = synth_gather(n_query=nq, n_db=ndb, db_size=size)
        if ndb > 20:
            dbh[11] = dbh[5].copy()                               # a tie: the lower index wins
            dbh[3] = np.zeros(0, dtype=np.uint64)
            dbh[4] = np.array([1, 2, 3], dtype=np.uint64)
        h, off = smd.pack_csr(dbh)
        q = torch.from_numpy(qh.view(np.int64).copy()).cuda()
        fh, foff = oracle.make_csr(dbh)
        for thr_bp in (0, 40_000):
            want = [(base + i, c) for i, c in oracle.gather(qh, fh, foff, threshold_bp=thr_bp, scaled=1000, nthreads=8)]
            st = be.gather_state(q, len(qh), h, off, len(dbh), base)
            st.begin(int(np.ceil(thr_bp / 1000)), 7)               # stop after 7 rounds ...
            first = st.run()
            assert first == want[:7], (nq, ndb, thr_bp, first[:3], want[:3])
            mid = st.counters().copy()
            st.begin(int(np.ceil(thr_bp / 1000)), len(dbh))        # ... and carry on from the state that loop left
            rest = st.run()
            assert rest == want[7:], (nq, ndb, thr_bp, len(rest), len(want))
            # what is left of every counter = |row ∩ still-uncovered query|
            covered = set()
            for gi, _ in want:
                covered.update(int(x) for x in dbh[gi - base])
            left = np.array([x for x in qh if int(x) not in covered], dtype=np.uint64)
            want_left = np.array([oracle.intersection_size(left, d)[0] for d in dbh], dtype=np.uint64)
            assert np.array_equal(st.counters(), want_left), (nq, ndb, thr_bp)
            digest.append([len(want), int(mid.sum()), int(want_left.sum())])
    print(json.dumps(digest))


def test_resident_loop_steps_aside_when_the_device_is_not_idle(monkeypatch):
    """The resident loop kernel needs a workgroup on every CU at once.  With somebody else's kernel holding CUs (here: 24
    workgroups that keep 120 KB of LDS each for 0.4 s on another stream) its grid is not resident as a whole: the kernel
    gives up at its gate with nothing touched and the two-kernel rounds run on the same state -- the oracle's picks, no error
    (round 3 raised `Internal: the device was shared?` with half-consumed counters).  Also through CounterGather's object
    protocol and the one-call gather of a loaded set, which share the drain (index/__init__.py:856-909, search.py:755-779)."""
    import ctypes as C
    import torch
    from sourmash_amd import device as smd, parallel
    from sourmash_amd._lowlevel import lib
    from sourmash_amd.synth import synth_gather
    monkeypatch.setenv("SMG_GATHER_BUILD", "ranges")
    be = parallel.DeviceBackend()
    qh, dbh = synth_gather(n_query=70_000, n_db=3000, db_size=400)
    dbh[11] = dbh[5].copy()
    h, off = smd.pack_csr(dbh)
    q = torch.from_numpy(qh.view(np.int64).copy()).cuda()
    want = oracle.gather(qh, *oracle.make_csr(dbh), threshold_bp=0, scaled=1000, nthreads=8)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        torch.zeros(1, device="cuda").add_(1)                     # (the stream's hardware queue exists before the timed part)
    torch.cuda.synchronize()
    # idle device: the resident loop answers
    st = be.gather_state(q, len(qh), h, off, len(dbh), 0)
    assert st.loop_eligible(0), "this index should be able to run the resident loop"
    st.begin(0, len(dbh))
    assert st.run() == want
    assert st.stats()["loop_fallbacks"] == 0
    # occupied device: 24 CUs held for 0.4 s -> the gate (20 ms) gives up, the two-kernel rounds answer meanwhile
    st2 = be.gather_state(q, len(qh), h, off, len(dbh), 0)
    st2.begin(0, len(dbh))
    lib.smgpu_debug_hold_cus(24, 120 * 1024, 400_000, C.c_void_p(side.cuda_stream))
    got = st2.run()
    busy = not side.query()                                       # the holder outlived the gather: it really ran side by side
    torch.cuda.synchronize()
    assert got == want
    assert st2.stats()["loop_fallbacks"] == 1 and busy, (st2.stats(), busy)
    # and the same index afterwards, device idle again: back on the resident loop, same answer, no new fallback
    st2.begin(0, len(dbh))
    assert st2.run() == []                                        # (everything is consumed: the state carried over)
    st3 = be.gather_state(q, len(qh), h, off, len(dbh), 0)
    st3.begin(0, 9)
    assert st3.run() == want[:9] and st3.stats()["loop_fallbacks"] == 0
