"""Bulk loading straight into HBM (SketchSet.load) and the row-number API on top of it: golden gather order,
compare vs the oracle, sketches materialised from rows.  Run with -m gpu."""
import gzip
import io
import os
import zipfile

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


def _gcf_paths():
    return [golden("gather", f) for f in sorted(os.listdir(golden("gather"))) if f.startswith("GCF_")]


def test_golden_gather_from_files(sm):
    # tests/test_index_protocol.py:1057-1097 of the reference, with the database never becoming Python objects
    from sourmash_amd.index import SketchSet
    db = SketchSet.load(_gcf_paths(), ksize=21, moltype="DNA")
    assert len(db) == 12 and db.params == (21, "DNA", 42, 10000, 0)
    query = sm.load_one_signature_from_json(golden("gather", "combined.sig"), ksize=21).minhash
    got = [(db.manifest[row]["name"].split()[0], n) for row, n in db.gather(query)]
    assert got == GOLDEN_GATHER
    assert [n for _, n in db.gather(query, threshold_bp=500_000)] == [487, 192, 169, 157, 152, 92, 76, 59]   # >= 50 hashes
    # one-pass overlaps = the add-time counters of CounterGather
    rows = [np.sort(next(d["mins"] for d in oracle.read_sig_json(p) if d["ksize"] == 21)) for p in _gcf_paths()]
    q = np.array(sorted(query.hashes), dtype=np.uint64)
    assert list(db.overlaps(query)) == [oracle.intersection_size(q, r)[0] for r in rows]
    assert list(db.sizes) == [len(r) for r in rows] and db.total_hashes == sum(len(r) for r in rows)


def test_compare_and_rows_from_a_zip(sm, tmp_path):
    from sourmash_amd.index import SketchSet
    from sourmash_amd.synth import synth_sketches
    sk = synth_sketches(150, pool_size=4000)
    # write the collection the way sourmash does: signatures/<md5>.sig.gz + manifest
    sigs = []
    for i, h in enumerate(sk):
        mh = sm.MinHash(0, 31, scaled=1000)
        mh.add_many(h)
        sigs.append(sm.SourmashSignature(mh, name=f"synthetic, number {i}", filename=f"s{i}.fa"))
    zpath = str(tmp_path / "coll.zip")
    with zipfile.ZipFile(zpath, "w", zipfile.ZIP_STORED) as zf:
        man = io.StringIO()
        man.write("# SOURMASH-MANIFEST-VERSION: 1.0\n")
        man.write("internal_location,md5,md5short,ksize,moltype,num,scaled,n_hashes,with_abundance,name,filename\r\n")
        for ss in sigs:
            md5 = ss.md5sum()
            loc = f"signatures/{md5}.sig.gz"
            if loc not in zf.namelist():                                # the planted duplicate shares its md5
                zf.writestr(loc, gzip.compress(sm.save_signatures_to_json([ss])))
            man.write(f'{loc},{md5},{md5[:8]},31,DNA,0,1000,{len(ss.minhash)},0,"{ss.name}",{ss.filename}\r\n')
        zf.writestr("SOURMASH-MANIFEST.csv", man.getvalue())
    db = SketchSet.load(zpath, ksize=31, moltype="DNA", scaled=1000, threads=4)
    assert len(db) == len(sk)
    common, jac = db.compare()
    wc, wj = oracle.compare_all_pairs(*oracle.make_csr(sk), nthreads=8)
    assert np.array_equal(common, wc) and np.array_equal(jac.view(np.uint64), wj.view(np.uint64))
    # rows come back as full objects when asked for
    for row in (0, 77, len(sk) - 1):
        ss = db.signature(row)
        assert ss.name == sigs[row].name and ss.md5sum() == db.manifest[row]["md5"] == sigs[row].md5sum()
        assert ss.minhash == sigs[row].minhash
    # search / prefetch by row number == the object route over a LinearIndex
    from sourmash_amd.index import LinearIndex
    lin = LinearIndex(sigs)
    q = sigs[5]
    for kw in ({}, {"do_containment": True}, {"do_max_containment": True}):
        want = lin.search(q, threshold=0.08, **kw)
        got = db.search(q.minhash, threshold=0.08, **kw)
        assert [round(s, 12) for s, _ in got] == [round(r.score, 12) for r in want]
        assert {sigs[row].name for _, row in got} == {r.signature.name for r in want}
    assert db.search(q.minhash, threshold=0.0, best_only=True)[0][1] in (5, len(sk) - 4)       # itself or its planted duplicate
    pre = db.prefetch(q.minhash, threshold_bp=100_000)
    assert [(row, n) for row, n in pre] == [(i, int(wc[5, i])) for i in range(len(sk)) if wc[5, i] >= 100]
    # downsampling at load time == downsampling the objects
    db2 = SketchSet.load(zpath, ksize=31, scaled=4000)
    assert db2.params[3] == 4000
    for row in (3, 50):
        assert db2.minhash(row) == sigs[row].minhash.downsample(scaled=4000)
    # the object route and the file route agree on gather
    from sourmash_amd.index import CounterGather
    query = sigs[-1]                                                    # the pool-wide sketch: everything overlaps it
    cg = CounterGather(query)
    cg.add_many(sigs[:-1])
    by_obj = cg.gather_all(threshold_bp=20_000)
    by_file = db.gather(query.minhash, threshold_bp=20_000)
    md5_of = [s.md5sum() for s in sigs]
    assert by_file[0] == (len(sk) - 1, len(sk[-1]))                     # in the file set the query itself is a row
    assert len(by_obj) > 3 and by_file[1:] == []                       # ... and covers everything at once
    assert by_obj == [(md5_of[i], n) for i, n in oracle.gather(np.array(sk[-1]), *oracle.make_csr(sk[:-1]), threshold_bp=20_000, scaled=1000)]


def test_load_needs_gpu_only_for_upload(sm):
    from sourmash_amd.index import Collection
    col = Collection(_gcf_paths()[:3], ksize=21)
    dev = col.to_device()
    assert len(dev) == 3 and list(dev.sizes) == list(np.diff(col.offsets))
    assert [r["md5"] for r in dev.manifest] == [r["md5"] for r in col.manifest]
