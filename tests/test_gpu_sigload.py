"""SketchSet.load with the device inflating the .sig.gz documents and parsing their hash arrays (csrc/sigload.hpp, csrc/sigjson.hip)
against the host loader of the same library (Collection: csrc/collection.hpp, which tests/test_collection_cpu.py pins to the
reference's files): same rows, same manifest, same hashes, same errors -- for well-formed collections and for documents the
device hands back to the host.  Run with -m gpu."""
import gzip
import hashlib
import io
import json
import os
import zipfile

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


def counters():
    import ctypes as C
    from sourmash_amd._lowlevel import lib
    out = (C.c_uint64 * 2)()
    lib.smgpu_sigload_counters(out)
    return int(out[0]), int(out[1])


def md5_of(ksize, mins):
    return hashlib.md5((str(ksize) + "".join(map(str, mins))).encode()).hexdigest()


def sketch_json(mins, ksize=31, max_hash=18446744073709551, num=0, molecule="dna", abund=None, md5=None, seed=42):
    d = {"num": num, "ksize": ksize, "seed": seed, "max_hash": max_hash, "mins": [int(m) for m in mins],
         "md5sum": md5_of(ksize, mins) if md5 is None else md5, "molecule": molecule}
    if abund is not None:
        d["abundances"] = [int(a) for a in abund]
    return d


def sig_doc(sketches, name="a name", filename="f.fa", **dumps):
    doc = [{"class": "sourmash_signature", "email": "", "hash_function": "0.murmur64", "filename": filename, "name": name,
            "license": "CC0", "signatures": sketches, "version": 0.4}]
    return json.dumps(doc, **(dumps or {"separators": (",", ":")})).encode()


def rand_mins(rng, n, top=18446744073709551):
    return np.unique(rng.integers(1, top, size=n, dtype=np.uint64)).tolist()


def same_collection(sm, paths, **sel):
    "device loader == host loader on everything a SketchSet holds"
    from sourmash_amd.index import SketchSet, Collection
    host = Collection(paths, **sel)
    dev = SketchSet.load(paths, **sel)
    assert len(dev) == len(host) and dev.total_hashes == host.total_hashes and dev.skipped == host.skipped
    assert dev.manifest == host.manifest
    off = host.offsets
    assert list(dev.sizes) == list(np.diff(off))
    hh = host.hashes
    for row in range(len(dev)):
        got = np.array(sorted(dev.minhash(row).hashes), dtype=np.uint64)
        assert np.array_equal(got, hh[off[row]:off[row + 1]]), row
    if len(host):
        assert dev.params == host.to_device().params
    return dev, host


def write_zip(path, docs, manifest_rows=None, compress=True):
    with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as zf:
        man = io.StringIO()
        man.write("# SOURMASH-MANIFEST-VERSION: 1.0\n")
        man.write("internal_location,md5,md5short,ksize,moltype,num,scaled,n_hashes,with_abundance,name,filename\r\n")
        for i, (doc, rows) in enumerate(docs):
            loc = f"signatures/doc{i}.sig.gz" if compress else f"signatures/doc{i}.sig"
            zf.writestr(loc, gzip.compress(doc, compresslevel=1 + i % 9) if compress else doc)
            for (md5, ksize, moltype, num, scaled, n, ab) in rows:
                man.write(f"{loc},{md5},{md5[:8]},{ksize},{moltype},{num},{scaled},{n},{int(ab)},name {i},f{i}.fa\r\n")
        if manifest_rows is not False:
            zf.writestr("SOURMASH-MANIFEST.csv", man.getvalue(), compress_type=zipfile.ZIP_DEFLATED)


def test_a_zip_of_sketches_loads_the_same_on_both_paths(sm, tmp_path):
    rng = np.random.default_rng(1)
    docs = []
    for i in range(300):
        mins = rand_mins(rng, int(rng.integers(0, 6000)))
        docs.append((sig_doc([sketch_json(mins)], name=f"genome {i}", filename=f"g{i}.fa"),
                     [(md5_of(31, mins), 31, "DNA", 0, 1000, len(mins), False)]))
    z = str(tmp_path / "coll.zip")
    write_zip(z, docs)
    before = counters()
    dev, host = same_collection(sm, z, ksize=31, moltype="DNA")
    after = counters()
    assert len(dev) == 300 and after[0] - before[0] == 300 and after[1] == before[1]      # every document went through the device
    # down-sampling on load: the kept prefix is counted on the device
    same_collection(sm, z, ksize=31, moltype="DNA", scaled=4000)
    same_collection(sm, z, ksize=31, moltype="DNA", scaled=1000)


def test_documents_with_several_sketches_selection_and_abundances(sm, tmp_path):
    rng = np.random.default_rng(2)
    docs = []
    for i in range(40):
        sks, rows = [], []
        for ksize, mol in ((21, "dna"), (31, "dna"), (51, "dna"), (30, "protein")):
            mins = rand_mins(rng, 200 + 50 * i)
            ab = rng.integers(1, 50, size=len(mins)).tolist() if i % 3 == 0 else None
            sks.append(sketch_json(mins, ksize=ksize, molecule=mol, abund=ab))
            rows.append((md5_of(ksize, mins), ksize if mol == "dna" else ksize // 3, "DNA" if mol == "dna" else "protein", 0, 1000, len(mins), ab is not None))
        docs.append((sig_doc(sks, name=f"multi {i}"), rows))
    z = str(tmp_path / "multi.zip")
    write_zip(z, docs)
    for sel in (dict(ksize=31, moltype="DNA"), dict(ksize=21), dict(ksize=10, moltype="protein"), dict(ksize=51, moltype="DNA", scaled=2000)):
        dev, _ = same_collection(sm, z, **sel)
        assert len(dev) == 40
    d = tmp_path / "dir"                                               # the same documents as files in a directory, no manifest
    d.mkdir()
    for i, (doc, _) in enumerate(docs[:12]):
        (d / f"s{i:02d}.sig.gz").write_bytes(gzip.compress(doc))
    same_collection(sm, str(d), ksize=31, moltype="DNA")


def test_golden_files_and_plain_json_take_the_host_path_with_the_same_result(sm, tmp_path):
    paths = [golden("gather", f) for f in sorted(os.listdir(golden("gather"))) if f.startswith("GCF_")]
    before = counters()
    dev, _ = same_collection(sm, paths, ksize=21, moltype="DNA")
    assert len(dev) == 12 and counters()[1] - before[1] == 12         # plain .sig files: the host parser
    z = str(tmp_path / "plain.zip")
    write_zip(z, [(open(p, "rb").read(), []) for p in paths], manifest_rows=False, compress=False)
    same_collection(sm, z, ksize=21, moltype="DNA")
    gz = []
    for p in paths:                                                    # ... and gzipped: the device
        q = tmp_path / (os.path.basename(p) + ".gz")
        q.write_bytes(gzip.compress(open(p, "rb").read()))
        gz.append(str(q))
    before = counters()
    same_collection(sm, gz, ksize=21, moltype="DNA")
    assert counters()[0] - before[0] == 12


def test_unusual_documents_go_to_the_host_parser_and_come_out_the_same(sm, tmp_path):
    rng = np.random.default_rng(3)
    mins = rand_mins(rng, 500)
    shuffled = list(mins)
    rng.shuffle(shuffled)
    cases = {
        "unsorted": sig_doc([sketch_json(shuffled, md5=md5_of(31, mins))]),                      # minhash.rs:161-171: sorted on load
        "repeated": sig_doc([sketch_json(mins + mins[-1:], md5=md5_of(31, mins))]),
        "spaces": sig_doc([sketch_json(mins)], indent=2),                                        # json.dumps with white space everywhere
        "spaces_compact_keys": sig_doc([sketch_json(mins)], separators=(", ", ": ")),
        "floats": sig_doc([sketch_json(mins)]).replace(b'"mins":[%d,' % mins[0], b'"mins":[%d.0,' % mins[0]),
        "empty_mins": sig_doc([sketch_json([])]),
        "one_value": sig_doc([sketch_json(mins[:1])]),
        "empty_md5": sig_doc([sketch_json(mins, md5="")]),
        "many_arrays": sig_doc([sketch_json(rand_mins(rng, 30), ksize=31) for _ in range(11)]),
        "nested_key": sig_doc([sketch_json(mins)]).replace(b'"license":"CC0"', b'"license":"CC0","extra":{"mins":[1,2,3]}'),
        "mins_in_a_string": sig_doc([sketch_json(mins)], name='the "mins":[ of it', filename='x\\"mins\\":[1]'),
        "big_values": sig_doc([sketch_json([1, 2**63, 2**64 - 1], max_hash=0, num=500)]),
        "num_sketch": sig_doc([sketch_json(mins[:100], max_hash=0, num=100)]),
        "name_null": sig_doc([sketch_json(mins)]).replace(b'"name":"a name"', b'"name":null'),
    }
    for name, doc in cases.items():
        p = tmp_path / f"{name}.sig.gz"
        p.write_bytes(gzip.compress(doc))
        sel = dict(ksize=31, moltype="DNA")
        dev, host = same_collection(sm, str(p), **sel)
        assert len(dev) == (11 if name == "many_arrays" else 1), name
    # all of them in one zip next to ordinary ones: the rows keep the input order
    ordinary = [sig_doc([sketch_json(rand_mins(rng, 300))], name=f"ok {i}") for i in range(5)]
    mixed = [ordinary[0], cases["unsorted"], ordinary[1], cases["spaces"], cases["floats"], ordinary[2], cases["empty_md5"], ordinary[3], cases["mins_in_a_string"], ordinary[4]]
    z = str(tmp_path / "mixed.zip")
    write_zip(z, [(d, []) for d in mixed], manifest_rows=False)
    dev, _ = same_collection(sm, z, ksize=31, moltype="DNA")
    assert [m["name"] for m in dev.manifest][:3] == ["ok 0", "a name", "ok 1"]


def test_malformed_documents_raise_what_the_host_loader_raises(sm, tmp_path):
    from sourmash_amd.index import SketchSet, Collection
    rng = np.random.default_rng(4)
    mins = rand_mins(rng, 100)
    good = sig_doc([sketch_json(mins)])
    bad = {
        "letters_in_mins": good.replace(b'"mins":[', b'"mins":[12x,'),
        "no_closing_bracket": good[:good.index(b'"md5sum"') - 2],
        "missing_field": good.replace(b'"seed":42,', b''),
        "trailing": good + b"xyz",
        "not_json": b"hello there, this is not a signature at all",
        "double_comma": good.replace(b'"mins":[', b'"mins":[,'),
    }
    for name, doc in bad.items():
        p = tmp_path / f"{name}.sig.gz"
        p.write_bytes(gzip.compress(doc))
        with pytest.raises(Exception) as he:
            Collection(str(p), ksize=31)
        with pytest.raises(Exception) as de:
            SketchSet.load(str(p), ksize=31)
        assert type(de.value) is type(he.value) and str(de.value) == str(he.value), name
    cut = tmp_path / "cut.sig.gz"                                      # a damaged gzip stream: refused by the device inflater, reported by the host's
    blob = gzip.compress(good)
    cut.write_bytes(blob[:len(blob) // 2])
    with pytest.raises(Exception):
        SketchSet.load(str(cut), ksize=31)
    mism = tmp_path / "mism.zip"                                       # two scaled values in one CSR: the compatibility check of the assembly
    write_zip(str(mism), [(sig_doc([sketch_json(mins)]), []), (sig_doc([sketch_json(mins, max_hash=1844674407370955)]), [])], manifest_rows=False)
    with pytest.raises(Exception) as he:
        Collection(str(mism), ksize=31)
    with pytest.raises(Exception) as de:
        SketchSet.load(str(mism), ksize=31)
    assert str(de.value) == str(he.value)


def test_a_larger_collection_and_what_is_done_with_it(sm, tmp_path):
    "3,000 sketches of ~2,000 hashes: several thousand gzip members inflated in one pass, compare / gather on the loaded rows"
    import oracle
    from sourmash_amd.index import SketchSet
    from sourmash_amd.synth import synth_sketches
    sk = synth_sketches(3000, pool_size=20000)
    docs = [(sig_doc([sketch_json(h.tolist())], name=f"s{i}"), [(md5_of(31, h.tolist()), 31, "DNA", 0, 1000, len(h), False)]) for i, h in enumerate(sk)]
    z = str(tmp_path / "big.zip")
    write_zip(z, docs)
    db = SketchSet.load(z, ksize=31, moltype="DNA")
    assert list(db.sizes) == [len(h) for h in sk]
    sub = db.subset(np.arange(0, 3000, 10))
    common, _ = sub.compare(jaccard=False)
    want, _ = oracle.compare_all_pairs(*oracle.make_csr(sk[::10]), nthreads=8)
    assert np.array_equal(common, want)


def test_many_groups_through_both_workers(sm, tmp_path):
    """Round 6: groups of documents are dealt to two workers, each with a stream and a staging buffer of its own (csrc/sigload.hpp).
    A collection cut into dozens of small groups (SMG_SIGLOAD_GROUP_BYTES, read once: a subprocess), with documents the device hands
    back to the host scattered through it -- rows, manifest and hashes equal the host loader's, in input order, five times over."""
    import subprocess, sys
    from conftest import ROOT
    rng = np.random.default_rng(41)
    docs = []
    for i in range(700):
        mins = rand_mins(rng, int(rng.integers(1, 1500)))
        if i % 97 == 5:                                             # unsorted: parsed (and re-sorted) by the host
            shuffled = mins[::-1]
            docs.append((sig_doc([sketch_json(shuffled, md5=md5_of(31, mins))], name=f"s{i}"), [(md5_of(31, mins), 31, "DNA", 0, 1000, len(mins), False)]))
        elif i % 53 == 7:                                           # two sketches in one document
            other = rand_mins(rng, 300)
            docs.append((sig_doc([sketch_json(mins), sketch_json(other, ksize=21)], name=f"s{i}"),
                         [(md5_of(31, mins), 31, "DNA", 0, 1000, len(mins), False), (md5_of(21, other), 21, "DNA", 0, 1000, len(other), False)]))
        else:
            docs.append((sig_doc([sketch_json(mins)], name=f"s{i}"), [(md5_of(31, mins), 31, "DNA", 0, 1000, len(mins), False)]))
    z = str(tmp_path / "groups.zip")
    write_zip(z, docs)
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import torch, sourmash_amd, test_gpu_sigload as t\n"
            "for rep in range(5):\n"
            "    dev, host = t.same_collection(sourmash_amd, %r, ksize=31, moltype='DNA')\n"
            "    assert len(dev) == 700\n"
            "print('ok', t.counters())\n" % (ROOT, os.path.join(ROOT, "tests"), z))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, SMG_SIGLOAD_GROUP_BYTES="65536"))
    assert p.returncode == 0 and p.stdout.strip().startswith("ok"), (p.stdout[-1500:], p.stderr[-1500:])
