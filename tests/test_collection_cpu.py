"""Bulk signature loading, host half: files -> CSR + manifest, no GPU.

Checked against (a) the reference's own zip fixtures and the counts its tests assert on them
(tests/test_index.py:821-906), (b) the oracle's independent reading of the same .sig files (python json),
(c) zips written here with Python's zipfile the way sourmash writes them (save_load.py:448-549)."""
import gzip
import io
import json
import os
import zipfile

import numpy as np
import pytest

import oracle
from conftest import golden


@pytest.fixture(scope="module")
def idx():
    from sourmash_amd import index
    return index


def _rows_of(col):
    off = col.offsets
    h = col.hashes
    return [h[off[i]:off[i + 1]] for i in range(len(col))]


def _oracle_sketches(path, ksize=None, moltype="dna"):
    "every sketch of a .sig/.sig.gz file via python's json: [(name, ksize, mins sorted)]"
    raw = open(path, "rb").read()
    if raw[:2] == b"\x1f\x8b":
        raw = gzip.decompress(raw)
    out = []
    for sig in json.loads(raw):
        for sk in sig["signatures"]:
            if sk["molecule"].lower() != moltype or (ksize and sk["ksize"] != ksize):
                continue
            out.append((sig.get("name", ""), sk["ksize"], np.array(sorted(sk["mins"]), dtype=np.uint64), sk["md5sum"]))
    return out


GATHER = sorted(f for f in os.listdir(golden("gather")) if f.startswith("GCF_"))


def test_directory_of_sigs_matches_independent_reader(idx):
    col = idx.Collection(golden("gather"), ksize=21, moltype="DNA")
    want = []
    for f in sorted(os.listdir(golden("gather"))):                       # the loader walks sorted names
        want += [(f, *sk) for sk in _oracle_sketches(golden("gather", f), ksize=21)]
    assert len(col) == len(want) == 13
    man = col.manifest
    for row, got, (f, name, k, mins, md5) in zip(man, _rows_of(col), want):
        assert np.array_equal(got, mins)
        assert row["name"] == name and row["md5"] == md5 and row["md5short"] == md5[:8]
        assert row["ksize"] == 21 and row["moltype"] == "DNA" and row["scaled"] == 10000 and row["num"] == 0
        assert row["n_hashes"] == len(mins) and row["internal_location"].endswith(f)
    assert col.total_hashes == sum(len(w[3]) for w in want)
    # the same files named one by one, in another order, through a path list
    listing = golden("gather", GATHER[3]) + "\n" + golden("gather", GATHER[0]) + "\n"
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as fh:
        fh.write(listing)
    try:
        col2 = idx.Collection(fh.name, ksize=21)
        assert [r["internal_location"] for r in col2.manifest] == [golden("gather", GATHER[3]), golden("gather", GATHER[0])]
    finally:
        os.unlink(fh.name)


def test_multi_ksize_file_selection_and_downsampling(idx):
    path = golden("ecoli", "GCF_000005845.2_ASM584v2_genomic.fna.gz.sig")
    for k, n, md5 in ((21, 4713, "2ebef1da342ce9a6a6039661612e2fee"), (31, 4476, "0a8632c67e6d88f737ddb510bef90337"),
                      (51, 4580, "5ac24aec77d8095e23c8a2514f3a16c6")):
        col = idx.Collection(path, ksize=k)
        assert len(col) == 1 and col.skipped == 2
        assert col.manifest[0]["n_hashes"] == n and col.manifest[0]["md5"] == md5
        assert np.array_equal(_rows_of(col)[0], _oracle_sketches(path, ksize=k)[0][2])
    with pytest.raises(ValueError) as e:                                 # three ksizes cannot share one CSR
        idx.Collection(path)
    assert "different ksizes cannot be compared" in str(e.value)
    # scaled=2000: keep h <= max_hash(2000); the manifest still describes the stored sketch
    col = idx.Collection(path, ksize=31, scaled=2000)
    mins = _oracle_sketches(path, ksize=31)[0][2]
    assert np.array_equal(_rows_of(col)[0], mins[mins <= np.uint64(int(2**64 / 2000))])
    assert col.manifest[0]["scaled"] == 1000 and col.manifest[0]["n_hashes"] == 4476
    assert len(idx.Collection(path, ksize=31, scaled=500)) == 0          # cannot upsample: not selected
    # num sketches are left out of a scaled selection, and load as-is without one
    num_sig = golden("num", "genome-s10.fa.gz.sig")
    assert len(idx.Collection(num_sig, ksize=21, moltype="DNA", scaled=1000)) == 0
    col = idx.Collection(num_sig, ksize=21, moltype="DNA")
    assert len(col) == 1 and col.manifest[0]["num"] == 500 and col.manifest[0]["scaled"] == 0


def test_reference_zip_fixtures(idx):
    # tests/test_index.py:821-906: 8 manifest rows; 2 of them DNA; per protein moltype 2
    z = golden("zips", "all.zip")
    dna = idx.Collection(z, moltype="DNA")
    assert len(dna) == 2 and dna.skipped == 6
    assert sorted(r["internal_location"] for r in dna.manifest) == ["dna-sig.noext", "dna-sig.sig.gz"]
    rows47, rows63 = _rows_of(dna)                                       # the 47.fa / 63.fa sketches of tests/golden/pairs
    assert np.array_equal(rows47, _oracle_sketches(golden("pairs", "47.fa.sig"), ksize=31)[0][2])
    assert np.array_equal(rows63, _oracle_sketches(golden("pairs", "63.fa.sig"), ksize=31)[0][2])
    for mt in ("protein", "dayhoff", "hp"):
        col = idx.Collection(z, moltype=mt, ksize=19)
        assert len(col) == 2 and [r["moltype"] for r in col.manifest] == [mt, mt]
        assert [r["ksize"] for r in col.manifest] == [19, 19] and [r["scaled"] for r in col.manifest] == [100, 100]
    with zipfile.ZipFile(z) as zf:
        want = json.loads(zf.read("protein/GCA_001593925.1_ASM159392v1_protein.faa.gz.sig"))[0]["signatures"][0]
    prot = idx.Collection(z, moltype="protein")
    assert np.array_equal(_rows_of(prot)[0], np.array(want["mins"], dtype=np.uint64)) and prot.manifest[0]["md5"] == want["md5sum"]
    with pytest.raises(ValueError):
        idx.Collection(z)                                                # DNA + protein in one CSR
    # a zip written by sourmash itself: stored .sig.gz members named by md5 + manifest; abundances are dropped
    t = golden("zips", "track_abund.zip")
    col = idx.Collection(t, ksize=31)
    assert [(r["md5"], r["n_hashes"], r["with_abundance"]) for r in col.manifest] == \
        [("09a08691ce52952152f0e866a59f6261", 5177, True), ("38729c6374925585db28916b82a6f513", 5238, True)]
    assert col.manifest[0]["name"] == "NC_009665.1 Shewanella baltica OS185, complete genome"     # quoted CSV field
    assert col.manifest[0]["filename"] == "podar-ref/47.fa"
    flat = _oracle_sketches(golden("pairs", "track_abund_47.fa.sig"), ksize=31)[0][2]
    assert np.array_equal(_rows_of(col)[0], flat)
    assert len(idx.Collection(t, ksize=21)) == 0


def _write_zip(path, members, manifest_rows=None, compression=zipfile.ZIP_STORED, force_zip64=False):
    with zipfile.ZipFile(path, "w", compression=compression) as zf:
        for name, data in members:
            with zf.open(zipfile.ZipInfo(name), "w", force_zip64=force_zip64) as fh:
                fh.write(data)
        if manifest_rows is not None:
            buf = io.StringIO()
            buf.write("# SOURMASH-MANIFEST-VERSION: 1.0\n")
            buf.write("internal_location,md5,md5short,ksize,moltype,num,scaled,n_hashes,with_abundance,name,filename\n")
            for r in manifest_rows:
                buf.write(",".join(str(x) for x in r) + "\n")
            zf.writestr("SOURMASH-MANIFEST.csv", buf.getvalue(), compress_type=zipfile.ZIP_DEFLATED)


def test_zips_written_here(idx, tmp_path):
    sigs = [golden("gather", f) for f in GATHER[:5]]
    raw = [open(p, "rb").read() for p in sigs]
    want = [_oracle_sketches(p, ksize=21)[0] for p in sigs]
    # 1. no manifest, deflated plain .sig members, an unrelated member, zip64 records forced
    z1 = str(tmp_path / "plain.zip")
    _write_zip(z1, [(f"sigs/{i}.sig", r) for i, r in enumerate(raw)] + [("README.txt", b"not a signature")],
               compression=zipfile.ZIP_DEFLATED, force_zip64=True)
    col = idx.Collection(z1, ksize=21)
    assert len(col) == 5 and [r["internal_location"] for r in col.manifest] == [f"sigs/{i}.sig" for i in range(5)]
    for got, w in zip(_rows_of(col), want):
        assert np.array_equal(got, w[2])
    # 2. manifest + stored .sig.gz members; the manifest decides what is opened (k=31 rows are never inflated:
    #    their member is garbage here) and in which order
    rows, members = [], []
    for i, (r, w) in enumerate(zip(raw, want)):
        members.append((f"signatures/{w[3]}.sig.gz", gzip.compress(r)))
        rows.append((f"signatures/{w[3]}.sig.gz", w[3], w[3][:8], 21, "DNA", 0, 10000, len(w[2]), 0, f'"{w[0]}"', "x.fa"))
    members.append(("signatures/bogus.sig.gz", b"\x00garbage"))
    rows.append(("signatures/bogus.sig.gz", "0" * 32, "0" * 8, 31, "DNA", 0, 10000, 5, 0, "bogus", "y.fa"))
    rows = rows[::-1]
    z2 = str(tmp_path / "manifest.zip")
    _write_zip(z2, members, manifest_rows=rows)
    col = idx.Collection(z2, ksize=21, moltype="DNA", threads=3)
    assert [r["md5"] for r in col.manifest] == [w[3] for w in want[::-1]]
    for got, w in zip(_rows_of(col), want[::-1]):
        assert np.array_equal(got, w[2])
    assert col.skipped == 1 + 2 * 5                                     # the bogus row + the k=31/51 sketches of each file
    with pytest.raises(Exception):
        idx.Collection(z2, ksize=31)                                     # now the bogus member is selected
    # 3. several inputs at once keep their order: zip, file, directory
    col = idx.Collection([z1, sigs[4], golden("gather")], ksize=21)
    assert len(col) == 5 + 1 + 13
    assert col.manifest[5]["internal_location"] == sigs[4]
    # 4. errors: a manifest naming a missing member; a file that is nothing we know
    z3 = str(tmp_path / "broken.zip")
    _write_zip(z3, members[:1], manifest_rows=rows)
    with pytest.raises(ValueError) as e:
        idx.Collection(z3, ksize=21)
    assert "missing member" in str(e.value)
    junk = tmp_path / "junk.bin"
    junk.write_bytes(b"\x01\x02\x03\x04 this is not a path list either")
    with pytest.raises(Exception):
        idx.Collection(str(junk))
    with pytest.raises(Exception):
        idx.Collection(str(tmp_path / "does-not-exist.sig"))


def test_scanner_field_semantics(idx, tmp_path):
    "minhash.rs:134-184 / signature.rs:569-659: unsorted mins are sorted, field order is free, missing fields fail"
    def sig(sketch, **top):
        d = {"class": "sourmash_signature", "email": "", "hash_function": "0.murmur64", "license": "CC0", "version": 0.4}
        d.update(top)
        d["signatures"] = [sketch]
        return d
    sk = {"mins": [30, 10, 20], "abundances": [3, 1, 2], "molecule": "DNA", "md5sum": "x" * 32, "max_hash": 1844674407370955,
          "seed": 42, "ksize": 31, "num": 0, "extra_field": {"nested": [1, {"a": 'b"c\\'}]}}
    p = tmp_path / "odd.sig"
    p.write_text(json.dumps([sig(sk, name="né \"quoted\", with comma", filename=None)]))
    col = idx.Collection(str(p))
    assert list(_rows_of(col)[0]) == [10, 20, 30]
    row = col.manifest[0]
    assert row["name"] == "né \"quoted\", with comma" and row["filename"] == "" and row["with_abundance"] is True
    assert row["scaled"] == 10000
    for drop in ("mins", "ksize", "seed", "max_hash", "md5sum", "molecule", "num"):
        bad = dict(sk)
        del bad[drop]
        p.write_text(json.dumps([sig(bad)]))
        with pytest.raises(Exception) as e:
            idx.Collection(str(p))
        assert f"missing field `{drop}`" in str(e.value)
    p.write_text(json.dumps([sig(dict(sk, molecule="rna"))]))
    with pytest.raises(ValueError):
        idx.Collection(str(p))
    p.write_text("[{\"signatures\": [")
    with pytest.raises(Exception):
        idx.Collection(str(p))
    p.write_text("[]")
    assert len(idx.Collection(str(p))) == 0
    # a bare object instead of a list, gzip-compressed
    pz = tmp_path / "one.sig.gz"
    pz.write_bytes(gzip.compress(json.dumps(sig(sk, name="solo")).encode()))
    assert idx.Collection(str(pz)).manifest[0]["name"] == "solo"


def test_manifest_csv_round_trip(idx):
    col = idx.Collection(golden("zips", "track_abund.zip"))
    text = col.manifest_csv
    assert text.startswith("# SOURMASH-MANIFEST-VERSION: 1.0\ninternal_location,md5,md5short,ksize,moltype,num,scaled,"
                           "n_hashes,with_abundance,name,filename\r\n")
    with zipfile.ZipFile(golden("zips", "track_abund.zip")) as zf:
        assert text == zf.read("SOURMASH-MANIFEST.csv").decode()         # byte-identical to the reference's writer
