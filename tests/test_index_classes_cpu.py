"""Host side of the collection classes (no GPU): manifests, picklists, zip storage, the save classes and the
signature walk of every Index class.  Cases follow the reference's tests/test_manifest.py, tests/test_picklist.py and
the loading/saving parts of tests/test_index.py; scoring on the GPU is covered by tests/test_gpu_index_protocol.py."""
import io
import os
import zipfile

import pytest

from conftest import golden


@pytest.fixture(scope="module")
def sm():
    import sourmash_amd
    return sourmash_amd


@pytest.fixture()
def three(sm):
    return [sm.load_one_signature_from_json(golden("pairs", "2.fa.sig"), ksize=31),
            sm.load_one_signature_from_json(golden("pairs", "47.fa.sig")),
            sm.load_one_signature_from_json(golden("pairs", "63.fa.sig"))]


def _md5s(sigs):
    return sorted(ss.md5sum() for ss in sigs)


def test_manifest_rows_csv_round_trip_and_select(sm, three, tmp_path):
    from sourmash_amd.manifest import BaseCollectionManifest, CollectionManifest
    mf = CollectionManifest.create_manifest((ss, f"loc{i}") for i, ss in enumerate(three))
    assert isinstance(mf, BaseCollectionManifest) and len(mf) == 3 and bool(mf)
    row = mf.rows[1]
    assert row["md5"] == three[1].md5sum() and row["md5short"] == row["md5"][:8] and row["ksize"] == 31
    assert row["moltype"] == "DNA" and row["scaled"] == 1000 and row["num"] == 0 and row["with_abundance"] is False
    assert row["n_hashes"] == len(three[1].minhash) and row["internal_location"] == "loc1" and row["signature"] is three[1]
    assert list(mf.locations()) == ["loc0", "loc1", "loc2"]
    assert all(ss in mf for ss in three)
    # CSV: version line, header, rows; booleans and ints come back typed; signatures are not kept
    path = str(tmp_path / "mf.csv")
    mf.write_to_filename(path)
    with pytest.raises(Exception, match="already exists"):
        mf.write_to_filename(path)
    text = open(path).read()
    assert text.startswith("# SOURMASH-MANIFEST-VERSION: 1.0\n" + ",".join(CollectionManifest.required_keys))
    back = CollectionManifest.load_from_filename(path)
    assert back == mf and back.rows[0]["signature"] is None and back.rows[0]["with_abundance"] is False
    with pytest.raises(ValueError, match="version header"):
        CollectionManifest.load_from_csv(io.StringIO("md5,ksize\n"))
    with pytest.raises(ValueError, match="missing column"):
        CollectionManifest.load_from_csv(io.StringIO("# SOURMASH-MANIFEST-VERSION: 1.0\nmd5,ksize\n"))
    # selection
    assert len(back.select_to_manifest(ksize=31)) == 3 and len(back.select_to_manifest(ksize=21)) == 0
    assert len(back.select_to_manifest(moltype="protein")) == 0 and len(back.select_to_manifest(num=500)) == 0
    assert len(back.select_to_manifest(scaled=1000, containment=True)) == 3 and len(back.select_to_manifest(abund=True)) == 0
    with pytest.raises(ValueError):
        list(back._select(ksize="31"))
    assert len(back.filter_rows(lambda r: "NC_009665" in r["name"])) == 1
    assert len(back.filter_on_columns(lambda vals: any("NC_011663" in v for v in vals), ["name", "filename"])) == 1
    both = back + mf
    assert len(both) == 6
    with pytest.raises(Exception):
        back += back


def test_picklists(sm, three, tmp_path):
    from sourmash_amd.manifest import CollectionManifest
    from sourmash_amd.picklist import PickStyle, SignaturePicklist, passes_all_picklists
    ss2, ss47, ss63 = three
    pl = SignaturePicklist("md5prefix8")
    pl.init([ss47.md5sum()[:8]])
    assert ss47 in pl and ss63 not in pl and pl.found == {ss47.md5sum()[:8]} and pl.n_queries == 2
    ex = SignaturePicklist("md5", pickstyle=PickStyle.EXCLUDE)
    ex.init([ss47.md5sum()])
    assert ss47 not in ex and ss63 in ex and [s.md5sum() for s in ex.filter(three)] == [ss2.md5sum(), ss63.md5sum()]
    assert passes_all_picklists(ss63, [ex]) and not passes_all_picklists(ss63, [ex, pl])
    ident = SignaturePicklist("identprefix")
    ident.init(["NC_009665"])
    assert ss47 in ident and ss63 not in ident
    with pytest.raises(ValueError):
        SignaturePicklist("nope")
    with pytest.raises(ValueError):
        SignaturePicklist("manifest", column_name="x")
    # from a CSV file, with duplicates and empty cells
    path = tmp_path / "pick.csv"
    path.write_text("name,md5\nx,%s\ny,%s\nz,\n" % (ss47.md5sum(), ss47.md5sum()))
    pf = SignaturePicklist.from_picklist_args(f"{path}:md5:md5")
    n_empty, dups = pf.load()
    assert n_empty == 1 and dups == {ss47.md5sum()} and ss47 in pf and ss2 not in pf
    assert SignaturePicklist.from_picklist_args(f"{path}:md5:md5:exclude").pickstyle == PickStyle.EXCLUDE
    for bad in (f"{path}:md5", f"{path}:md5:md5:sideways"):
        with pytest.raises(ValueError):
            SignaturePicklist.from_picklist_args(bad)
    with pytest.raises(ValueError, match="not in pickfile"):
        SignaturePicklist.from_picklist_args(f"{path}:nocol:md5").load()
    # a manifest is a picklist on (ident, md5 prefix), and a manifest CSV can be used as a pick file
    mf = CollectionManifest.create_manifest([(ss47, "a"), (ss63, "b")])
    mpl = mf.to_picklist()
    assert ss47 in mpl and ss2 not in mpl and mpl.matches_manifest_row(mf.rows[1])
    mpath = str(tmp_path / "mf.csv")
    mf.write_to_filename(mpath)
    fpl = SignaturePicklist.from_picklist_args(f"{mpath}::manifest")
    fpl.load()
    assert ss63 in fpl and ss2 not in fpl


def test_save_classes_and_index_walks(sm, three, tmp_path):
    from sourmash_amd.index import (LazyLinearIndex, LinearIndex, MultiIndex, StandaloneManifestIndex,
                                    ZipFileLinearIndex)
    from sourmash_amd.manifest import CollectionManifest
    from sourmash_amd.save_load import (SaveSignatures_Directory, SaveSignatures_NoOutput, SaveSignatures_SigFile,
                                        SaveSignatures_ZipFile, SaveSignaturesToLocation, load_file_as_index,
                                        load_file_as_signatures)
    want = _md5s(three)
    assert isinstance(SaveSignaturesToLocation(None), SaveSignatures_NoOutput)
    # one JSON file (plain and gzip)
    for name in ("all.sig", "all.sig.gz"):
        loc = str(tmp_path / name)
        with SaveSignaturesToLocation(loc) as save:
            assert isinstance(save, SaveSignatures_SigFile)
            save.add_many(three)
        assert len(save) == 3 and _md5s(load_file_as_signatures(loc)) == want
        assert open(loc, "rb").read(2) == (b"\x1f\x8b" if name.endswith(".gz") else b'[{')
    # a directory of <md5>.sig.gz, a second copy gets a numbered name
    dloc = str(tmp_path / "dir") + "/"
    with SaveSignaturesToLocation(dloc) as save:
        assert isinstance(save, SaveSignatures_Directory)
        save.add_many(three)
        save.add(three[0])
    assert sorted(os.listdir(dloc)) == sorted([m + ".sig.gz" for m in want] + [three[0].md5sum() + "_0.sig.gz"])
    midx = load_file_as_index(dloc)
    assert isinstance(midx, MultiIndex) and len(midx) == 4 and midx.location == dloc
    locs = [loc for _, loc in midx.signatures_with_location()]
    assert all(loc.startswith(dloc) and os.path.exists(loc) for loc in locs)
    assert len(midx.select(ksize=31, moltype="DNA")) == 4 and len(midx.select(ksize=21)) == 0
    # a zip file: stored members under signatures/, deflated manifest; reopened, appended to
    zloc = str(tmp_path / "coll.zip")
    with SaveSignaturesToLocation(zloc) as save:
        assert isinstance(save, SaveSignatures_ZipFile)
        save.add(three[0])
        save.add(three[1])
    with zipfile.ZipFile(zloc) as zf:
        names = zf.namelist()
        assert sorted(names) == sorted(["SOURMASH-MANIFEST.csv"] + [f"signatures/{m}.sig.gz" for m in _md5s(three[:2])])
        assert zf.getinfo("SOURMASH-MANIFEST.csv").compress_type == zipfile.ZIP_DEFLATED
        assert zf.getinfo(names[0]).compress_type == zipfile.ZIP_STORED
    with SaveSignaturesToLocation(zloc) as save:
        save.add(three[2])
        save.add(three[0])                                   # same bytes: same member, a second manifest row
    zidx = ZipFileLinearIndex.load(zloc)
    assert zidx.manifest is not None and len(zidx) == 4 and bool(zidx) and zidx.location == os.path.abspath(zloc)
    assert sorted(set(_md5s(zidx.signatures()))) == want
    with zipfile.ZipFile(zloc) as zf:
        assert len(zf.namelist()) == 4 and zf.namelist().count("SOURMASH-MANIFEST.csv") == 1
    assert len(zidx.select(ksize=31)) == 4 and len(zidx.select(moltype="protein")) == 0
    assert _md5s(ss for ss, _ in zidx._signatures_with_internal()) == want
    # without using the manifest: selection is remembered and applied on the walk
    raw = ZipFileLinearIndex.load(zloc, use_manifest=False)
    assert raw.manifest is None and len(raw) == 3 and len(raw.select(ksize=31).select(moltype="DNA")) == 3
    with pytest.raises(ValueError, match="incompatible select"):
        raw.select(ksize=31).select(ksize=21)
    # a zip without a manifest cannot be appended to
    bare = str(tmp_path / "bare.zip")
    with zipfile.ZipFile(bare, "w") as zf:
        zf.writestr("x.sig", sm.save_signatures_to_json([three[0]]))
    assert _md5s(load_file_as_signatures(bare)) == [three[0].md5sum()]
    with pytest.raises(ValueError, match="without a manifest"):
        SaveSignatures_ZipFile(bare).open()
    with pytest.raises(FileNotFoundError):
        ZipFileLinearIndex.load(str(tmp_path / "missing.zip"))
    # lazy wrapper: selection dictionaries merge, conflicts raise
    lazy = LazyLinearIndex(LinearIndex(three, filename="mem"))
    assert len(lazy) == 3 and bool(lazy) and _md5s(lazy.select(ksize=31).signatures()) == want
    assert not LazyLinearIndex(LinearIndex(three)).select(ksize=21)
    with pytest.raises(ValueError, match="two different values"):
        lazy.select(ksize=31).select(ksize=21)
    # a standalone manifest naming members by relative path
    mf = CollectionManifest.create_manifest(((ss, "all.sig") for ss in three), include_signature=False)
    mf.write_to_filename(str(tmp_path / "standalone.csv"))
    sidx = load_file_as_index(str(tmp_path / "standalone.csv"))
    assert isinstance(sidx, StandaloneManifestIndex) and len(sidx) == 3 and _md5s(sidx.signatures()) == want
    only47 = sidx.select(picklist=CollectionManifest.create_manifest([(three[1], "x")]).to_picklist())
    assert len(only47) == 1 and _md5s(only47.signatures()) == [three[1].md5sum()]
    # a list of paths
    plist = tmp_path / "paths.txt"
    plist.write_text(f"{tmp_path / 'all.sig'}\n{zloc}\n")
    pidx = load_file_as_index(str(plist))
    assert isinstance(pidx, MultiIndex) and len(pidx) == 6      # the zip walk yields each distinct member once
    # errors
    with pytest.raises(ValueError, match="Error while reading signatures from"):
        load_file_as_index(str(tmp_path / "does-not-exist"))
    fa = tmp_path / "x.fa"
    fa.write_text(">r\nACGT\n")
    with pytest.raises(ValueError, match="got sequences instead"):
        load_file_as_index(str(fa))
    # the reference's own zip fixture (mixed molecule types, manifest inside)
    allzip = ZipFileLinearIndex.load(golden("zips", "all.zip"))
    assert len(allzip) == 8 and len(allzip.select(moltype="DNA")) == 2
    assert len(list(allzip.select(moltype="protein", ksize=19).signatures())) == 2
