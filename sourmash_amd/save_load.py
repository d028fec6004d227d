"""Loading any flat collection as an Index, and saving signatures to a location.

API of src/sourmash/save_load.py (load_file_as_index :42-58, SaveSignaturesToLocation
:61-83, the SaveSignatures_* classes :218-549) and of sourmash_args.load_file_as_signatures
(:765-815), for the collection formats on the hot path: JSON signature files (.sig,
.sig.gz), directories of them, zip files with or without a manifest, standalone manifests
and path lists.  SBT, LCA and SQLite databases are not loaded (out of scope).
For bulk loading straight into HBM, without signature objects, see `SketchSet.load`.
"""
import gzip
import io
import os
import sys
import zipfile

from . import signature as sigmod
from .exceptions import SourmashError
from .index import LinearIndex, MultiIndex, StandaloneManifestIndex, ZipFileLinearIndex, ZipStorage
from .manifest import CollectionManifest

__all__ = ["load_file_as_index", "load_file_as_signatures", "SaveSignaturesToLocation", "SaveSignatures_NoOutput",
           "SaveSignatures_Directory", "SaveSignatures_SigFile", "SaveSignatures_ZipFile"]


def _looks_like_sequences(filename):
    try:
        with open(filename, "rb") as fh:
            head = fh.read(2)
        if head == b"\x1f\x8b":
            with gzip.open(filename, "rb") as fh:
                head = fh.read(1)
        return head[:1] in (b">", b"@")
    except OSError:
        return False


def load_file_as_index(filename, *, yield_all_files=False):
    """An Index for `filename`.  Tried in the reference's order: standalone manifest, path (JSON file or directory),
    list of paths, zip file; sequence files are reported as such."""
    filename = os.fspath(filename)
    if filename == "-":
        lidx = LinearIndex.load(sys.stdin, filename="-")
        return MultiIndex.load((lidx,), (None,), parent="-")
    loaders = (
        lambda: StandaloneManifestIndex.load(filename),
        lambda: MultiIndex.load_from_path(filename, yield_all_files),
        lambda: MultiIndex.load_from_pathlist(filename),
        lambda: ZipFileLinearIndex.load(filename, traverse_yield_all=yield_all_files) if filename.endswith(".zip") else None,
    )
    for attempt in loaders:
        try:
            db = attempt()
        except (ValueError, SourmashError, OSError, UnicodeDecodeError, zipfile.BadZipFile, gzip.BadGzipFile, EOFError):
            db = None
        if db is not None:
            return db
    if _looks_like_sequences(filename):
        raise ValueError(f"Error while reading signatures from '{filename}' - got sequences instead! "
                         "Is this a FASTA/FASTQ file?")
    raise ValueError(f"Error while reading signatures from '{filename}'.")


def load_file_as_signatures(filename, *, select_moltype=None, ksize=None, picklist=None, yield_all_files=False,
                            progress=None, pattern=None, _use_manifest=True):
    "The signatures of `filename` (any format load_file_as_index reads), filtered by molecule type, ksize, picklist."
    if progress:
        progress.notify(filename)
    db = load_file_as_index(filename, yield_all_files=yield_all_files)
    if not _use_manifest and db.manifest:
        db.manifest = None
    db = db.select(moltype=select_moltype, ksize=ksize)
    if picklist is not None:
        db = db.select(picklist=picklist)
    if pattern is not None:
        manifest = db.manifest
        if manifest is None:
            raise ValueError("pattern matching needs a collection with a manifest")
        db = db.select(picklist=manifest.filter_on_columns(pattern, ["name", "filename", "md5"]).to_picklist())
    loader = db.signatures()
    return progress.start_file(filename, loader) if progress is not None else loader


# ---- saving ------------------------------------------------------------------------------------------------------
class Base_SaveSignaturesToLocation:
    "Context manager that counts what it is given; subclasses decide where it goes."

    def __init__(self, location):
        self.location = location
        self.count = 0

    @classmethod
    def matches(cls, location):
        raise NotImplementedError

    def __len__(self):
        return self.count

    def open(self):
        pass

    def close(self):
        pass

    def __enter__(self):
        self.open()
        return self

    def __exit__(self, type, value, traceback):
        self.close()

    def add(self, ss):
        self.count += 1

    def add_many(self, sslist):
        for ss in sslist:
            self.add(ss)


def _one_sketch_each(siglist):
    "signatures holding several sketches come back as one signature per sketch (round trip through JSON)"
    yield from sigmod.load_signatures_from_json(sigmod.save_signatures_to_json(siglist))


_get_signatures_from_rust = _one_sketch_each               # the reference's name for it (save_load.py:250-259)


class SaveSignatures_NoOutput(Base_SaveSignaturesToLocation):
    def __repr__(self):
        return "SaveSignatures_NoOutput()"

    @classmethod
    def matches(cls, location):
        return location is None


class SaveSignatures_Directory(Base_SaveSignaturesToLocation):
    "One <md5>.sig.gz per signature under a directory (a location ending in '/')."

    def __repr__(self):
        return f"SaveSignatures_Directory('{self.location}')"

    @classmethod
    def matches(cls, location):
        return bool(location) and location.endswith("/")

    def open(self):
        os.makedirs(self.location, exist_ok=True)

    def add(self, ss):
        super().add(ss)
        md5 = ss.md5sum()
        outname = os.path.join(self.location, f"{md5}.sig.gz")
        i = 0
        while os.path.exists(outname):
            outname = os.path.join(self.location, f"{md5}_{i}.sig.gz")
            i += 1
        with open(outname, "wb") as fp:
            sigmod.save_signatures_to_json([ss], fp, compression=1)


class SaveSignatures_SigFile(Base_SaveSignaturesToLocation):
    "Everything into one JSON file, written at close ('-' is stdout; '.gz' compresses)."

    def __init__(self, location):
        super().__init__(location)
        self.keep = []
        self.compress = 1 if self.location.endswith(".gz") else 0

    @classmethod
    def matches(cls, location):
        return bool(location)

    def __repr__(self):
        return f"SaveSignatures_SigFile('{self.location}')"

    def close(self):
        if self.location == "-":
            sigmod.save_signatures_to_json(self.keep, sys.stdout)
            return
        with open(self.location, "wb") as fp:
            sigmod.save_signatures_to_json(self.keep, fp, compression=self.compress)

    def add(self, ss):
        super().add(ss)
        self.keep.append(ss)


class SaveSignatures_ZipFile(Base_SaveSignaturesToLocation):
    """signatures/<md5>.sig.gz members, stored, plus SOURMASH-MANIFEST.csv (deflated) written at close.  Adding to an
    existing zip file requires that it already has a manifest."""

    def __init__(self, location):
        super().__init__(location)
        self.storage = None

    @classmethod
    def matches(cls, location):
        return bool(location) and location.endswith(".zip")

    def __repr__(self):
        return f"SaveSignatures_ZipFile('{self.location}')"

    def open(self):
        existed = os.path.exists(self.location)
        try:
            storage = ZipStorage(self.location, mode="w")
        except zipfile.BadZipFile:
            raise ValueError(f"File '{self.location}' cannot be opened as a zip file.")
        if not storage.subdir:
            storage.subdir = "signatures"
        try:
            data = storage.load("SOURMASH-MANIFEST.csv")
        except (FileNotFoundError, KeyError):
            if existed:
                raise ValueError(f"Cannot add to existing zipfile '{self.location}' without a manifest")
            self.manifest_rows = []
        else:
            manifest = CollectionManifest.load_from_csv(io.StringIO(data.decode("utf-8"), newline=""))
            self.manifest_rows = list(manifest._select())
        self.storage = storage

    def close(self):
        fp = io.StringIO(newline="")
        CollectionManifest(self.manifest_rows).write_to_csv(fp, write_header=True)
        self.storage.save("SOURMASH-MANIFEST.csv", fp.getvalue().encode("utf-8"), overwrite=True, compress=True)
        self.storage.flush()
        self.storage.close()

    def add(self, add_sig):
        if not self.storage:
            raise ValueError("this output is not open")
        for ss in _one_sketch_each([add_sig]):
            buf = sigmod.save_signatures_to_json([ss], compression=1)
            location = self.storage.save(f"{self.storage.subdir}/{ss.md5sum()}.sig.gz", buf)
            self.manifest_rows.append(CollectionManifest.make_manifest_row(ss, location, include_signature=False))
            super().add(ss)


_save_classes = [(10, SaveSignatures_NoOutput), (20, SaveSignatures_Directory), (30, SaveSignatures_ZipFile),
                 (1000, SaveSignatures_SigFile)]


def SaveSignaturesToLocation(location):
    "`with SaveSignaturesToLocation(path) as save: save.add(sig)` -- the class is chosen by the shape of the path"
    if location is not None and location.endswith(".sqldb"):
        raise NotImplementedError("SQLite output is outside the hot path")
    for _priority, cls in sorted(_save_classes, key=lambda x: x[0]):
        if cls.matches(location):
            return cls(location)
    raise Exception(f"cannot determine how to open location {location} for saving; this should never happen!?")
