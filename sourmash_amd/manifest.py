"""Manifests: one metadata row per sketch of a collection, for selection without loading sketches.

API of src/sourmash/manifest.py (BaseCollectionManifest :17-254, CollectionManifest
:257-387).  The CSV format is the reference's: a '# SOURMASH-MANIFEST-VERSION: 1.0'
line, then the columns of `required_keys`.  The bulk loader (csrc/collection.hpp) reads
and writes the same format natively; this class is the row-level view the Index
classes select on.
"""
import ast
import csv
import gzip
import itertools
import os

from . import picklist as _picklist

__all__ = ["BaseCollectionManifest", "CollectionManifest"]

_VERSION_PREFIX = "# SOURMASH-MANIFEST-VERSION: "


class BaseCollectionManifest:
    "Signature metadata for a collection; the container protocol answers `ss in manifest`."

    required_keys = ("internal_location", "md5", "md5short", "ksize", "moltype", "num", "scaled", "n_hashes",
                     "with_abundance", "name", "filename")

    # ---- reading ---------------------------------------------------------------------------------------------
    @classmethod
    def load_from_filename(cls, filename):
        opener = gzip.open if filename.endswith(".gz") else open
        with opener(filename, "rt", newline="") as fp:
            return cls.load_from_csv(fp)

    @classmethod
    def load_from_csv(cls, fp):
        first = fp.readline().rstrip()
        if not first.startswith(_VERSION_PREFIX):
            raise ValueError("manifest is missing version header")
        version = first[len(_VERSION_PREFIX):]
        if float(version) != 1.0:
            raise ValueError(f"unknown manifest version number {version}")
        r = csv.DictReader(fp)
        if not r.fieldnames:
            raise ValueError("missing column headers in manifest")
        for k in cls.required_keys:
            if k not in r.fieldnames:
                raise ValueError(f"missing column '{k}' in manifest.")
        rows = []
        for row in r:
            for k in ("num", "scaled", "ksize", "n_hashes"):
                row[k] = int(row[k])
            row["with_abundance"] = bool(ast.literal_eval(str(row["with_abundance"])))
            row["signature"] = None
            rows.append(row)
        return CollectionManifest(rows)

    # ---- writing ---------------------------------------------------------------------------------------------
    def write_to_filename(self, filename, *, database_format="csv", ok_if_exists=False):
        if database_format != "csv":
            raise NotImplementedError("only CSV manifests are supported (SQLite manifests are outside the hot path)")
        if os.path.exists(filename) and not ok_if_exists:
            raise Exception("output manifest already exists")
        opener = gzip.open if filename.endswith(".gz") else open
        with opener(filename, "wt", newline="") as fp:
            return self.write_to_csv(fp, write_header=True)

    @classmethod
    def write_csv_header(cls, fp):
        fp.write(_VERSION_PREFIX + "1.0\n")
        csv.DictWriter(fp, fieldnames=cls.required_keys).writeheader()

    def write_to_csv(self, fp, write_header=False):
        w = csv.DictWriter(fp, fieldnames=self.required_keys, extrasaction="ignore")
        if write_header:
            self.write_csv_header(fp)
        for row in self.rows:
            row.pop("signature", None)
            w.writerow(row)

    # ---- building ----------------------------------------------------------------------------------------------
    @classmethod
    def make_manifest_row(cls, ss, location, *, include_signature=True):
        mh = ss.minhash
        md5 = ss.md5sum()
        row = {"internal_location": location, "md5": md5, "md5short": md5[:8], "ksize": int(mh.ksize),
               "moltype": mh.moltype, "num": int(mh.num), "scaled": int(mh.scaled), "n_hashes": len(mh),
               "with_abundance": mh.track_abundance, "name": ss.name, "filename": ss.filename}
        if include_signature:
            row["signature"] = ss
        return row

    @classmethod
    def create_manifest(cls, locations_iter, *, include_signature=True):
        "from an iterator of (signature, location); load errors of the iterator pass through"
        return cls([cls.make_manifest_row(ss, loc, include_signature=include_signature) for ss, loc in locations_iter])

    def _check_row_values(self):
        from .index import _check_select_parameters
        for row in self.rows:
            _check_select_parameters(num=row["num"], ksize=row["ksize"], moltype=row["moltype"], scaled=row["scaled"],
                                     abund=row["with_abundance"])


class CollectionManifest(BaseCollectionManifest):
    "Rows in a list, md5s in a set."

    def __init__(self, rows=[]):
        self.rows = []
        self._md5_set = set()
        self._add_rows(rows)

    @classmethod
    def load_from_manifest(cls, manifest, **kwargs):
        return cls(manifest.rows)

    def add_row(self, row):
        self._add_rows([row])

    def _add_rows(self, rows):
        for row in rows:
            self.rows.append(row)
            self._md5_set.add(row["md5"])

    def __iadd__(self, other):
        if self is other:
            raise Exception("cannot directly add manifest to itself")
        self._add_rows(other.rows)
        return self

    def __add__(self, other):
        out = CollectionManifest(self.rows)
        out._add_rows(other.rows)
        return out

    def __bool__(self):
        return bool(self.rows)

    def __len__(self):
        return len(self.rows)

    def __eq__(self, other):
        "row by row, in order, on the required columns"
        for a, b in itertools.zip_longest(self.rows, other.rows):
            if a is None or b is None or any(a[k] != b[k] for k in self.required_keys):
                return False
        return True

    # ---- selection ---------------------------------------------------------------------------------------------
    def _select(self, *, ksize=None, moltype=None, scaled=0, num=0, containment=False, abund=None, picklist=None):
        from .index import _check_select_parameters
        _check_select_parameters(ksize=ksize, num=num, abund=abund, moltype=moltype, scaled=scaled)
        for row in self.rows:
            if ksize and row["ksize"] != ksize:
                continue
            if moltype and row["moltype"] != moltype:
                continue
            if (scaled or containment) and not (row["scaled"] and not row["num"]):
                continue
            if num and not (row["num"] and not row["scaled"]):
                continue
            if abund and not row["with_abundance"]:
                continue
            if picklist and not picklist.matches_manifest_row(row):
                continue
            yield row

    def select_to_manifest(self, **kwargs):
        return CollectionManifest(self._select(**kwargs))

    def filter_rows(self, row_filter_fn):
        return CollectionManifest([row for row in self.rows if row_filter_fn(row)])

    def filter_on_columns(self, col_filter_fn, col_names):
        return self.filter_rows(lambda row: col_filter_fn([row[c] for c in col_names if row[c] is not None]))

    def locations(self):
        "distinct internal locations, first-seen order"
        seen = set()
        for row in self.rows:
            loc = row["internal_location"]
            if loc not in seen:
                seen.add(loc)
                yield loc

    def __contains__(self, ss):
        return ss.md5sum() in self._md5_set

    def to_picklist(self):
        pl = _picklist.SignaturePicklist("manifest")
        pl.pickset = {pl._get_value_for_manifest_row(row) for row in self.rows}
        return pl
