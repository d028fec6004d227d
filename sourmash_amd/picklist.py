"""Picklists: include/exclude sets that Index.select and manifests filter signatures with.

API of src/sourmash/picklist.py (SignaturePicklist :71-337, PickStyle :66-68,
passes_all_picklists :340-345).  A picklist holds a set of values of one column type
(a name, an md5, an identifier, or the (identifier, md5 prefix) pair that manifest /
gather / prefetch / search CSVs carry) and answers "is this signature / manifest row in it".
"""
import csv
import gzip
import os
from enum import Enum

__all__ = ["PickStyle", "SignaturePicklist", "passes_all_picklists"]


class PickStyle(Enum):
    INCLUDE = 1
    EXCLUDE = 2


def _ident(name):
    return name.split(" ")[0]


def _ident_md5(pair):
    name, md5 = pair
    return (_ident(name), md5[:8])


# value as found (signature attribute, manifest column or CSV cell) -> value as stored in the pick set
_NORMALISE = {
    "name": lambda x: x,
    "md5": lambda x: x,
    "md5prefix8": lambda x: x[:8],
    "md5short": lambda x: x[:8],
    "ident": _ident,
    "identprefix": lambda x: _ident(x).split(".")[0],
    "manifest": _ident_md5,
    "prefetch": _ident_md5,
    "gather": _ident_md5,
    "search": _ident_md5,
}


class SignaturePicklist:
    """Subset collections of signatures by 'pickfile:column:coltype[:include|exclude]'."""

    meta_coltypes = ("manifest", "gather", "prefetch", "search")
    supported_coltypes = ("md5", "md5prefix8", "md5short", "name", "ident", "identprefix")

    def __init__(self, coltype, *, pickfile=None, column_name=None, pickstyle=PickStyle.INCLUDE):
        if coltype not in self.meta_coltypes and coltype not in self.supported_coltypes:
            raise ValueError(f"invalid picklist column type '{coltype}'")
        self.orig_coltype = coltype
        self.orig_colname = column_name
        if coltype in self.meta_coltypes:
            if column_name:
                raise ValueError(f"no column name allowed for coltype '{coltype}'")
            column_name = "(match_name, match_md5)" if coltype == "prefetch" else "(name, md5)"
        self.coltype = coltype
        self.pickfile = pickfile
        self.column_name = column_name
        self.pickstyle = pickstyle
        self.preprocess_fn = _NORMALISE[coltype]
        self.pickset = None
        self.found = set()
        self.n_queries = 0

    @classmethod
    def from_picklist_args(cls, argstr):
        "'pickfile:col:coltype[:style]' -> picklist (not yet loaded)"
        parts = argstr.split(":")
        pickstyle = PickStyle.INCLUDE
        if len(parts) == 4:
            style = parts.pop()
            if style == "include":
                pickstyle = PickStyle.INCLUDE
            elif style == "exclude":
                pickstyle = PickStyle.EXCLUDE
            else:
                raise ValueError(f"invalid picklist 'pickstyle' argument 4: '{style}' must be 'include' or 'exclude'")
        if len(parts) != 3:
            raise ValueError(f"invalid picklist argument '{argstr}'")
        pickfile, column, coltype = parts
        return cls(coltype, pickfile=pickfile, column_name=column, pickstyle=pickstyle)

    # ---- the value a signature / manifest row / CSV row is looked up by -------------------------------------
    def _get_sig_attribute(self, ss):
        if self.coltype in self.meta_coltypes:
            return (ss.name, ss.md5sum())
        if self.coltype in ("md5", "md5prefix8", "md5short"):
            return ss.md5sum()
        return ss.name

    def _get_value_for_manifest_row(self, row):
        if self.coltype in self.meta_coltypes:
            q = (row["name"], row["md5"])
        elif self.coltype == "md5":
            q = row.get("md5")
        elif self.coltype in ("md5prefix8", "md5short"):
            q = row.get("md5short")
        else:
            q = row.get("name")
        assert q
        return self.preprocess_fn(q)

    def _get_value_for_csv_row(self, row):
        if self.coltype == "prefetch":
            q = (row["match_name"], row["match_md5"])
        elif self.coltype in self.meta_coltypes:
            q = (row["name"], row["md5"])
        else:
            q = row[self.column_name]
        return self.preprocess_fn(q) if q else q

    # ---- contents ------------------------------------------------------------------------------------------------
    def init(self, values=[]):
        if self.pickset is not None:
            raise ValueError("already initialized?")
        self.pickset = set(values)
        return self.pickset

    def add(self, value):
        self.pickset.add(value)

    def load(self, *, allow_empty=False):
        "read the pick file; returns (number of empty values, set of duplicated values)"
        pickset = self.init()
        pickfile = self.pickfile
        if not os.path.exists(pickfile) or not os.path.isfile(pickfile):
            raise ValueError(f"pickfile '{pickfile}' must exist and be a regular file")
        with open(pickfile, "rb") as probe:
            gz = probe.read(2) == b"\x1f\x8b"
        n_empty, dups = 0, set()
        with (gzip.open if gz else open)(pickfile, "rt", newline="") as fp:
            first = fp.readline()
            if not first.startswith("# SOURMASH-MANIFEST-VERSION"):     # manifests carry a version line first
                fp.seek(0)
            r = csv.DictReader(fp)
            if not r.fieldnames:
                if not allow_empty:
                    raise ValueError(f"empty or improperly formatted pickfile '{pickfile}'")
                return 0, 0
            if not (self.column_name in r.fieldnames or self.coltype in self.meta_coltypes):
                raise ValueError(f"column '{self.column_name}' not in pickfile '{pickfile}'")
            for row in r:
                col = self._get_value_for_csv_row(row)
                if not col:
                    n_empty += 1
                elif col in pickset:
                    dups.add(col)
                else:
                    self.add(col)
        return n_empty, dups

    # ---- matching ------------------------------------------------------------------------------------------------
    def _match(self, q):
        self.n_queries += 1
        hit = (q in self.pickset) if self.pickstyle == PickStyle.INCLUDE else (q not in self.pickset)
        if hit:
            self.found.add(q)
        return hit

    def __contains__(self, ss):
        return self._match(self.preprocess_fn(self._get_sig_attribute(ss)))

    def matches_manifest_row(self, row):
        return self._match(self._get_value_for_manifest_row(row))

    def matched_csv_row(self, row):
        "was this row of the original pick file ever matched?"
        q = self._get_value_for_csv_row(row)
        self.n_queries += 1
        return q in self.found

    def filter(self, it):
        for ss in it:
            if ss in self:
                yield ss


def passes_all_picklists(ss, picklists):
    return all(ss in pl for pl in picklists)
