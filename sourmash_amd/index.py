"""Index / CounterGather -- search, prefetch and gather over collections of signatures.

API of src/sourmash/index/__init__.py: IndexSearchResult (:55), Index.find (:115-170),
search (:202-239), prefetch (:241-256), best_containment (:258-270), counter_gather
(:302-320), LinearIndex (:397-453) and CounterGather (:735-909).

What changes underneath: the collection lives in HBM as one CSR (`SketchSet`), so
  * find() scores the query against EVERY signature with one overlap kernel
    (the reference loops over signatures, two FFI clones + one merge each);
  * CounterGather keeps its counters on the GPU: add-time overlaps, the arg-max with
    the reference tie-break, and consume() are single kernel launches.
Results (which signatures, which order, which numbers) are those of the reference.
"""
import csv
import ctypes as C
import io
import math
import os
import zipfile
from collections import Counter, namedtuple

import numpy as np

from ._lowlevel import lib
from .minhash import flatten_and_downsample_num, flatten_and_downsample_scaled, flatten_and_intersect_scaled
from .search import calc_threshold_from_bp, make_containment_query, make_jaccard_search_query
from .signature import SourmashSignature, load_signatures_from_json, save_signatures_to_json
from .utils import RustObject, decode_str, rustcall
from .manifest import CollectionManifest

__all__ = ["IndexSearchResult", "Collection", "SketchSet", "select_signature", "Index", "LinearIndex", "LazyLinearIndex",
           "ZipStorage", "ZipFileLinearIndex", "MultiIndex", "StandaloneManifestIndex", "CounterGather"]

IndexSearchResult = namedtuple("Result", "score, signature, location")


def _path_array(paths):
    if isinstance(paths, (str, bytes)) or hasattr(paths, "__fspath__"):
        paths = [paths]
    enc = [os.fsencode(p) for p in paths]
    return (C.c_char_p * max(len(enc), 1))(*enc), len(enc)


def _manifest_rows(csv_text):
    "manifest CSV -> list of dict rows with the reference's column types (manifest.py:84-98)"
    lines = csv_text.splitlines(keepends=True)
    rows = list(csv.DictReader(io.StringIO("".join(lines[1:]), newline="")))
    for row in rows:
        for k in ("num", "scaled", "ksize", "n_hashes"):
            row[k] = int(row[k])
        row["with_abundance"] = bool(int(row["with_abundance"]))
    return rows


class Collection(RustObject):
    """Signature files parsed into one host CSR + manifest by the native multi-threaded loader
    (smgpu_collection_*; no GPU involved).  `to_device()` uploads it as a SketchSet."""
    __dealloc_func__ = lib.smgpu_collection_free

    def __init__(self, paths, *, ksize=0, moltype=None, scaled=0, threads=0):
        arr, n = _path_array(paths)
        self._objptr = rustcall(lib.smgpu_collection_load, arr, n, int(ksize or 0),
                                moltype.encode() if moltype else None, int(scaled or 0), int(threads))

    def __len__(self):
        return self._methodcall(lib.smgpu_collection_len)

    @property
    def total_hashes(self):
        return self._methodcall(lib.smgpu_collection_total_hashes)

    @property
    def skipped(self):
        "sketches seen in the inputs that the selection left out"
        return self._methodcall(lib.smgpu_collection_skipped)

    @property
    def manifest_csv(self):
        return decode_str(self._methodcall(lib.smgpu_collection_manifest))

    @property
    def manifest(self):
        return _manifest_rows(self.manifest_csv)

    @property
    def offsets(self):
        ptr = self._methodcall(lib.smgpu_collection_offsets)
        return np.ctypeslib.as_array(ptr, shape=(len(self) + 1,)).copy()

    @property
    def hashes(self):
        n = self.total_hashes
        if not n:
            return np.zeros(0, dtype=np.uint64)
        return np.ctypeslib.as_array(self._methodcall(lib.smgpu_collection_hashes), shape=(n,)).copy()

    def to_device(self):
        return SketchSet._from_objptr(self._methodcall(lib.smgpu_sketchset_from_collection))


class SketchSet(RustObject):
    """n flat sketches packed as one device-resident CSR (smgpu_sketchset_*).

    Built from MinHash objects (`SketchSet(minhashes)`) or straight from files (`SketchSet.load(paths, ...)`:
    .sig / .sig.gz / .zip / directories / path lists, parsed by the native loader without creating a Python
    object per sketch).  A loaded set answers compare / search / gather by row number; `manifest[row]` says
    which signature that is, `signature(row)` materialises it."""
    __dealloc_func__ = lib.smgpu_sketchset_free
    _keep = ()
    _manifest = None

    def __init__(self, minhashes):
        self._keep = list(minhashes)
        ptrs = (C.c_void_p * max(len(self._keep), 1))(*[mh._get_objptr() for mh in self._keep])
        self._objptr = rustcall(lib.smgpu_sketchset_new, ptrs, len(self._keep))

    @classmethod
    def load(cls, paths, *, ksize=0, moltype=None, scaled=0, threads=0):
        arr, n = _path_array(paths)
        return cls._from_objptr(rustcall(lib.smgpu_sketchset_load, arr, n, int(ksize or 0),
                                         moltype.encode() if moltype else None, int(scaled or 0), int(threads)))

    def __len__(self):
        return self._methodcall(lib.smgpu_sketchset_len)

    @property
    def total_hashes(self):
        return self._methodcall(lib.smgpu_sketchset_total_hashes)

    @property
    def manifest(self):
        if self._manifest is None:
            self._manifest = _manifest_rows(decode_str(self._methodcall(lib.smgpu_sketchset_manifest)))
        return self._manifest

    @property
    def sizes(self):
        out = np.zeros(max(len(self), 1), dtype=np.uint64)
        self._methodcall(lib.smgpu_sketchset_sizes, out.ctypes.data_as(C.c_void_p))
        return out[:len(self)]

    @property
    def params(self):
        "(ksize, moltype, seed, scaled, num) shared by every row of a loaded set"
        k, hf, seed, mx, num = C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._methodcall(lib.smgpu_sketchset_params, C.byref(k), C.byref(hf), C.byref(seed), C.byref(mx), C.byref(num))
        from .minhash import _get_scaled_for_max_hash
        moltype = {1: "DNA", 2: "protein", 3: "dayhoff", 4: "hp"}[hf.value]
        return k.value, moltype, seed.value, _get_scaled_for_max_hash(mx.value) if mx.value else 0, num.value

    def minhash(self, row):
        from .minhash import MinHash
        return MinHash._from_objptr(self._methodcall(lib.smgpu_sketchset_get, int(row)))

    def subset(self, rows):
        "the given rows of a loaded set as a new set (device-side row gather; the manifest rows follow)"
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        return SketchSet._from_objptr(self._methodcall(lib.smgpu_sketchset_subset, rows.ctypes.data_as(C.c_void_p), len(rows)))

    def signature(self, row):
        "SourmashSignature of a row of a loaded set (name / filename from the manifest)"
        m = self.manifest[row]
        return SourmashSignature(self.minhash(row), name=m["name"], filename=m["filename"])

    def compare(self, *, jaccard=True):
        "-> (common u32 [n, n], jaccard f64 [n, n] or None) for the whole set"
        n = len(self)
        common = np.zeros((n, n), dtype=np.uint32)
        jac = np.zeros((n, n), dtype=np.float64) if jaccard else None
        self._methodcall(lib.smgpu_sketchset_compare, common.ctypes.data_as(C.c_void_p),
                         jac.ctypes.data_as(C.c_void_p) if jaccard else None)
        return common, jac

    def overlaps(self, query_mh):
        "|query ∩ row| for every row (one streaming pass over the CSR)"
        out = np.zeros(max(len(self), 1), dtype=np.uint64)
        self._methodcall(lib.smgpu_sketchset_overlaps, query_mh.flatten()._get_objptr(), out.ctypes.data_as(C.c_void_p))
        return out[:len(self)]

    def search(self, query_mh, *, threshold=0.0, do_containment=False, do_max_containment=False, best_only=False):
        """Rows scoring >= threshold against the query, best first -> [(score, row)].  Jaccard by default, query
        containment or max containment on request: the scores of JaccardSearch (search.py:88-160) from one overlap
        pass over the whole set."""
        if do_containment and do_max_containment:
            raise TypeError("'do_containment' and 'do_max_containment' cannot both be True")
        if (do_containment or do_max_containment) and not query_mh.scaled:
            raise TypeError("this search requires a scaled signature")
        shared = self.overlaps(query_mh).astype(np.float64)
        sizes = self.sizes.astype(np.float64)
        nq = float(len(query_mh))
        if do_containment:
            score = shared / nq if nq else np.zeros_like(shared)
        elif do_max_containment:
            score = np.divide(shared, np.minimum(sizes, nq), out=np.zeros_like(shared), where=np.minimum(sizes, nq) > 0)
        else:
            union = sizes + nq - shared
            score = np.divide(shared, union, out=np.zeros_like(shared), where=union > 0)
        keep = np.flatnonzero((score >= threshold) & (score > 0))
        order = keep[np.argsort(-score[keep], kind="stable")]
        hits = [(float(score[r]), int(r)) for r in order]
        return hits[:1] if best_only else hits

    def prefetch(self, query_mh, threshold_bp=0):
        "Rows sharing at least threshold_bp with the query -> [(row, |intersect|)] in row order (search.py:956-976)."
        scaled = query_mh.scaled
        if not scaled:
            raise ValueError("prefetch requires scaled signatures")
        shared = self.overlaps(query_mh)
        need = float(threshold_bp) / scaled if threshold_bp else 0.0
        return [(int(r), int(shared[r])) for r in np.flatnonzero((shared >= need) & (shared > 0))]

    def gather(self, query_mh, threshold_bp=0):
        """Min-set-cover of the query by the rows of this set -> [(row, |intersect|)] in rank order; the whole
        loop runs on the GPU (GatherDatabases semantics, search.py:877-949, for equal scaled)."""
        scaled = query_mh.scaled
        if not scaled:
            raise ValueError("gather requires scaled signatures")
        thr = math.ceil(float(threshold_bp) / scaled) if threshold_bp else 0
        idx, isect = _DeviceCounter(self, query_mh.flatten()).gather(thr)
        return [(int(i), int(c)) for i, c in zip(idx, isect)]


class _DeviceCounter(RustObject):
    "c[d] = |query ∩ D_d| on the GPU (smgpu_counter_*)."
    __dealloc_func__ = lib.smgpu_counter_free

    def __init__(self, sketchset, query_mh):
        self._set = sketchset
        self._objptr = rustcall(lib.smgpu_counter_new, sketchset._get_objptr(), query_mh._get_objptr())

    def values(self):
        out = np.zeros(max(len(self._set), 1), dtype=np.uint64)
        self._methodcall(lib.smgpu_counter_get, out.ctypes.data_as(C.c_void_p))
        return out[:len(self._set)]

    def best(self):
        "-> (index, count) of the largest counter (ties: lowest index) or None"
        idx, cnt = C.c_uint64(0), C.c_uint64(0)
        ok = self._methodcall(lib.smgpu_counter_best, C.byref(idx), C.byref(cnt))
        return (idx.value, cnt.value) if ok else None

    def consume(self, intersect_mh):
        self._methodcall(lib.smgpu_counter_consume, intersect_mh._get_objptr())

    def set(self, index, value):
        self._methodcall(lib.smgpu_counter_set, index, value)

    def gather(self, threshold_hashes=0):
        "Every remaining round of the min-set-cover loop on the GPU -> (winner indices, |intersect| per round)."
        n = max(len(self._set), 1)
        idx, isect = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        k = self._methodcall(lib.smgpu_counter_gather, int(threshold_hashes), idx.ctypes.data_as(C.c_void_p),
                             isect.ctypes.data_as(C.c_void_p), n)
        return idx[:k], isect[:k]


def _check_select_parameters(**kw):
    "types of the 'select' arguments (index/__init__.py:1229-1270)"
    extra = set(kw) - {"ksize", "num", "moltype", "scaled", "abund", "picklist", "containment"}
    if extra:
        raise ValueError(f"unknown 'select' parameters: {extra}")
    for key in ("ksize", "scaled", "num"):
        v = kw.get(key)
        if v is not None and not isinstance(v, int):
            raise ValueError(f"{key} value '{v}' must be an integer, is: {type(v)}")
    moltype = kw.get("moltype")
    if moltype is not None and moltype not in ("DNA", "protein", "dayhoff", "hp"):
        raise ValueError(f"unknown moltype: {moltype}")
    for key in ("containment", "abund"):
        v = kw.get(key)
        if v is not None and not isinstance(v, bool):
            raise ValueError(f"{key} value '{v}' must be a bool, is: {type(v)}")


def select_signature(ss, *, ksize=None, moltype=None, scaled=0, num=0, containment=False, abund=None, picklist=None):
    "does the signature meet the requirements? (index/__init__.py:349-394)"
    mh = ss.minhash
    if ksize and ksize != mh.ksize:
        return False
    if moltype and moltype != mh.moltype:
        return False
    if containment:                              # containment needs scaled sketches; similarity does not
        if not scaled:
            raise ValueError("'containment' requires 'scaled' in Index.select'")
        if not mh.scaled:
            return False
    if scaled and mh.num:                        # 'scaled' and 'num' exclude each other
        return False
    if num and (mh.scaled or num != mh.num):
        return False
    if abund and not mh.track_abundance:         # a sketch with abundances can always be flattened
        return False
    if picklist is not None and ss not in picklist:
        return False
    return True


def _zero_overlap_never_matches(search_fn, q_size):
    "true for the Jaccard / containment searches of search.py: no shared hash -> score 0 -> never passes"
    try:
        return search_fn.score_fn(q_size, 0, 1, q_size + 1) == 0 and not search_fn.passes(0)
    except Exception:                                        # noqa: BLE001  (an unusual search object: score every row)
        return False


class Index:
    """Base of every collection of signatures: selection, the signature walk, and search / prefetch / gather on top
    of one `find` (index/__init__.py:58-346).  `find` here scores the query against ALL signatures of the walk with one
    overlap kernel when they are scaled sketches; results and their order are those of the reference's per-signature loop."""
    is_database = False
    manifest = None                  # set by classes that select through a manifest

    def __len__(self):
        raise NotImplementedError

    @property
    def location(self):
        return None

    def signatures(self):
        raise NotImplementedError

    def signatures_with_location(self):
        for ss in self.signatures():
            yield ss, self.location

    def _signatures_with_internal(self):
        "(signature, internal location) for ALL signatures, selection ignored (used to build manifests)"
        raise NotImplementedError

    def insert(self, signature):
        raise NotImplementedError

    def save(self, path):
        raise NotImplementedError

    @classmethod
    def load(cls, location):
        raise NotImplementedError

    def select(self, **kwargs):
        raise NotImplementedError

    # ---- batched scoring --------------------------------------------------------------------------
    def _walk(self):
        """The signatures of the walk with their locations and sketches, and the exception that cut it short (or
        None).  The reference touches one signature at a time, so a signature that cannot be produced or read fails
        only when the walk reaches it: everything in front of it is still scored and yielded first."""
        items, pending = [], None
        it = iter(self.signatures_with_location())
        while True:
            try:
                ss, loc = next(it)
                mh = ss.minhash
            except StopIteration:
                break
            except Exception as exc:                         # noqa: BLE001  (re-raised by find at this position)
                pending = exc
                break
            items.append((ss, loc, mh))
        return items, pending

    def _subject_set(self, query_scaled, items):
        "subject sketches flattened and downsampled to the query's scaled when finer (find :125-131), and their CSR"
        subj = [flatten_and_downsample_scaled(mh, query_scaled) for _, _, mh in items]
        return (SketchSet(subj) if subj else None), subj

    def find(self, search_fn, query, **kwargs):
        search_fn.check_is_compatible(query)
        query_mh = query.minhash
        assert not query_mh.track_abundance
        items, pending = self._walk()
        if query_mh.scaled and all(mh.scaled for _, _, mh in items):
            sset, subj = self._subject_set(query_mh.scaled, items)
            shared = sset.overlaps(query_mh) if sset is not None else []
            skip_zero = _zero_overlap_never_matches(search_fn, len(query_mh))
            for i, ((ss, loc, _), subj_mh) in enumerate(zip(items, subj)):
                if skip_zero and not shared[i]:
                    continue
                # the query is downsampled to the subject's scaled when the subject is coarser (:129-131)
                q_mh = query_mh if subj_mh.scaled <= query_mh.scaled else flatten_and_downsample_scaled(query_mh, subj_mh.scaled)
                q_size, s_size = len(q_mh), len(subj_mh)
                # plain |Q ∩ D| is unchanged by downsampling either side to the coarser scaled
                n_shared = int(shared[i])
                total = q_size + s_size - n_shared
                score = search_fn.score_fn(q_size, n_shared, s_size, total)
                if search_fn.passes(score) and search_fn.collect(score, ss):
                    yield IndexSearchResult(score, ss, loc)
        else:
            # num sketches (or mixed): per-pair GPU intersections, like the reference loop
            for ss, loc, mh in items:
                if query_mh.scaled:
                    subj_mh = flatten_and_downsample_scaled(mh, query_mh.scaled)
                    q_mh = flatten_and_downsample_scaled(query_mh, subj_mh.scaled)
                else:
                    subj_mh = flatten_and_downsample_num(mh, query_mh.num)
                    q_mh = flatten_and_downsample_num(query_mh, subj_mh.num)
                n_shared, total = q_mh.intersection_and_union_size(subj_mh)
                score = search_fn.score_fn(len(q_mh), n_shared, len(subj_mh), total)
                if search_fn.passes(score) and search_fn.collect(score, ss):
                    yield IndexSearchResult(score, ss, loc)
        if pending is not None:
            raise pending

    def search(self, query, *, threshold=None, do_containment=False, do_max_containment=False, best_only=False, **kw):
        if threshold is None:
            raise TypeError("'search' requires 'threshold'")
        search_obj = make_jaccard_search_query(do_containment=do_containment, do_max_containment=do_max_containment,
                                               best_only=best_only, threshold=float(threshold))
        matches = list(self.find(search_obj, query, **kw))
        matches.sort(key=lambda x: -x.score)
        return matches

    def search_abund(self, query, *, threshold=None, **kw):
        if not query.minhash.track_abundance:
            raise TypeError("'search_abund' requires query signature with abundance information")
        if threshold is None:
            raise TypeError("'search_abund' requires 'threshold'")
        out = []
        for ss, loc in self.signatures_with_location():
            if not ss.minhash.track_abundance:
                raise TypeError("'search_abund' requires subject signatures with abundance information")
            score = query.similarity(ss, downsample=True)
            if score >= float(threshold):
                out.append(IndexSearchResult(score, ss, loc))
        out.sort(key=lambda x: -x.score)
        return out

    def prefetch(self, query, threshold_bp, **kwargs):
        if not self:
            raise ValueError("no signatures to search")
        search_fn = make_containment_query(query.minhash, threshold_bp, best_only=kwargs.get("best_only", False))
        yield from self.find(search_fn, query, **kwargs)

    def best_containment(self, query, threshold_bp=None, **kwargs):
        results = sorted(self.prefetch(query, threshold_bp, best_only=True, **kwargs),
                         key=lambda x: (-x.score, x.signature.md5sum()))
        return results[0] if results else None

    def peek(self, query_mh, *, threshold_bp=0):
        try:
            result = self.best_containment(SourmashSignature(query_mh), threshold_bp=threshold_bp)
        except ValueError:
            result = None
        if not result:
            return []
        return [result, flatten_and_intersect_scaled(result.signature.minhash, query_mh)]

    def consume(self, intersect_mh):
        pass

    def counter_gather(self, query, threshold_bp, **kwargs):
        "Prefetch, then hand every match to a CounterGather in ONE batched add (index/__init__.py:302-320)."
        prefetch_query = query.to_mutable()
        prefetch_query.minhash = prefetch_query.minhash.flatten()
        counter = CounterGather(prefetch_query)
        matches = list(self.prefetch(prefetch_query, threshold_bp, **kwargs))
        counter.add_many([m.signature for m in matches], locations=[m.location for m in matches])
        return counter

    def gather(self, query, threshold_bp=None, **kwargs):
        "Best containment match only (index/__init__.py:272-300)."
        if not query.minhash.scaled:
            raise ValueError("gather requires scaled signatures")
        res = self.best_containment(query, threshold_bp=threshold_bp, **kwargs)
        return [res] if res else []


class LinearIndex(Index):
    "An in-memory list of signatures searched exhaustively -- on the GPU, all at once (index/__init__.py:397-453)."

    def __init__(self, _signatures=None, filename=None):
        self._signatures = list(_signatures) if _signatures else []
        self.filename = filename
        self._packed = None          # (query scaled, number of subjects) -> (SketchSet, prepared minhashes)

    @property
    def location(self):
        return self.filename

    def signatures(self):
        yield from self._signatures

    def __bool__(self):
        return bool(self._signatures)

    def __len__(self):
        return len(self._signatures)

    def insert(self, node):
        self._signatures.append(node)
        self._packed = None

    def save(self, path):
        "All signatures as one JSON file (index/__init__.py:427-429)."
        with open(path, "wb") as fp:
            save_signatures_to_json(self.signatures(), fp)

    @classmethod
    def load(cls, location, filename=None):
        "From a JSON signature file; raises when it cannot be parsed (index/__init__.py:431-439)."
        si = load_signatures_from_json(location, do_raise=True)
        return cls(si, filename=location if filename is None else filename)

    def select(self, **kwargs):
        """New LinearIndex with the signatures that match the requirements; never raises for 'nothing matches'
        (index/__init__.py:441-453 over select_signature :349-394, parameters checked as in :1229-1270)."""
        _check_select_parameters(**kwargs)
        return LinearIndex([ss for ss in self._signatures if select_signature(ss, **kwargs)], self.filename)

    def _subject_set(self, query_scaled, items):
        "the list only changes through insert(): keep the device CSR between queries of the same scaled"
        key = (query_scaled, len(items))
        if self._packed is None or self._packed[0] != key:
            self._packed = (key,) + Index._subject_set(self, query_scaled, items)
        return self._packed[1], self._packed[2]


class LazyLinearIndex(Index):
    """Wraps another index: selection is remembered and applied only when signatures are asked for, and search is
    always the linear (batched) `find` of the base class (index/__init__.py:456-526)."""

    def __init__(self, db, selection_dict={}):
        self.db = db
        self.selection_dict = dict(selection_dict)

    def signatures(self):
        yield from self.db.select(**self.selection_dict).signatures()

    def signatures_with_location(self):
        yield from self.db.select(**self.selection_dict).signatures_with_location()

    def __bool__(self):
        return next(iter(self.signatures()), None) is not None

    def __len__(self):
        return len(self.db.select(**self.selection_dict))

    def select(self, **kwargs):
        _check_select_parameters(**kwargs)
        merged = dict(self.selection_dict)
        for k, v in kwargs.items():
            if k in merged and merged[k] != v:
                raise ValueError(f"cannot select on two different values for {k}")
            merged[k] = v
        return LazyLinearIndex(self.db, merged)


class ZipStorage:
    """Members of one zip file by name (the part of src/sourmash/sbt_storage.py:96-330 that collections of
    signatures use).  mode 'r': read only.  mode 'w': create, or add to an existing file -- a member saved under a
    name that already holds different content gets the next free `name_N`; replaced members (the manifest) take
    effect when the storage is closed."""

    def __init__(self, path, *, mode="r"):
        self.path = os.path.abspath(path)
        self.mode = mode
        self._added = {}                                      # name -> (bytes, deflate?)  written at close()
        if mode == "w" and not os.path.exists(self.path):
            os.makedirs(os.path.dirname(self.path), exist_ok=True)
            self._zf = None
            names = []
        else:
            self._zf = zipfile.ZipFile(self.path, "r")
            names = self._zf.namelist()
        subdirs = [n for n in names if n.endswith("/")]
        self.subdir = subdirs[0] if len(subdirs) == 1 else ""

    @staticmethod
    def can_open(location):
        return zipfile.is_zipfile(location)

    def _filenames(self):
        names = self._zf.namelist() if self._zf is not None else []
        return names + [n for n in self._added if n not in names]

    def _read(self, name):
        if name in self._added:
            return self._added[name][0]
        if self._zf is None:
            raise KeyError(name)
        return self._zf.read(name)

    def load(self, path):
        try:
            return self._read(path)
        except KeyError:
            try:
                return self._read(os.path.join(self.subdir, path))
            except KeyError:
                raise FileNotFoundError(path) from None

    def save(self, path, content, *, overwrite=False, compress=False):
        if self.mode != "w":
            raise NotImplementedError("storage opened read-only")
        name = path
        if not overwrite:
            n = 0
            while True:
                try:
                    if self._read(name) == content:
                        return name                         # same bytes already stored under this name
                except KeyError:
                    break
                name = f"{path}_{n}"
                n += 1
        self._added[name] = (bytes(content), compress)
        return name

    def flush(self):
        pass

    def close(self):
        if self.mode != "w" or not self._added:
            if self._zf is not None:
                self._zf.close()
                self._zf = None
            return
        old = self._zf
        tmp = self.path + ".tmp"
        with zipfile.ZipFile(tmp, "w", compression=zipfile.ZIP_STORED) as out:
            if old is not None:
                for info in old.infolist():
                    if info.filename not in self._added:
                        out.writestr(info, old.read(info), compress_type=info.compress_type)
            for name, (content, deflate) in self._added.items():
                zi = zipfile.ZipInfo(name)
                zi.external_attr = (0o755 if name.endswith("/") else 0o444) << 16
                out.writestr(zi, content, compress_type=zipfile.ZIP_DEFLATED if deflate else zipfile.ZIP_STORED)
        if old is not None:
            old.close()
        os.replace(tmp, self.path)
        self._zf = zipfile.ZipFile(self.path, "r")
        self._added = {}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class ZipFileLinearIndex(Index):
    """A read-only collection of signatures in a zip file, selected through its manifest when it has one and
    loaded from the archive on demand (index/__init__.py:529-732)."""
    is_database = True

    def __init__(self, storage, *, selection_dict=None, traverse_yield_all=False, manifest=None, use_manifest=True):
        self.storage = storage
        self.selection_dict = selection_dict
        self.traverse_yield_all = traverse_yield_all
        self.use_manifest = use_manifest
        self.manifest = None
        if use_manifest:
            self.manifest = manifest if manifest is not None else self._load_manifest()
        if self.manifest is not None:
            assert not self.selection_dict, self.selection_dict
        if self.selection_dict:
            assert self.manifest is None

    def _load_manifest(self):
        try:
            data = self.storage.load("SOURMASH-MANIFEST.csv")
        except (KeyError, FileNotFoundError):
            return None
        return CollectionManifest.load_from_csv(io.StringIO(data.decode("utf-8"), newline=""))

    def __bool__(self):
        return next(iter(self.signatures()), None) is not None

    def __len__(self):
        if self.manifest is not None:
            return len(self.manifest)
        return sum(1 for _ in self.signatures())

    @property
    def location(self):
        return self.storage.path

    @classmethod
    def load(cls, location, traverse_yield_all=False, use_manifest=True):
        if not os.path.exists(location):
            raise FileNotFoundError(location)
        return cls(ZipStorage(location), traverse_yield_all=traverse_yield_all, use_manifest=use_manifest)

    def _member_names(self):
        for name in self.storage._filenames():
            if name.endswith(".sig") or name.endswith(".sig.gz") or self.traverse_yield_all:
                yield name

    def _signatures_with_internal(self):
        for name in self._member_names():
            for ss in load_signatures_from_json(self.storage.load(name)):
                yield ss, name

    def signatures(self):
        if self.manifest is not None:
            assert not self.selection_dict
            for name in self.manifest.locations():
                for ss in load_signatures_from_json(self.storage.load(name)):
                    if ss in self.manifest:               # a member may hold more sketches than were selected
                        yield ss
            return
        sel = self.selection_dict
        for name in self._member_names():
            for ss in load_signatures_from_json(self.storage.load(name)):
                if not sel or select_signature(ss, **sel):
                    yield ss

    # ---- search: the archive goes to HBM once, signatures are materialised for the matches only --------------
    _bulk_cache = None

    def _bulk(self, query_mh):
        """(SketchSet, sizes, rows per member) of the selected sketches, parsed by the native reader straight into one
        CSR in HBM (no signature objects) and downsampled to the query's scaled; kept for the next query.  None when
        the rows cannot share one CSR with this query (num sketches, a coarser scaled, another ksize or molecule
        type) -- the per-signature walk of the base class handles, or rejects, those exactly like the reference."""
        m = self.manifest
        qs = query_mh.scaled
        if m is None or not qs or not m.rows:
            return None
        ksize, moltype = query_mh.ksize, query_mh.moltype
        for row in m.rows:
            if row["num"] or not row["scaled"] or row["scaled"] > qs or row["ksize"] != ksize or row["moltype"] != moltype:
                return None
        if self._bulk_cache is None or self._bulk_cache[0] != (qs, ksize, moltype):
            sset = SketchSet.load([self.storage.path], ksize=ksize, moltype=moltype, scaled=qs)
            by_member = {}
            for r, row in enumerate(sset.manifest):
                by_member.setdefault(row["internal_location"], []).append(r)
            # the order of signatures(): members as the manifest lists them, then the selected sketches within each
            wanted = m._md5_set
            walk = [r for member in m.locations() for r in by_member.get(member, ()) if sset.manifest[r]["md5"] in wanted]
            self._bulk_cache = ((qs, ksize, moltype), sset, sset.sizes, np.asarray(walk, dtype=np.int64))
        return self._bulk_cache[1:]

    def find(self, search_fn, query, **kwargs):
        search_fn.check_is_compatible(query)
        query_mh = query.minhash
        assert not query_mh.track_abundance
        bulk = self._bulk(query_mh)
        if bulk is None:
            yield from Index.find(self, search_fn, query, **kwargs)
            return
        sset, sizes, walk = bulk
        shared = sset.overlaps(query_mh)
        q_size = len(query_mh)
        if _zero_overlap_never_matches(search_fn, q_size):
            walk = walk[shared[walk] > 0]                     # most of a large database: nothing to score, nothing to load
        rows = sset.manifest
        loaded_member, loaded = None, None
        for r in walk.tolist():
            n_shared, s_size = int(shared[r]), int(sizes[r])
            score = search_fn.score_fn(q_size, n_shared, s_size, q_size + s_size - n_shared)
            if not search_fn.passes(score):
                continue
            member, md5 = rows[r]["internal_location"], rows[r]["md5"]
            if member != loaded_member:                      # signatures are materialised for the matches only
                loaded = {ss.md5sum(): ss for ss in load_signatures_from_json(self.storage.load(member))}
                loaded_member = member
            if search_fn.collect(score, loaded[md5]):
                yield IndexSearchResult(score, loaded[md5], self.location)

    def counter_gather(self, query, threshold_bp, **kwargs):
        """The reference prefetches and adds every match to a CounterGather, one signature object each
        (index/__init__.py:302-320).  Here the prefetch is the overlap pass over the resident CSR, the matching rows are
        gathered into their own CSR on the device, and the counter reads signatures from the archive only when a
        round returns one."""
        prefetch_query = query.to_mutable()
        prefetch_query.minhash = prefetch_query.minhash.flatten()
        bulk = self._bulk(prefetch_query.minhash)
        if bulk is None:
            return Index.counter_gather(self, query, threshold_bp, **kwargs)
        if not self:
            raise ValueError("no signatures to search")
        sset, sizes, walk = bulk
        query_mh = prefetch_query.minhash
        search_fn = make_containment_query(query_mh, threshold_bp, best_only=kwargs.get("best_only", False))
        search_fn.check_is_compatible(prefetch_query)
        shared = sset.overlaps(query_mh)
        q_size = len(query_mh)
        walk = walk[shared[walk] > 0]
        keep = [r for r in walk.tolist()
                if search_fn.passes(search_fn.score_fn(q_size, int(shared[r]), int(sizes[r]), q_size + int(sizes[r]) - int(shared[r])))]
        return _ArchiveCounterGather(prefetch_query, self, sset, keep)

    def select(self, **kwargs):
        _check_select_parameters(**kwargs)
        if self.manifest is not None:
            return ZipFileLinearIndex(self.storage, selection_dict=None, traverse_yield_all=self.traverse_yield_all,
                                      manifest=self.manifest.select_to_manifest(**kwargs), use_manifest=True)
        if self.selection_dict:
            merged = dict(self.selection_dict)
            for k, v in kwargs.items():
                if k in merged and merged[k] is not None and merged[k] != v:
                    raise ValueError(f"incompatible select on '{k}'")
                merged[k] = v
            kwargs = merged
        return ZipFileLinearIndex(self.storage, selection_dict=kwargs, traverse_yield_all=self.traverse_yield_all,
                                  manifest=None, use_manifest=False)


def traverse_find_sigs(filenames, yield_all_files=False):
    "every .sig / .sig.gz file in and beneath `filenames` (all files when asked); sourmash_args.py:275-295"
    def wanted(name):
        return yield_all_files or name.endswith((".sig", ".sig.gz"))
    for filename in filenames:
        if os.path.isfile(filename):
            if wanted(filename):
                yield filename
        elif os.path.isdir(filename):
            for root, _dirs, files in os.walk(filename):
                for name in sorted(files):
                    if wanted(os.path.join(root, name)):
                        yield os.path.join(root, name)


def load_pathlist_from_file(filename):
    "a text file of paths, one per line, all of which must exist (sourmash_args.py:380-399)"
    try:
        with open(filename) as fp:
            file_list = set(x.rstrip("\r\n") for x in fp)
    except OSError:
        raise ValueError(f"pathlist file '{filename}' does not exist")
    except UnicodeDecodeError:
        raise ValueError(f"cannot parse file '{filename}' as list of filenames")
    if not file_list:
        raise ValueError("pathlist is empty")
    for checkfile in file_list:
        if not os.path.exists(checkfile):
            raise ValueError(f"file '{checkfile}' inside the pathlist does not exist")
    return file_list


class MultiIndex(Index):
    """Signatures gathered from several indices or files, kept in memory inside a manifest that remembers where each
    one came from (index/__init__.py:910-1125)."""

    def __init__(self, manifest, parent, *, prepend_location=False):
        if prepend_location and parent is None:
            raise ValueError("must set 'parent' if 'prepend_location' is set")
        self.manifest = manifest
        self.parent = parent
        self.prepend_location = prepend_location

    @property
    def location(self):
        return self.parent

    def signatures(self):
        for row in self.manifest.rows:
            yield row["signature"]

    def signatures_with_location(self):
        for row in self.manifest.rows:
            loc = row["internal_location"]
            yield row["signature"], (os.path.join(self.parent, loc) if self.prepend_location else loc)

    def _signatures_with_internal(self):
        for row in self.manifest.rows:
            yield row["signature"], row["internal_location"]

    def __len__(self):
        return 0 if self.manifest is None else len(self.manifest)

    _packed = None

    def _subject_set(self, query_scaled, items):
        "the manifest never changes after construction: keep the device CSR between queries of the same scaled"
        key = (query_scaled, len(items))
        if self._packed is None or self._packed[0] != key:
            self._packed = (key,) + Index._subject_set(self, query_scaled, items)
        return self._packed[1], self._packed[2]

    @classmethod
    def load(cls, index_list, source_list, parent, *, prepend_location=False):
        "from loaded indices and as many sources; a source of None keeps the index's own location"
        assert len(index_list) == len(source_list)

        def sigloc_iter():
            for idx, iloc in zip(index_list, source_list):
                for ss in idx.signatures():
                    yield ss, (idx.location if iloc is None else iloc)
        return cls(CollectionManifest.create_manifest(sigloc_iter()), parent, prepend_location=prepend_location)

    @classmethod
    def load_from_directory(cls, pathname, *, force=False):
        "every .sig / .sig.gz under a directory (every file with force, unreadable ones skipped)"
        from .exceptions import SourmashError
        if not os.path.isdir(pathname):
            raise ValueError(f"'{pathname}' must be a directory.")
        index_list, source_list = [], []
        for thisfile in traverse_find_sigs([pathname], yield_all_files=force):
            try:
                index_list.append(LinearIndex.load(thisfile))
                source_list.append(os.path.relpath(thisfile, pathname))
            except (OSError, SourmashError, ValueError) as exc:
                if not force:
                    raise ValueError(exc)
        if not index_list:
            raise ValueError(f"no signatures to load under directory '{pathname}'")
        return cls.load(index_list, source_list, pathname, prepend_location=True)

    @classmethod
    def load_from_path(cls, pathname, force=False):
        from .exceptions import SourmashError
        if not os.path.exists(pathname):
            raise ValueError(f"'{pathname}' must exist.")
        if os.path.isdir(pathname):
            return cls.load_from_directory(pathname, force=force)
        try:
            idx = LinearIndex.load(pathname)
        except (OSError, SourmashError, ValueError):
            if not force:
                raise ValueError(f"no signatures to load from '{pathname}'")
            return None
        return cls.load([idx], [pathname], pathname)

    @classmethod
    def load_from_pathlist(cls, filename):
        from .save_load import load_file_as_index
        idx_list, src_list = [], []
        for fname in load_pathlist_from_file(filename):
            idx_list.append(load_file_as_index(fname))
            src_list.append(fname)
        return cls.load(idx_list, src_list, filename)

    def select(self, **kwargs):
        _check_select_parameters(**kwargs)
        return MultiIndex(self.manifest.select_to_manifest(**kwargs), self.parent, prepend_location=self.prepend_location)


class StandaloneManifestIndex(Index):
    """A manifest file on its own: selection works on its rows, signatures are loaded from the locations the rows
    name only when asked for (relative paths are taken from the manifest's directory); index/__init__.py:1128-1226."""
    is_database = True

    def __init__(self, manifest, location, *, prefix=None):
        assert manifest is not None
        self.manifest = manifest
        self._location = location
        self.prefix = prefix

    @classmethod
    def load(cls, location, *, prefix=None):
        if not os.path.isfile(location):
            raise ValueError(f"provided manifest location '{location}' is not a file")
        m = CollectionManifest.load_from_filename(location)
        return cls(m, location, prefix=os.path.dirname(location) if prefix is None else prefix)

    @property
    def location(self):
        return self._location

    def signatures_with_location(self):
        yield from self._signatures_with_internal()

    def signatures(self):
        for ss, _ in self._signatures_with_internal():
            yield ss

    def _signatures_with_internal(self):
        "only the rows that survived selection -- the original manifest is not kept"
        from .save_load import load_file_as_index
        picklist = self.manifest.to_picklist()
        for iloc in self.manifest.locations():
            if not iloc.startswith("/") and self.prefix:
                iloc = os.path.join(self.prefix, iloc)
            for ss in load_file_as_index(iloc).select(picklist=picklist).signatures():
                yield ss, iloc

    def __len__(self):
        return len(self.manifest)

    def __bool__(self):
        return bool(self.manifest)

    def select(self, **kwargs):
        _check_select_parameters(**kwargs)
        return StandaloneManifestIndex(self.manifest.select_to_manifest(**kwargs), self._location, prefix=self.prefix)


class CounterGather:
    """Track overlaps between a query and candidate matches for min-set-cover gather.

    Same protocol as the reference class (add / peek / consume, keyed by md5 with
    first-inserted winning ties); the counters live on the GPU.  `add_many` is the
    batch extension: all overlaps in one kernel.
    """

    def __init__(self, query):
        query_mh = query.minhash
        if not query_mh.scaled:
            raise ValueError("gather requires scaled signatures")
        self.orig_query_mh = query_mh.copy().flatten()
        self.scaled = query_mh.scaled
        self.siglist = {}            # md5 -> signature, insertion ordered
        self.locations = {}
        self.query_started = 0
        self._pending = []           # signatures added since the device state was last built
        self._host_counts = {}       # md5 -> overlap, until the device counter exists
        self._dev = None             # (_DeviceCounter, [md5 in CSR order])

    # ---- loading ---------------------------------------------------------------------------------
    def add(self, ss, *, location=None, require_overlap=True):
        "Add one potential match (overlap counted against the original query on the GPU)."
        if self.query_started:
            raise ValueError("cannot add more signatures to counter after peek/consume")
        overlap = self.orig_query_mh.count_common(ss.minhash, True)
        if overlap:
            self._register(ss, overlap, location)
        elif require_overlap:
            raise ValueError("no overlap between query and signature!?")

    def add_many(self, siglist, *, locations=None, require_overlap=False):
        "Batch extension: overlaps of every candidate in one kernel launch."
        if self.query_started:
            raise ValueError("cannot add more signatures to counter after peek/consume")
        siglist = list(siglist)
        if not siglist:
            return
        mhs = [ss.minhash.flatten() for ss in siglist]
        for mh in mhs:                                   # compatibility as count_common(downsample=True) would check
            if mh.ksize != self.orig_query_mh.ksize or mh.moltype != self.orig_query_mh.moltype \
                    or mh.seed != self.orig_query_mh.seed:
                self.orig_query_mh.count_common(mh, True)   # raises the reference's error
        overlaps = _DeviceCounter(SketchSet(mhs), self.orig_query_mh).values()
        for i, ss in enumerate(siglist):
            if overlaps[i]:
                self._register(ss, int(overlaps[i]), locations[i] if locations else None)
            elif require_overlap:
                raise ValueError("no overlap between query and signature!?")

    def _register(self, ss, overlap, location):
        md5 = ss.md5sum()
        if md5 not in self.siglist:
            self._pending.append(md5)
        self._host_counts[md5] = overlap
        self.siglist[md5] = ss
        self.locations[md5] = location
        self.downsample(ss.minhash.scaled)

    def downsample(self, scaled):
        if scaled > self.scaled:
            self.scaled = scaled
        return self.scaled

    def signatures(self):
        yield from self.siglist.values()

    @property
    def union_found(self):
        found_mh = self.orig_query_mh.copy_and_clear()
        for ss in self.siglist.values():
            found_mh.add_many(flatten_and_intersect_scaled(ss.minhash, self.orig_query_mh))
        return found_mh

    # ---- device state ---------------------------------------------------------------------------------
    def _device(self):
        if self._dev is None:
            order = list(self.siglist)
            mhs = [self.siglist[m].minhash.flatten() for m in order]
            dev = _DeviceCounter(SketchSet(mhs), self.orig_query_mh)
            # |Q ∩ D| computed in one launch equals the add-time overlaps (plain set intersections)
            self._dev = (dev, order)
        return self._dev

    @property
    def counter(self):
        "md5 -> remaining overlap: a collections.Counter snapshot of the device counters, zero entries dropped."
        if not self.siglist:
            return Counter()
        dev, order = self._device()
        vals = dev.values()
        return Counter({m: int(v) for m, v in zip(order, vals) if v})

    # ---- gather protocol ---------------------------------------------------------------------------------
    def peek(self, cur_query_mh, *, threshold_bp=0):
        "Best remaining match for the current query, without changing the counters."
        self.query_started = 1
        if not self.siglist:
            return []
        dev, order = self._device()
        scaled = self.downsample(cur_query_mh.scaled)
        cur_query_mh = cur_query_mh.downsample(scaled=scaled)
        if not cur_query_mh:
            return []
        if cur_query_mh.contained_by(self.orig_query_mh, downsample=True) < 1:
            raise ValueError("current query not a subset of original query")
        try:
            threshold, n_threshold_hashes = calc_threshold_from_bp(threshold_bp, scaled, len(cur_query_mh))
        except ValueError:
            return []
        best = dev.best()                               # highest count, ties to the first inserted
        if best is None:
            return []
        idx, match_size = best
        if match_size < n_threshold_hashes:
            return []
        md5 = order[idx]
        match = self.siglist[md5]
        cont = cur_query_mh.contained_by(match.minhash, downsample=True)
        assert cont and cont >= threshold
        match_mh = match.minhash.downsample(scaled=scaled).flatten()
        intersect_mh = cur_query_mh & match_mh
        return (IndexSearchResult(cont, match, self.locations[md5]), intersect_mh)

    def consume(self, intersect_mh):
        "Subtract |intersect ∩ D_d| from every live counter (one kernel)."
        self.query_started = 1
        if not intersect_mh or not self.siglist:
            return
        dev, _ = self._device()
        dev.consume(intersect_mh)

    # ---- batch extension: the whole min-set-cover loop ---------------------------------------------------
    def gather_all(self, threshold_bp=0):
        """Run gather to exhaustion; -> list of (md5, |intersect|) in rank order.  Same decisions as
        GatherDatabases over this single counter, without building result objects.  When every sketch shares
        the query's scaled the whole loop is one native call (smgpu_counter_gather); otherwise it goes round
        by round through peek/consume."""
        if not self.siglist:
            return []
        scaled = self.orig_query_mh.scaled
        if all(ss.minhash.scaled == scaled for ss in self.siglist.values()):
            self.query_started = 1
            dev, order = self._device()
            # search.py:15-37: stop below threshold_bp / scaled hashes; integer counts compare against its ceiling
            thr = math.ceil(float(threshold_bp) / scaled) if threshold_bp else 0
            idx, isect = dev.gather(thr)
            return [(order[int(i)], int(c)) for i, c in zip(idx, isect)]
        out = []
        query_mh = self.orig_query_mh.to_mutable()
        while query_mh:
            res = self.peek(query_mh, threshold_bp=threshold_bp)
            if not res:
                break
            sr, intersect_mh = res
            self.consume(intersect_mh)
            out.append((sr.signature.md5sum(), len(intersect_mh)))
            scaled = max(query_mh.scaled, sr.signature.minhash.scaled)
            query_mh = query_mh.downsample(scaled=scaled).to_mutable() if scaled != query_mh.scaled else query_mh
            query_mh.remove_many(sr.signature.minhash.downsample(scaled=scaled).flatten())
        return out


class _LazySignatures:
    "md5 -> signature for rows of an archive-backed set, read from the archive on first use (insertion ordered)"

    def __init__(self, index, rows):
        self._index = index
        self._rows = {row["md5"]: row for row in rows}      # first row wins, like the md5-keyed dict of the reference
        self._loaded = {}

    def __len__(self):
        return len(self._rows)

    def __bool__(self):
        return bool(self._rows)

    def __iter__(self):
        return iter(self._rows)

    def __contains__(self, md5):
        return md5 in self._rows

    def __getitem__(self, md5):
        if md5 not in self._loaded:
            member = self._rows[md5]["internal_location"]
            for ss in load_signatures_from_json(self._index.storage.load(member)):
                self._loaded.setdefault(ss.md5sum(), ss)
        return self._loaded[md5]

    def keys(self):
        return self._rows.keys()

    def values(self):
        return (self[md5] for md5 in self._rows)

    def items(self):
        return ((md5, self[md5]) for md5 in self._rows)


class _ArchiveCounterGather(CounterGather):
    """CounterGather over rows of a collection that is already in HBM: same protocol, same decisions; the candidate
    rows are gathered into their own CSR on the device instead of being added one signature object at a time."""

    def __init__(self, query, index, sset, rows):
        CounterGather.__init__(self, query)
        # one entry per md5, first occurrence wins (CounterGather keys by md5)
        seen, first = set(), []
        for r in rows:
            md5 = sset.manifest[r]["md5"]
            if md5 not in seen:
                seen.add(md5)
                first.append(r)
        self._cand = sset.subset(first) if first else None
        man = self._cand.manifest if first else []
        self.siglist = _LazySignatures(index, man)
        self.locations = {row["md5"]: index.location for row in man}
        for row in man:                                      # every candidate has the collection's scaled after loading
            self.downsample(row["scaled"])
        self.downsample(sset.params[3])

    def add(self, *args, **kwargs):
        raise ValueError("this counter was built from a resident collection; candidates cannot be added")

    add_many = add

    def _device(self):
        if self._dev is None:
            self._dev = (_DeviceCounter(self._cand, self.orig_query_mh), list(self.siglist))
        return self._dev

    @property
    def union_found(self):
        found_mh = self.orig_query_mh.copy_and_clear()
        for i in range(len(self._cand) if self._cand is not None else 0):
            found_mh.add_many(flatten_and_intersect_scaled(self._cand.minhash(i), self.orig_query_mh))
        return found_mh

    def gather_all(self, threshold_bp=0):
        if not self.siglist:
            return []
        self.query_started = 1
        dev, order = self._device()
        scaled = self.orig_query_mh.scaled
        thr = math.ceil(float(threshold_bp) / scaled) if threshold_bp else 0
        idx, isect = dev.gather(thr)
        return [(order[int(i)], int(c)) for i, c in zip(idx, isect)]
