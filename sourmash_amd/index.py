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
from collections import Counter, namedtuple

import numpy as np

from ._lowlevel import lib
from .minhash import flatten_and_downsample_num, flatten_and_downsample_scaled, flatten_and_intersect_scaled
from .search import calc_threshold_from_bp, make_containment_query, make_jaccard_search_query
from .signature import SourmashSignature, load_signatures_from_json, save_signatures_to_json
from .utils import RustObject, decode_str, objptr_array, rustcall

__all__ = ["IndexSearchResult", "Collection", "SketchSet", "select_signature", "Index", "LinearIndex", "CounterGather"]

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


def rank_search_hits(shared, sizes, n_query, *, threshold=0.0, do_containment=False, do_max_containment=False, best_only=False):
    """[(score, row)] best first from one overlap pass: shared[r] = |query ∩ row r|, sizes[r] = |row r|.  Jaccard by default,
    query containment or max containment on request -- the scores of JaccardSearch (search.py:88-160); rows scoring below
    the threshold or sharing nothing are dropped (index/__init__.py:115-170).  Shared by SketchSet.search (one GPU) and
    parallel.search_distributed (database sharded over the ranks)."""
    shared = np.asarray(shared).astype(np.float64)
    sizes = np.asarray(sizes).astype(np.float64)
    nq = float(n_query)
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


def prefetch_rows(shared, threshold_bp, scaled):
    "[(row, |intersect|)] in row order for the rows sharing at least threshold_bp with the query (search.py:956-976)"
    shared = np.asarray(shared)
    need = float(threshold_bp) / scaled if threshold_bp else 0.0
    return [(int(r), int(shared[r])) for r in np.flatnonzero((shared >= need) & (shared > 0))]


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
        ptrs, _alive = objptr_array(self._keep)
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
    def skipped(self):
        "sketches seen in the inputs that the selection left out (a loaded set)"
        return self._methodcall(lib.smgpu_sketchset_skipped)

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
        return rank_search_hits(self.overlaps(query_mh), self.sizes, len(query_mh), threshold=threshold,
                                do_containment=do_containment, do_max_containment=do_max_containment, best_only=best_only)

    def prefetch(self, query_mh, threshold_bp=0):
        "Rows sharing at least threshold_bp with the query -> [(row, |intersect|)] in row order (search.py:956-976)."
        scaled = query_mh.scaled
        if not scaled:
            raise ValueError("prefetch requires scaled signatures")
        return prefetch_rows(self.overlaps(query_mh), threshold_bp, scaled)

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
