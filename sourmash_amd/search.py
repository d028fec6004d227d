"""Search / gather drivers (host logic over GPU counters).

API subset of src/sourmash/search.py: calc_threshold_from_bp (:15-37),
SearchType / JaccardSearch / JaccardSearchBestOnly and their factories (:40-160),
_find_best (:755-779), GatherDatabases (:782-949) and a GatherResult carrying
the numeric columns of the reference's gather CSV that derive from sketch
intersections (intersect_bp, f_orig_query, f_match, f_match_orig,
f_unique_to_query, unique_intersect_bp, remaining_bp, rank).  Abundance
weighting and ANI columns belong to the host float layer.
"""
from dataclasses import dataclass
from enum import Enum

from .signature import SourmashSignature

__all__ = ["calc_threshold_from_bp", "SearchType", "JaccardSearch", "JaccardSearchBestOnly",
           "make_jaccard_search_query", "make_containment_query", "GatherDatabases", "GatherResult", "format_bp"]


def calc_threshold_from_bp(threshold_bp, scaled, query_size):
    "threshold_bp -> (containment threshold, minimum number of hashes); ValueError if unattainable."
    threshold, n_threshold_hashes = 0.0, 0
    if threshold_bp:
        if threshold_bp < 0:
            raise TypeError("threshold_bp must be non-negative")
        n_threshold_hashes = float(threshold_bp) / scaled
        threshold = n_threshold_hashes / query_size
        if threshold > 1.0:
            raise ValueError("requested threshold_bp is unattainable with this query")
    return threshold, n_threshold_hashes


class SearchType(Enum):
    JACCARD = 1
    CONTAINMENT = 2
    MAX_CONTAINMENT = 3


class JaccardSearch:
    "Scoring + threshold protocol used by Index.find (search.py:88-160)."

    def __init__(self, search_type, threshold=None):
        self.search_type = search_type
        self.require_scaled = search_type in (SearchType.CONTAINMENT, SearchType.MAX_CONTAINMENT)
        self.score_fn = {SearchType.JACCARD: self.score_jaccard, SearchType.CONTAINMENT: self.score_containment,
                         SearchType.MAX_CONTAINMENT: self.score_max_containment}[search_type]
        self.threshold = float(threshold or 0)

    def check_is_compatible(self, sig):
        if self.require_scaled and not sig.minhash.scaled:
            raise TypeError("this search requires a scaled signature")
        if sig.minhash.track_abundance:
            raise TypeError("this search cannot be done with an abund signature")

    def passes(self, score):
        return bool(score and score >= self.threshold)

    def collect(self, score, match_sig):
        return True

    @staticmethod
    def score_jaccard(query_size, shared_size, subject_size, total_size):
        return shared_size / total_size if total_size else 0

    @staticmethod
    def score_containment(query_size, shared_size, subject_size, total_size):
        return shared_size / query_size if query_size else 0

    @staticmethod
    def score_max_containment(query_size, shared_size, subject_size, total_size):
        m = min(query_size, subject_size)
        return shared_size / m if m else 0


class JaccardSearchBestOnly(JaccardSearch):
    def collect(self, score, match):
        self.threshold = max(self.threshold, score)
        return True


def make_jaccard_search_query(*, do_containment=False, do_max_containment=False, best_only=False, threshold=None):
    if do_containment and do_max_containment:
        raise TypeError("'do_containment' and 'do_max_containment' cannot both be True")
    cls = JaccardSearchBestOnly if best_only else JaccardSearch
    kind = SearchType.CONTAINMENT if do_containment else SearchType.MAX_CONTAINMENT if do_max_containment \
        else SearchType.JACCARD
    return cls(kind, threshold)


def make_containment_query(query_mh, threshold_bp, *, best_only=True):
    if not query_mh:
        raise ValueError("query is empty!?")
    if not query_mh.scaled:
        raise TypeError("query signature must be calculated with scaled")
    threshold, _ = calc_threshold_from_bp(threshold_bp, query_mh.scaled, len(query_mh))
    return (JaccardSearchBestOnly if best_only else JaccardSearch)(SearchType.CONTAINMENT, threshold=threshold)


def format_bp(bp):
    bp = float(bp)
    if bp < 500:
        return f"{bp:.0f} bp"
    if bp <= 500e3:
        return f"{round(bp / 1e3, 1):.1f} kbp"
    if bp < 500e6:
        return f"{round(bp / 1e6, 1):.1f} Mbp"
    if bp < 500e9:
        return f"{round(bp / 1e9, 1):.1f} Gbp"
    return "???"


@dataclass
class GatherResult:
    "One gather round (numeric columns of search.py:473-620 that come from sketch intersections)."
    match: SourmashSignature
    filename: str
    gather_result_rank: int
    cmp_scaled: int
    intersect_bp: int            # |orig query ∩ match| * scaled
    unique_intersect_bp: int     # |remaining query ∩ match| * scaled
    f_orig_query: float
    f_match: float               # match contained in the REMAINING query (de-biased)
    f_match_orig: float          # match contained in the ORIGINAL query (de-biased)
    f_unique_to_query: float
    f_unique_weighted: float
    remaining_bp: int
    query_bp: int
    query_n_hashes: int
    n_intersect: int             # |remaining query ∩ match| in hashes (the golden gather sizes)

    @property
    def name(self):
        return self.match.name

    @property
    def md5(self):
        return self.match.md5sum()

    @property
    def gatherresultdict(self):
        d = {k: getattr(self, k) for k in ("intersect_bp", "f_orig_query", "f_match", "f_unique_to_query",
                                           "f_unique_weighted", "f_match_orig", "unique_intersect_bp",
                                           "gather_result_rank", "remaining_bp", "query_bp", "query_n_hashes")}
        d.update(filename=self.filename, name=self.name, md5=self.md5, scaled=self.cmp_scaled,
                 ksize=self.match.minhash.ksize, moltype=self.match.minhash.moltype)
        return d


def _find_best(counters, query, threshold_bp):
    "search.py:755-779: best score over counters (strict >: the first counter wins ties), then consume everywhere."
    best_result, best_intersect_mh = None, None
    for counter in counters:
        result = counter.peek(query.minhash, threshold_bp=threshold_bp)
        if result:
            sr, intersect_mh = result
            if best_result is None or sr.score > best_result.score:
                best_result, best_intersect_mh = sr, intersect_mh
    if best_result:
        for counter in counters:
            counter.consume(best_intersect_mh)
        return best_result, best_intersect_mh
    return None, None


class GatherDatabases:
    "Iterator performing gather / min-set-cover over CounterGather objects (search.py:782-949)."

    def __init__(self, query, counters, *, threshold_bp=0, ignore_abundance=False, noident_mh=None, ident_mh=None,
                 estimate_ani_ci=False):
        self.orig_query = query
        query_mh = query.minhash
        if noident_mh is None:
            noident_mh = query_mh.copy_and_clear()
        self.noident_mh = noident_mh.to_frozen()
        if ident_mh is None:
            query_mh = query_mh.to_mutable()
            query_mh.remove_many(noident_mh)
        else:
            query_mh = ident_mh.to_mutable()
        orig_query_mh = query_mh.flatten()
        query = query.to_mutable()
        query.minhash = orig_query_mh
        self.query = query
        self.counters = counters
        self.threshold_bp = threshold_bp
        self.result_n = 0
        self.orig_query_mh = orig_query_mh
        self.cmp_scaled = 0
        self._update_scaled(orig_query_mh.scaled)

    def _update_scaled(self, scaled):
        max_scaled = max(self.cmp_scaled, scaled)
        if self.cmp_scaled != max_scaled:
            self.cmp_scaled = max_scaled
            self.orig_query_mh = self.orig_query_mh.downsample(scaled=scaled)
            self.noident_mh = self.noident_mh.downsample(scaled=scaled)
        return max_scaled

    @property
    def scaled(self):
        return self.cmp_scaled

    def __iter__(self):
        return self

    def __next__(self):
        query = self.query
        if not query.minhash:
            raise StopIteration
        best_result, intersect_mh = _find_best(self.counters, query, self.threshold_bp)
        if not best_result:
            raise StopIteration
        best_match = best_result.signature
        scaled = self._update_scaled(best_match.minhash.scaled)
        orig_query_mh, noident_mh = self.orig_query_mh, self.noident_mh
        orig_query_len = len(orig_query_mh) + len(noident_mh)
        query_mh = query.minhash.downsample(scaled=scaled)
        found_mh = best_match.minhash.downsample(scaled=scaled).flatten()

        # numbers of this round, all from sketch intersections
        n_unique = len(intersect_mh)                                   # |remaining query ∩ match|
        n_orig = orig_query_mh.count_common(found_mh)                  # |orig query ∩ match|
        result = GatherResult(
            match=best_match, filename=best_result.location, gather_result_rank=self.result_n, cmp_scaled=scaled,
            intersect_bp=n_orig * scaled, unique_intersect_bp=n_unique * scaled,
            f_orig_query=n_orig / orig_query_len,
            f_match=found_mh.contained_by(query_mh), f_match_orig=found_mh.contained_by(orig_query_mh),
            f_unique_to_query=n_unique / orig_query_len, f_unique_weighted=n_unique / orig_query_len,
            remaining_bp=len(noident_mh) * noident_mh.scaled + len(query_mh) * scaled - n_unique * scaled,
            query_bp=orig_query_len * self.orig_query.minhash.scaled, query_n_hashes=orig_query_len,
            n_intersect=n_unique)

        new_query_mh = query_mh.to_mutable()
        new_query_mh.remove_many(found_mh)                              # the WHOLE match leaves the query (:915-919)
        self.query = SourmashSignature(new_query_mh)
        self.result_n += 1
        return result
