"""Search / gather drivers (host logic over GPU counters).

API subset of src/sourmash/search.py: calc_threshold_from_bp (:15-37),
SearchType / JaccardSearch / JaccardSearchBestOnly and their factories (:40-160),
_find_best (:755-779), GatherDatabases (:782-949) and a GatherResult carrying
the numeric columns of the reference's gather CSV that derive from sketch
intersections (intersect_bp, f_orig_query, f_match, f_match_orig,
f_unique_to_query, unique_intersect_bp, remaining_bp, rank).  Abundance
weighting and ANI columns belong to the host float layer.
"""
import csv
from enum import Enum

from .signature import SourmashSignature
from .sketchcomparison import FracMinHashComparison, NumMinHashComparison

__all__ = ["calc_threshold_from_bp", "SearchType", "JaccardSearch", "JaccardSearchBestOnly",
           "make_jaccard_search_query", "make_containment_query", "GatherDatabases", "BaseResult", "SearchResult",
           "PrefetchResult", "GatherResult", "search_databases_with_flat_query", "search_databases_with_abund_query",
           "prefetch_database", "format_bp"]


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


class BaseResult:
    """What every result row knows: the two signatures, how they were compared, and how to turn itself into a
    CSV row (search.py:171-280 of the reference: same attribute and column names)."""
    write_cols = None

    def __init__(self, query, match, filename=None, ignore_abundance=False, estimate_ani_ci=False,
                 ani_confidence=0.95, threshold_bp=None, cmp_scaled=None):
        self.query, self.match, self.filename = query, match, filename
        self.ignore_abundance, self.estimate_ani_ci, self.ani_confidence = ignore_abundance, estimate_ani_ci, ani_confidence
        self.threshold_bp, self.cmp_scaled = threshold_bp, cmp_scaled
        self.potential_false_negative = False

    def init_result(self):
        self.mh1, self.mh2 = self.query.minhash, self.match.minhash

    def build_fracminhashcomparison(self):
        self.cmp = FracMinHashComparison(self.mh1, self.mh2, cmp_scaled=self.cmp_scaled, threshold_bp=self.threshold_bp,
                                         ignore_abundance=self.ignore_abundance, estimate_ani_ci=self.estimate_ani_ci,
                                         ani_confidence=self.ani_confidence)
        self.cmp_scaled = self.cmp.cmp_scaled
        self.query_scaled, self.match_scaled = self.mh1.scaled, self.mh2.scaled
        self.size_may_be_inaccurate = self.cmp.size_may_be_inaccurate

    def build_numminhashcomparison(self, cmp_num=None):
        self.cmp = NumMinHashComparison(self.mh1, self.mh2, cmp_num=cmp_num, ignore_abundance=self.ignore_abundance)
        self.cmp_num = self.cmp.cmp_num
        self.query_num, self.match_num = self.mh1.num, self.mh2.num
        self.size_may_be_inaccurate = self.cmp.size_may_be_inaccurate

    def get_cmpinfo(self):
        self.ksize, self.moltype = self.mh1.ksize, self.mh1.moltype
        self.query_name, self.query_filename, self.query_md5 = self.query.name, self.query.filename, self.query.md5sum()
        self.match_name, self.match_filename, self.match_md5 = self.match.name, self.match.filename, self.match.md5sum()
        if self.filename is None and self.match_filename is not None:      # search may pass the location in instead
            self.filename = self.match_filename
        self.md5, self.name = self.match_md5, self.match_name
        self.query_abundance, self.match_abundance = self.mh1.track_abundance, self.mh2.track_abundance
        self.query_n_hashes, self.match_n_hashes = len(self.mh1), len(self.mh2)

    @property
    def pass_threshold(self):
        return self.cmp.pass_threshold

    @staticmethod
    def shorten_md5(md5):
        return md5[:8]

    def to_write(self, columns=()):
        return {k: v for k, v in self.__dict__.items() if k in columns and v is not None}

    def init_dictwriter(self, csv_handle):
        w = csv.DictWriter(csv_handle, fieldnames=self.write_cols)
        w.writeheader()
        return w

    def prep_result(self):
        self.query_md5 = self.shorten_md5(self.query_md5)

    def write(self, w):
        self.prep_result()
        w.writerow(self.to_write(columns=w.fieldnames))

    @property
    def resultdict(self):
        self.prep_result()
        return self.to_write(columns=self.write_cols)


class SearchResult(BaseResult):
    "One `sourmash search` row (search.py:283-366)."
    search_write_cols = ["similarity", "md5", "filename", "name", "query_filename", "query_name", "query_md5", "ani"]
    ci_cols = ["ani_low", "ani_high"]
    search_write_cols_ci = search_write_cols + ci_cols

    def __init__(self, query, match, *, similarity=None, cmp_num=None, searchtype=None, **kw):
        super().__init__(query, match, **kw)
        self.similarity, self.cmp_num, self.searchtype = similarity, cmp_num, searchtype
        self.init_result()
        if self.mh1.scaled or self.mh2.scaled:
            self.build_fracminhashcomparison()
        elif self.mh1.num or self.mh2.num:
            self.build_numminhashcomparison(cmp_num=self.cmp_num)
        self.get_cmpinfo()
        if self.similarity is None:
            raise ValueError("Error: Must provide 'similarity' for SearchResult.")
        if self.cmp_scaled is not None and self.searchtype is not None:
            self.estimate_search_ani()
        self.write_cols = self.search_write_cols_ci if self.estimate_ani_ci else self.search_write_cols

    def estimate_search_ani(self):
        if self.cmp_scaled is None:
            raise TypeError("Error: ANI can only be estimated from scaled signatures.")
        cmp = self.cmp
        if self.searchtype == SearchType.CONTAINMENT:
            cmp.estimate_ani_from_mh1_containment_in_mh2(containment=self.similarity)
            self.ani = cmp.ani_from_mh1_containment_in_mh2
            if self.estimate_ani_ci:
                self.ani_low, self.ani_high = cmp.ani_from_mh1_containment_in_mh2_low, cmp.ani_from_mh1_containment_in_mh2_high
        elif self.searchtype == SearchType.MAX_CONTAINMENT:
            cmp.estimate_max_containment_ani()
            self.ani = cmp.max_containment_ani
            if self.estimate_ani_ci:
                self.ani_low, self.ani_high = cmp.max_containment_ani_low, cmp.max_containment_ani_high
        elif self.searchtype == SearchType.JACCARD:
            cmp.estimate_jaccard_ani(jaccard=self.similarity)
            self.jaccard_ani_untrustworthy = cmp.jaccard_ani_untrustworthy
            self.ani = cmp.jaccard_ani
        self.potential_false_negative = cmp.potential_false_negative


class PrefetchResult(BaseResult):
    "One `sourmash prefetch` row (search.py:369-470)."
    prefetch_write_cols = ["intersect_bp", "jaccard", "max_containment", "f_query_match", "f_match_query",
                           "match_filename", "match_name", "match_md5", "match_bp", "query_filename", "query_name",
                           "query_md5", "query_bp", "ksize", "moltype", "scaled", "query_n_hashes", "query_abundance",
                           "query_containment_ani", "match_containment_ani", "average_containment_ani",
                           "max_containment_ani", "potential_false_negative"]
    ci_cols = ["query_containment_ani_low", "query_containment_ani_high", "match_containment_ani_low",
               "match_containment_ani_high"]
    prefetch_write_cols_ci = prefetch_write_cols + ci_cols

    def __init__(self, query, match, **kw):
        super().__init__(query, match, **kw)
        self.init_sigcomparison()
        self.build_prefetch_result()

    def init_sigcomparison(self):
        self.init_result()
        if not (self.mh1.scaled and self.mh2.scaled):
            raise TypeError("Error: prefetch and gather results must be between scaled signatures.")
        self.build_fracminhashcomparison()
        self.get_cmpinfo()
        self.intersect_bp = self.cmp.total_unique_intersect_hashes
        self.max_containment = self.cmp.max_containment
        self.query_bp, self.match_bp = self.mh1.unique_dataset_hashes, self.mh2.unique_dataset_hashes
        self.threshold = self.threshold_bp
        self.estimate_containment_ani()

    def estimate_containment_ani(self):
        cmp = self.cmp
        cmp.estimate_all_containment_ani()
        self.query_containment_ani = cmp.ani_from_mh1_containment_in_mh2
        self.match_containment_ani = cmp.ani_from_mh2_containment_in_mh1
        self.average_containment_ani = cmp.avg_containment_ani
        self.max_containment_ani = cmp.max_containment_ani
        self.potential_false_negative = cmp.potential_false_negative
        if self.estimate_ani_ci:
            self.query_containment_ani_low, self.query_containment_ani_high = \
                cmp.ani_from_mh1_containment_in_mh2_low, cmp.ani_from_mh1_containment_in_mh2_high
            self.match_containment_ani_low, self.match_containment_ani_high = \
                cmp.ani_from_mh2_containment_in_mh1_low, cmp.ani_from_mh2_containment_in_mh1_high

    def build_prefetch_result(self):
        self.jaccard = self.cmp.jaccard
        self.f_query_match = self.cmp.mh2_containment_in_mh1          # db_mh.contained_by(query_mh)
        self.f_match_query = self.cmp.mh1_containment_in_mh2          # query_mh.contained_by(db_mh)
        self.write_cols = self.prefetch_write_cols_ci if self.estimate_ani_ci else self.prefetch_write_cols

    def prep_prefetch_result(self):
        self.scaled = self.cmp_scaled
        self.query_md5, self.md5, self.match_md5 = (self.shorten_md5(x) for x in (self.query_md5, self.md5, self.match_md5))

    prep_result = prep_prefetch_result

    @property
    def prefetchresultdict(self):
        self.prep_prefetch_result()
        return self.to_write(columns=self.write_cols)


class GatherResult(PrefetchResult):
    """One `sourmash gather` row (search.py:473-665): the prefetch numbers of (original query, match) plus the
    numbers of (remaining query, match) and, unless abundances are ignored, the abundance-weighted columns."""
    gather_write_cols = ["intersect_bp", "f_orig_query", "f_match", "f_unique_to_query", "f_unique_weighted",
                         "average_abund", "median_abund", "std_abund", "filename", "name", "md5", "f_match_orig",
                         "unique_intersect_bp", "gather_result_rank", "remaining_bp", "query_filename", "query_name",
                         "query_md5", "query_bp", "ksize", "moltype", "scaled", "query_n_hashes", "query_abundance",
                         "query_containment_ani", "match_containment_ani", "average_containment_ani",
                         "max_containment_ani", "potential_false_negative", "n_unique_weighted_found",
                         "sum_weighted_found", "total_weighted_hashes"]
    gather_write_cols_ci = gather_write_cols + PrefetchResult.ci_cols

    def __init__(self, query, match, *, gather_querymh=None, gather_result_rank=None, orig_query_len=None,
                 orig_query_abunds=None, sum_weighted_found=None, total_weighted_hashes=None, noident_len=0, **kw):
        BaseResult.__init__(self, query, match, **kw)
        self.gather_querymh, self.gather_result_rank = gather_querymh, gather_result_rank
        self.orig_query_len, self.orig_query_abunds = orig_query_len, orig_query_abunds
        self.sum_weighted_found, self.total_weighted_hashes = sum_weighted_found, total_weighted_hashes
        self.noident_len = noident_len
        self.check_gatherresult_input()
        self.init_sigcomparison()                                        # original query vs match
        self.gather_comparison = FracMinHashComparison(self.gather_querymh, self.match.minhash.flatten())   # remaining vs match
        self.build_gather_result()
        self.write_cols = self.gather_write_cols_ci if self.estimate_ani_ci else self.gather_write_cols

    def check_gatherresult_input(self):
        if self.cmp_scaled is None:
            raise ValueError("Error: must provide comparison scaled value ('cmp_scaled') for GatherResult")
        if self.gather_querymh is None:
            raise ValueError("Error: must provide current gather sketch (remaining hashes) for GatherResult")
        if self.gather_result_rank is None:
            raise ValueError("Error: must provide 'gather_result_rank' to GatherResult")
        if not self.total_weighted_hashes:
            raise ValueError("Error: must provide sum of all abundances ('total_weighted_hashes') to GatherResult")
        if not self.orig_query_abunds:
            raise ValueError("Error: must provide original query abundances ('orig_query_abunds') to GatherResult")

    def build_gather_result(self):
        g = self.gather_comparison
        # the query handed to gather is what remained after subtracting the unidentifiable hashes
        self.query_bp = self.orig_query_len * self.query.minhash.scaled
        self.query_n_hashes = self.orig_query_len
        unique_isect = g.intersect_mh                                   # remaining query ∩ match
        self.n_intersect = len(unique_isect)
        self.unique_intersect_bp = self.n_intersect * g.cmp_scaled
        self.f_match_orig = self.cmp.mh2_containment_in_mh1
        self.f_match = g.mh2_containment_in_mh1
        self.f_orig_query = self.cmp.mh1_cmp.flatten().count_common(self.cmp.mh2_cmp.flatten()) / self.orig_query_len
        self.f_unique_to_query = self.n_intersect / self.orig_query_len
        self.remaining_bp = self.noident_len + g.mh1_cmp.unique_dataset_hashes - self.unique_intersect_bp
        self.average_abund = self.median_abund = self.std_abund = None
        if self.ignore_abundance:
            self.f_unique_weighted = self.f_unique_to_query
            self.query_abundance = False
            return
        weighted = g.weighted_intersection(from_abundD=self.orig_query_abunds)
        self.query_weighted_unique_intersection = weighted
        self.average_abund, self.median_abund, self.std_abund = \
            weighted.mean_abundance, weighted.median_abundance, weighted.std_abundance
        self.query_abundance = weighted.track_abundance
        self.n_unique_weighted_found = weighted.sum_abundances
        self.f_unique_weighted = self.n_unique_weighted_found / self.total_weighted_hashes

    def prep_gather_result(self):
        self.scaled = self.cmp_scaled
        self.query_md5 = self.shorten_md5(self.query_md5)

    prep_result = prep_gather_result

    @property
    def gatherresultdict(self):
        self.prep_gather_result()
        return self.to_write(columns=self.write_cols)

    @property
    def prefetchresultdict(self):
        cols = self.prefetch_write_cols_ci if self.estimate_ani_ci else self.prefetch_write_cols
        self.jaccard = self.cmp.jaccard
        self.f_query_match = self.cmp.mh2_containment_in_mh1
        self.f_match_query = self.cmp.mh1_containment_in_mh2
        self.prep_prefetch_result()
        return self.to_write(columns=cols)


def _dedup_sorted(hits):
    "first occurrence of every md5, best score first (search.py:668-752)"
    seen, out = set(), []
    for score, match, filename in hits:
        md5 = match.md5sum()
        if md5 not in seen:
            seen.add(md5)
            out.append((score, match, filename))
    out.sort(key=lambda t: -t[0])
    return out


def search_databases_with_flat_query(query, databases, **kwargs):
    "`sourmash search` over several databases with a flat query -> [SearchResult], best first"
    hits = _dedup_sorted(hit for db in databases for hit in db.search(query, **kwargs))
    search_type, ci = SearchType.JACCARD, False
    if kwargs.get("do_containment"):
        search_type, ci = SearchType.CONTAINMENT, bool(kwargs.get("estimate_ani_ci"))
    elif kwargs.get("do_max_containment"):
        search_type, ci = SearchType.MAX_CONTAINMENT, bool(kwargs.get("estimate_ani_ci"))
    return [SearchResult(query, match, similarity=score, filename=filename, searchtype=search_type, estimate_ani_ci=ci)
            for score, match, filename in hits]


def search_databases_with_abund_query(query, databases, **kwargs):
    "`sourmash search` with an abundance query (angular similarity) -> [SearchResult]"
    if kwargs.get("do_containment") or kwargs.get("do_max_containment"):
        raise TypeError("containment searches cannot be done with abund sketches")
    hits = _dedup_sorted(hit for db in databases for hit in db.search_abund(query, **kwargs))
    return [SearchResult(query, match, similarity=score, filename=filename) for score, match, filename in hits]


def prefetch_database(query, database, threshold_bp, *, estimate_ani_ci=False):
    "every match sharing >= threshold_bp with the query -> PrefetchResult rows (search.py:956-976)"
    scaled = query.minhash.scaled
    assert scaled
    for hit in database.prefetch(query, threshold_bp):
        yield PrefetchResult(query, hit.signature, threshold_bp=threshold_bp, estimate_ani_ci=estimate_ani_ci)


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
    """Iterator performing gather / min-set-cover over CounterGather objects (search.py:782-949).

    Round by round: best match over all counters (GPU arg-max per counter), consume everywhere (GPU), subtract the
    match from the query, and report the round as a GatherResult (with abundance-weighted columns when the
    query tracks abundance and ignore_abundance is not set)."""

    def __init__(self, query, counters, *, threshold_bp=0, ignore_abundance=False, noident_mh=None, ident_mh=None,
                 estimate_ani_ci=False):
        query_mh = query.minhash
        self.track_abundance = bool(query_mh.track_abundance and not ignore_abundance)
        self.orig_query = query
        self.orig_query_bp = query_mh.unique_dataset_hashes
        self.orig_query_filename, self.orig_query_name = query.filename, query.name
        self.orig_query_md5 = query.md5sum()[:8]
        hashes = query_mh.hashes
        self.orig_query_abunds = dict(hashes) if self.track_abundance else {h: 1 for h in hashes}
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
        self.estimate_ani_ci = estimate_ani_ci
        self.cmp_scaled = 0
        self._update_scaled(orig_query_mh.scaled)

    def _update_scaled(self, scaled):
        max_scaled = max(self.cmp_scaled, scaled)
        if self.cmp_scaled != max_scaled:
            self.cmp_scaled = max_scaled
            self.orig_query_mh = self.orig_query_mh.downsample(scaled=scaled)
            self.noident_mh = self.noident_mh.downsample(scaled=scaled)
            abunds = self.orig_query_abunds                             # usable as is at any coarser scaled
            self.noident_query_sum_abunds = sum(abunds[h] for h in self.noident_mh.hashes)
            self.total_weighted_hashes = sum(abunds[h] for h in self.orig_query_mh.hashes) + self.noident_query_sum_abunds
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
        assert best_match.minhash.scaled
        scaled = self._update_scaled(best_match.minhash.scaled)          # the coarsest resolution seen so far
        orig_query_len = len(self.orig_query_mh) + len(self.noident_mh)
        query_mh = query.minhash.downsample(scaled=scaled)
        found_mh = best_match.minhash.downsample(scaled=scaled).flatten()
        new_query_mh = query_mh.to_mutable()
        new_query_mh.remove_many(found_mh)                              # the WHOLE match leaves the query (:915-919)
        abunds = self.orig_query_abunds
        n_weighted_missed = sum(abunds[h] for h in new_query_mh.hashes) + self.noident_query_sum_abunds
        result = GatherResult(
            self.orig_query, best_match, cmp_scaled=scaled, filename=best_result.location,
            gather_result_rank=self.result_n, gather_querymh=query.minhash, ignore_abundance=not self.track_abundance,
            threshold_bp=self.threshold_bp, orig_query_len=orig_query_len, orig_query_abunds=abunds,
            estimate_ani_ci=self.estimate_ani_ci, sum_weighted_found=self.total_weighted_hashes - n_weighted_missed,
            total_weighted_hashes=self.total_weighted_hashes, noident_len=len(self.noident_mh) * self.noident_mh.scaled)
        self.result_n += 1
        self.query = SourmashSignature(new_query_mh)
        return result
