"""Pairwise comparison objects behind the search / prefetch / gather result rows.

Interface of src/sourmash/sketchcomparison.py (BaseMinHashComparison :12-80, NumMinHashComparison :83-97,
FracMinHashComparison :100-256): two sketches brought to a common resolution once, then every number the
result columns need as an attribute or property of the same names.  Counts come from the GPU intersections
behind MinHash; the ANI / containment floats are the host float layer (distance_utils)."""
from .minhash import MinHash

__all__ = ["BaseMinHashComparison", "NumMinHashComparison", "FracMinHashComparison"]


class BaseMinHashComparison:
    "Two sketches at one resolution (flattened first when ignore_abundance is set)."

    def __init__(self, mh1, mh2, ignore_abundance=False, jaccard_ani_untrustworthy=False):
        self.mh1, self.mh2 = mh1, mh2
        self.ignore_abundance = ignore_abundance
        self.jaccard_ani_untrustworthy = jaccard_ani_untrustworthy

    def downsample_and_handle_ignore_abundance(self, cmp_num=None, cmp_scaled=None):
        a, b = (self.mh1.flatten(), self.mh2.flatten()) if self.ignore_abundance else (self.mh1, self.mh2)
        if cmp_scaled is not None:
            a, b = a.downsample(scaled=cmp_scaled), b.downsample(scaled=cmp_scaled)
        elif cmp_num is not None:
            a, b = a.downsample(num=cmp_num), b.downsample(num=cmp_num)
        else:
            raise ValueError("Error: must pass in a comparison scaled or num value.")
        self.mh1_cmp, self.mh2_cmp = a, b

    def check_compatibility_and_downsample(self, cmp_num=None, cmp_scaled=None):
        both_num = self.mh1.num and self.mh2.num
        both_scaled = self.mh1.scaled and self.mh2.scaled
        if not (both_num or both_scaled):
            raise TypeError("Error: Both sketches must be 'num' or 'scaled'.")
        self.downsample_and_handle_ignore_abundance(cmp_num=cmp_num, cmp_scaled=cmp_scaled)   # is_compatible looks at scaled
        if not self.mh1_cmp.is_compatible(self.mh2_cmp):
            raise TypeError("Error: Cannot compare incompatible sketches.")
        self.ksize, self.moltype = self.mh1.ksize, self.mh1.moltype

    @property
    def intersect_mh(self):
        return self.mh1_cmp.flatten().intersection(self.mh2_cmp.flatten())

    @property
    def jaccard(self):
        return self.mh1_cmp.jaccard(self.mh2_cmp)

    def estimate_jaccard_ani(self, jaccard=None):
        info = self.mh1_cmp.jaccard_ani(self.mh2_cmp, jaccard=jaccard)
        self.jaccard_ani = info.ani
        if info.p_exceeds_threshold:
            self.potential_false_negative = True
        self.jaccard_ani_untrustworthy = info.je_exceeds_threshold

    @property
    def angular_similarity(self):
        return self.mh1_cmp.angular_similarity(self.mh2_cmp)      # TypeError when either side is flat

    cosine_similarity = angular_similarity


class NumMinHashComparison(BaseMinHashComparison):
    "Bottom-k sketches compared at the smaller num."

    def __init__(self, mh1, mh2, ignore_abundance=False, jaccard_ani_untrustworthy=False, cmp_num=None):
        super().__init__(mh1, mh2, ignore_abundance, jaccard_ani_untrustworthy)
        self.cmp_num = min(mh1.num, mh2.num) if cmp_num is None else cmp_num
        self.check_compatibility_and_downsample(cmp_num=self.cmp_num)

    @property
    def size_may_be_inaccurate(self):
        return False


class FracMinHashComparison(BaseMinHashComparison):
    "Scaled sketches compared at the coarser scaled (or a forced cmp_scaled)."

    def __init__(self, mh1, mh2, ignore_abundance=False, jaccard_ani_untrustworthy=False, cmp_scaled=None,
                 threshold_bp=0, estimate_ani_ci=False, ani_confidence=0.95):
        super().__init__(mh1, mh2, ignore_abundance, jaccard_ani_untrustworthy)
        self.cmp_scaled = max(mh1.scaled, mh2.scaled) if cmp_scaled is None else cmp_scaled
        self.threshold_bp, self.estimate_ani_ci, self.ani_confidence = threshold_bp, estimate_ani_ci, ani_confidence
        self.check_compatibility_and_downsample(cmp_scaled=self.cmp_scaled)
        self.potential_false_negative = False

    @property
    def pass_threshold(self):
        return self.total_unique_intersect_hashes >= self.threshold_bp

    @property
    def size_may_be_inaccurate(self):
        return not (self.mh1_cmp.size_is_accurate() and self.mh2_cmp.size_is_accurate())

    @property
    def total_unique_intersect_hashes(self):
        "|mh1 ∩ mh2| * scaled: the 'bp' of the result tables (hashes, not k-1-corrected bases)"
        return self.mh1_cmp.flatten().count_common(self.mh2_cmp.flatten()) * self.cmp_scaled

    @property
    def mh1_containment_in_mh2(self):
        return self.mh1_cmp.contained_by(self.mh2_cmp)

    @property
    def mh2_containment_in_mh1(self):
        return self.mh2_cmp.contained_by(self.mh1_cmp)

    @property
    def max_containment(self):
        return self.mh1_cmp.max_containment(self.mh2_cmp)

    @property
    def avg_containment(self):
        return self.mh1_cmp.avg_containment(self.mh2_cmp)

    def _record_ani(self, prefix, info):
        setattr(self, prefix, info.ani)
        if info.p_exceeds_threshold:
            self.potential_false_negative = True
        if self.estimate_ani_ci:
            setattr(self, prefix + "_low", info.ani_low)
            setattr(self, prefix + "_high", info.ani_high)

    def estimate_ani_from_mh1_containment_in_mh2(self, containment=None):
        self._record_ani("ani_from_mh1_containment_in_mh2", self.mh1_cmp.containment_ani(
            self.mh2_cmp, containment=containment, confidence=self.ani_confidence, estimate_ci=self.estimate_ani_ci))

    def estimate_ani_from_mh2_containment_in_mh1(self, containment=None):
        self._record_ani("ani_from_mh2_containment_in_mh1", self.mh2_cmp.containment_ani(
            self.mh1_cmp, containment=containment, confidence=self.ani_confidence, estimate_ci=self.estimate_ani_ci))

    def estimate_max_containment_ani(self, max_containment=None):
        self._record_ani("max_containment_ani", self.mh1_cmp.max_containment_ani(
            self.mh2_cmp, max_containment=max_containment, confidence=self.ani_confidence,
            estimate_ci=self.estimate_ani_ci))

    def _both_containment_anis(self):
        self.estimate_ani_from_mh1_containment_in_mh2()
        self.estimate_ani_from_mh2_containment_in_mh1()
        pair = (self.ani_from_mh1_containment_in_mh2, self.ani_from_mh2_containment_in_mh1)
        return None if None in pair else pair

    @property
    def avg_containment_ani(self):
        pair = self._both_containment_anis()
        return None if pair is None else (pair[0] + pair[1]) / 2

    def estimate_all_containment_ani(self):
        pair = self._both_containment_anis()
        self.max_containment_ani = None if pair is None else max(pair)

    def weighted_intersection(self, from_mh=None, from_abundD={}):
        "the intersection carrying abundances from `from_mh` / a {hash: abundance} map (missing hashes count 1)"
        isect = self.intersect_mh
        if from_mh is not None and from_mh.track_abundance:
            from_abundD = from_mh.hashes
        if not from_abundD:
            return isect
        weighted = isect.copy_and_clear()
        weighted.track_abundance = True
        weighted.set_abundances({h: from_abundD.get(h, 1) for h in isect.hashes})
        return weighted
