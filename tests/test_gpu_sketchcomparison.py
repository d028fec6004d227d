"""FracMinHashComparison / NumMinHashComparison on the GPU path, following the reference's
tests/test_sketchcomparison.py (cited per block).  Run with -m gpu."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu

A_VALUES = {1: 5, 3: 3, 5: 2, 8: 2}
B_VALUES = {1: 3, 3: 2, 5: 1, 6: 1, 8: 1, 10: 1}
ANGULAR = "Error: Angular (cosine) similarity requires both sketches to track hash abundance."


@pytest.fixture(scope="module")
def sm():
    import torch  # noqa: F401
    import sourmash_amd
    assert sourmash_amd.gpu_available()
    return sourmash_amd


@pytest.fixture(params=[True, False])
def track_abundance(request):
    return request.param


def _pair(sm, track_abundance, a_kw, b_kw, ksize_a=21, ksize_b=21):
    a = sm.MinHash(a_kw.pop("n", 0), ksize_a, track_abundance=track_abundance, **a_kw)
    b = sm.MinHash(b_kw.pop("n", 0), ksize_b, track_abundance=track_abundance, **b_kw)
    if track_abundance:
        a.set_abundances(A_VALUES)
        b.set_abundances(B_VALUES)
    else:
        a.add_many(A_VALUES.keys())
        b.add_many(B_VALUES.keys())
    return a, b


def test_frac_comparison_fields(sm, track_abundance):
    # :16-88, 270-369
    from sourmash_amd.sketchcomparison import FracMinHashComparison
    a, b = _pair(sm, track_abundance, {"scaled": 1}, {"scaled": 1})
    cmp = FracMinHashComparison(a, b)
    assert cmp.mh1 == a and cmp.mh2 == b and cmp.ignore_abundance is False
    assert (cmp.cmp_scaled, cmp.ksize, cmp.moltype) == (1, 21, "DNA")
    assert cmp.mh1_containment_in_mh2 == a.contained_by(b) and cmp.mh2_containment_in_mh1 == b.contained_by(a)
    assert cmp.avg_containment == a.avg_containment(b) and cmp.max_containment == a.max_containment(b)
    assert cmp.jaccard == a.jaccard(b) == b.jaccard(a)
    intersect_mh = a.flatten().intersection(b.flatten())
    assert cmp.intersect_mh == intersect_mh == b.flatten().intersection(a.flatten())
    assert cmp.total_unique_intersect_hashes == 4 and cmp.pass_threshold
    if track_abundance:
        assert cmp.angular_similarity == a.angular_similarity(b) == cmp.cosine_similarity
        assert cmp.weighted_intersection(from_mh=cmp.mh1).hashes == intersect_mh.inflate(a).hashes
        assert cmp.weighted_intersection(from_mh=cmp.mh2).hashes == intersect_mh.inflate(b).hashes
        assert cmp.weighted_intersection(from_abundD=A_VALUES).hashes == intersect_mh.inflate(a).hashes
        assert cmp.weighted_intersection(from_abundD=B_VALUES).hashes == intersect_mh.inflate(b).hashes
    else:
        for attr in ("angular_similarity", "cosine_similarity"):
            with pytest.raises(TypeError) as exc:
                getattr(cmp, attr)
            assert ANGULAR in str(exc.value)
        assert cmp.weighted_intersection(from_mh=cmp.mh1).hashes == intersect_mh.hashes
    # a forced coarser scaled, with and without abundances, and a threshold that the overlap misses
    ds_a, ds_b = a.flatten().downsample(scaled=2), b.flatten().downsample(scaled=2)
    flat = FracMinHashComparison(a, b, cmp_scaled=2, ignore_abundance=True)
    assert flat.mh1_cmp == ds_a and flat.mh2_cmp == ds_b and flat.cmp_scaled == 2
    assert not flat.mh1_cmp.track_abundance and not flat.mh2_cmp.track_abundance
    assert flat.jaccard == ds_a.jaccard(ds_b) and flat.total_unique_intersect_hashes == 8 and flat.pass_threshold
    assert flat.avg_containment == ds_b.avg_containment(ds_a) and flat.max_containment == ds_a.max_containment(ds_b)
    with pytest.raises(TypeError) as exc:
        flat.angular_similarity
    assert ANGULAR in str(exc.value)
    strict = FracMinHashComparison(a, b, cmp_scaled=2, threshold_bp=40)
    assert strict.total_unique_intersect_hashes == 8 and not strict.pass_threshold
    assert strict.mh1_containment_in_mh2 == ds_a.contained_by(ds_b) and strict.jaccard == a.jaccard(b)
    # scaled differs: the coarser one is picked automatically (:180-268)
    a10, b1 = _pair(sm, track_abundance, {"scaled": 10}, {"scaled": 1})
    auto = FracMinHashComparison(a10, b1)
    assert auto.cmp_scaled == 10 and auto.mh2_cmp.scaled == 10 and auto.jaccard == a10.jaccard(b1.downsample(scaled=10))


def test_frac_comparison_errors(sm, track_abundance):
    # :434-537
    from sourmash_amd.sketchcomparison import FracMinHashComparison
    for a_kw, b_kw, ka, kb, err, text in (
            ({"scaled": 1}, {"scaled": 2}, 31, 21, TypeError, "Error: Cannot compare incompatible sketches."),
            ({"scaled": 1}, {"scaled": 2, "is_protein": True}, 31, 31, TypeError, "Error: Cannot compare incompatible sketches."),
            ({"scaled": 1}, {"n": 10}, 31, 31, TypeError, "Error: Both sketches must be 'num' or 'scaled'.")):
        a, b = _pair(sm, track_abundance, dict(a_kw), dict(b_kw), ka, kb)
        with pytest.raises(err) as exc:
            FracMinHashComparison(a, b)
        assert text in str(exc.value)
    a, b = _pair(sm, track_abundance, {"scaled": 1}, {"scaled": 10}, 31, 31)
    with pytest.raises(ValueError) as exc:
        FracMinHashComparison(a, b, cmp_scaled=1)
    assert "new scaled 1 is lower than current sample scaled 10" in str(exc.value)
    cmp = FracMinHashComparison(a, b)
    assert cmp.cmp_scaled == 10
    with pytest.raises(ValueError) as exc:
        cmp.downsample_and_handle_ignore_abundance()
    assert "Error: must pass in a comparison scaled or num value." in str(exc.value)


def test_num_comparison(sm, track_abundance):
    # :539-816
    from sourmash_amd.sketchcomparison import NumMinHashComparison
    a, b = _pair(sm, track_abundance, {"n": 10}, {"n": 10})
    cmp = NumMinHashComparison(a, b)
    assert cmp.cmp_num == 10 and (cmp.ksize, cmp.moltype) == (21, "DNA") and not cmp.size_may_be_inaccurate
    assert cmp.jaccard == a.jaccard(b) == b.jaccard(a)
    assert cmp.intersect_mh == a.flatten().intersection(b.flatten())
    if track_abundance:
        assert cmp.angular_similarity == a.angular_similarity(b)
    else:
        with pytest.raises(TypeError) as exc:
            cmp.angular_similarity
        assert ANGULAR in str(exc.value)
    down = NumMinHashComparison(a, b, cmp_num=5)
    assert down.mh1_cmp == a.downsample(num=5) and down.jaccard == a.downsample(num=5).jaccard(b.downsample(num=5))
    a2, b2 = _pair(sm, track_abundance, {"n": 10}, {"n": 5})
    assert NumMinHashComparison(a2, b2).cmp_num == 5                        # the smaller num wins
    a3, b3 = _pair(sm, track_abundance, {"n": 200}, {"n": 100}, 31, 31)
    with pytest.raises(ValueError) as exc:
        NumMinHashComparison(a3, b3, cmp_num=150)
    assert "new sample num is higher than current sample num" in str(exc.value)
    for a_kw, b_kw, ka, kb, text in (({"n": 10}, {"n": 10}, 31, 21, "Error: Cannot compare incompatible sketches."),
                                     ({"n": 10}, {"n": 10, "is_protein": True}, 31, 31, "Error: Cannot compare incompatible sketches."),
                                     ({"n": 10}, {"scaled": 1}, 31, 31, "Error: Both sketches must be 'num' or 'scaled'.")):
        x, y = _pair(sm, track_abundance, dict(a_kw), dict(b_kw), ka, kb)
        with pytest.raises(TypeError) as exc:
            NumMinHashComparison(x, y)
        assert text in str(exc.value)


def test_ani_estimates_and_false_negative_flag(sm):
    # :371-432, 818-1076 on the two scaled=100 genomes we carry
    from sourmash_amd.sketchcomparison import FracMinHashComparison
    a = sm.load_one_signature_from_json(golden("scaled100", "GCF_000005845.2_ASM584v2_genomic.fna.gz.sig.gz"), ksize=21).minhash
    b = sm.load_one_signature_from_json(golden("scaled100", "GCF_000006945.1_ASM694v1_genomic.fna.gz.sig.gz")).minhash
    assert a.size_is_accurate() and b.size_is_accurate()
    cmp = FracMinHashComparison(a, b)
    cmp.estimate_jaccard_ani()
    assert cmp.jaccard_ani == a.jaccard_ani(b).ani == b.jaccard_ani(a).ani
    assert cmp.potential_false_negative is False and cmp.jaccard_ani_untrustworthy == a.jaccard_ani(b).je_exceeds_threshold
    cmp.estimate_ani_from_mh1_containment_in_mh2()
    cmp.estimate_ani_from_mh2_containment_in_mh1()
    assert cmp.ani_from_mh1_containment_in_mh2 == a.containment_ani(b).ani
    assert cmp.ani_from_mh2_containment_in_mh1 == b.containment_ani(a).ani
    cmp.estimate_max_containment_ani()
    assert cmp.max_containment_ani == max(a.containment_ani(b).ani, b.containment_ani(a).ani) == a.max_containment_ani(b).ani
    assert cmp.avg_containment_ani == np.mean([a.containment_ani(b).ani, b.containment_ani(a).ani])
    assert cmp.potential_false_negative is False
    coarse = FracMinHashComparison(a, b, cmp_scaled=16000)                  # too few hashes left: may miss a real overlap
    coarse.estimate_ani_from_mh1_containment_in_mh2()
    assert coarse.potential_false_negative is True
    # a containment handed in is used as is; confidence intervals at 95 % and 99 % (:874-1019)
    given = FracMinHashComparison(a, b)
    given.estimate_ani_from_mh1_containment_in_mh2(containment=a.contained_by(b))
    assert given.ani_from_mh1_containment_in_mh2 == cmp.ani_from_mh1_containment_in_mh2
    for conf in (0.95, 0.99):
        ci = FracMinHashComparison(a, b, estimate_ani_ci=True, ani_confidence=conf)
        ci.estimate_ani_from_mh1_containment_in_mh2()
        want = a.containment_ani(b, estimate_ci=True, confidence=conf)
        assert (ci.ani_from_mh1_containment_in_mh2, ci.ani_from_mh1_containment_in_mh2_low, ci.ani_from_mh1_containment_in_mh2_high) == \
            (want.ani, want.ani_low, want.ani_high)
        ci.estimate_max_containment_ani()
        wm = a.max_containment_ani(b, estimate_ci=True, confidence=conf)
        assert (ci.max_containment_ani_low, ci.max_containment_ani_high) == (wm.ani_low, wm.ani_high)
    w95 = a.containment_ani(b, estimate_ci=True)
    w99 = a.containment_ani(b, estimate_ci=True, confidence=0.99)
    assert w99.ani_low < w95.ani_low < w95.ani < w95.ani_high < w99.ani_high
    ds = FracMinHashComparison(a.downsample(scaled=1100), b, cmp_scaled=2000)   # :1021-1076
    ds.estimate_all_containment_ani()
    assert ds.max_containment_ani == max(a.downsample(scaled=2000).containment_ani(b.downsample(scaled=2000)).ani,
                                         b.downsample(scaled=2000).containment_ani(a.downsample(scaled=2000)).ani)
