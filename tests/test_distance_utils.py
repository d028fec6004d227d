"""Host float layer against the reference's known answers (tests/test_distance_utils.py values)."""
import numpy as np
import pytest

from sourmash_amd.distance_utils import (ANIResult, ciANIResult, containment_to_distance, get_exp_probability_nothing_common,
                                          handle_seqlen_nkmers, jaccard_to_distance, jaccardANIResult, set_size_chernoff,
                                          set_size_exact_prob, var_n_mutated)


def test_result_classes():
    assert ANIResult(0.4, 0.1).ani == 0.6 and ANIResult(0.4, 0.1).p_exceeds_threshold
    for bad in (1.1, -0.1):
        with pytest.raises(ValueError):
            ANIResult(bad, 0.1)
    assert jaccardANIResult(0.4, 0.1, jaccard_error=0.03).ani is None
    assert jaccardANIResult(0.4, 0.1, jaccard_error=0.03, je_threshold=0.1).ani == 0.6
    with pytest.raises(ValueError):
        jaccardANIResult(0.4, 0.1, None)
    r = ciANIResult(0.4, 0.1, dist_low=0.3, dist_high=0.5)
    assert (r.ani, r.ani_low, r.ani_high) == (0.6, 0.5, 0.7)


def test_containment_to_distance_kats():
    # reference tests/test_distance_utils.py:144-260
    r = containment_to_distance(0.5, 21, 1, n_unique_kmers=10000, estimate_ci=True)
    assert (r.dist, r.ani) == (0.032468221476108394, 0.9675317785238916)
    assert (r.dist_low, r.dist_high) == (0.028709912966405623, 0.03647860197289783)
    assert (r.ani_high, r.ani_low, r.p_nothing_in_common) == (0.9712900870335944, 0.9635213980271021, 0.0)
    r = containment_to_distance(0.1, 31, 100, n_unique_kmers=10000, estimate_ci=True)
    assert (r.dist, r.dist_low, r.dist_high) == (0.07158545548052564, 0.05320779238601372, 0.09055547672455365)
    assert r.p_nothing_in_common == 4.3171247410658655e-05 and not r.p_exceeds_threshold
    r = containment_to_distance(0.5, 21, 100, n_unique_kmers=10000, estimate_ci=True)
    assert (r.dist_low, r.dist_high) == (0.023712063916639017, 0.04309960543965866)
    r = containment_to_distance(0.5, 10, 100, n_unique_kmers=10000, estimate_ci=True)
    assert (r.dist, r.dist_low, r.dist_high) == (0.06696700846319259, 0.04982777541057476, 0.08745108232411622)
    r = containment_to_distance(0.1, 31, 100, confidence=0.99, n_unique_kmers=10000, estimate_ci=True)
    assert (r.dist_low, r.dist_high) == (0.04802880300938562, 0.09619930040790341)
    assert containment_to_distance(0, 21, 1, n_unique_kmers=10000).dist == 1.0
    z = containment_to_distance(1, 21, 1, n_unique_kmers=10000, estimate_ci=True)
    assert (z.dist, z.ani_low, z.ani_high) == (0.0, 1.0, 1.0)


def test_jaccard_to_distance_kats():
    # reference tests/test_distance_utils.py:322-375
    r = jaccard_to_distance(0.5, 21, 1, n_unique_kmers=10000)
    assert round(r.dist, 3) == round(0.019122659390482077, 3) and r.ani is None
    assert r.jaccard_error == 0.00018351337045518042 and r.je_exceeds_threshold
    r2 = jaccard_to_distance(0.5, 31, 100, n_unique_kmers=10000, err_threshold=0.1)
    assert r2.ani == 0.9870056455892898
    assert jaccard_to_distance(0.1, 31, 100, n_unique_kmers=10000).ani == 0.9464928391768298
    assert jaccard_to_distance(0, 31, 100, n_unique_kmers=10000).dist == 1.0
    assert jaccard_to_distance(1, 31, 100, n_unique_kmers=10000).dist == 0.0


def test_probabilities():
    n = handle_seqlen_nkmers(31, sequence_len_bp=1000030)
    assert get_exp_probability_nothing_common(0.25, 31, 10, n_unique_kmers=n) == 7.437016945722123e-07
    assert abs(set_size_chernoff(1000000, 10, relative_error=0.01) - 0.928652) < 1e-6
    assert abs(set_size_chernoff(10000, 1, relative_error=0.05) - 0.999519) < 1e-6
    for (n, s, e), want in (((100, 2, 0.05), 0.382701), ((200, 5, 0.15), 0.749858), ((10, 10, 0.10), 0.38742),
                            ((1000, 10, 0.10), 0.73182)):
        np.testing.assert_array_almost_equal(want, set_size_exact_prob(n, s, relative_error=e), decimal=3)
    assert var_n_mutated(100, 21, 0) == 0.0
    with pytest.raises(ValueError):
        handle_seqlen_nkmers(31)
