"""Host float layer: Jaccard / containment -> evolutionary distance (1 - ANI).

Counterpart of src/sourmash/distance_utils.py (:17-407).  The formulas are those of Hera, Pierce-Ward
& Koslicki, "Deriving confidence intervals for mutation rates across a wide range of evolutionary
distances using FracMinHash" (doi:10.1101/2022.01.11.475870), evaluated in double precision on the
host from the integer intersection counts the GPU kernels produce; scipy supplies brentq / norm.ppf /
binom exactly as in the reference, so results agree to the last bit on the same libm.
"""
from dataclasses import dataclass, field
from math import exp, log

import numpy as np

__all__ = ["ANIResult", "jaccardANIResult", "ciANIResult", "containment_to_distance", "jaccard_to_distance",
           "set_size_chernoff", "set_size_exact_prob", "get_exp_probability_nothing_common",
           "handle_seqlen_nkmers", "var_n_mutated", "exp_n_mutated", "r1_to_q"]


def check_distance(dist):
    if not 0 <= dist <= 1:
        raise ValueError(f"Error: distance value {dist :.4f} is not between 0 and 1!")
    return dist


def check_prob_threshold(val, threshold=1e-3):
    "probability of sharing no hash by chance alone (false negative); flag when above threshold"
    return val, bool(threshold is not None and val > threshold)


def check_jaccard_error(val, threshold=1e-4):
    return val, bool(threshold is not None and val > threshold)


@dataclass
class ANIResult:
    "distance / ANI estimate from k-mer containment"
    dist: float
    p_nothing_in_common: float
    p_threshold: float = 1e-3
    size_is_inaccurate: bool = False
    p_exceeds_threshold: bool = field(init=False)

    def check_dist_and_p_threshold(self):
        self.dist = check_distance(self.dist)
        self.p_nothing_in_common, self.p_exceeds_threshold = check_prob_threshold(self.p_nothing_in_common,
                                                                                  self.p_threshold)

    def __post_init__(self):
        self.check_dist_and_p_threshold()

    @property
    def ani(self):
        return None if self.size_is_inaccurate else 1 - self.dist


@dataclass
class jaccardANIResult(ANIResult):
    "…from Jaccard: carries a lower bound of the approximation error"
    jaccard_error: float = None
    je_threshold: float = 1e-4

    def __post_init__(self):
        self.check_dist_and_p_threshold()
        if self.jaccard_error is None:
            raise ValueError("Error: jaccard_error cannot be None.")
        self.jaccard_error, self.je_exceeds_threshold = check_jaccard_error(self.jaccard_error, self.je_threshold)

    @property
    def ani(self):
        if self.je_exceeds_threshold or self.size_is_inaccurate:
            return None
        return 1 - self.dist


@dataclass
class ciANIResult(ANIResult):
    "…from containment, optionally with a confidence interval"
    dist_low: float = None
    dist_high: float = None

    def __post_init__(self):
        self.check_dist_and_p_threshold()
        if self.dist_low is not None and self.dist_high is not None:
            self.dist_low = check_distance(self.dist_low)
            self.dist_high = check_distance(self.dist_high)

    @property
    def ani_low(self):
        if self.dist_high is None or self.size_is_inaccurate:
            return None
        return 1 - self.dist_high

    @property
    def ani_high(self):
        if self.dist_low is None or self.size_is_inaccurate:
            return None
        return 1 - self.dist_low


def r1_to_q(k, r1):
    "probability that a k-mer contains at least one mutation at per-base rate r1"
    r1 = float(r1)
    return float(1 - (1 - r1) ** k)


def var_n_mutated(L, k, r1, *, q=None):
    "variance of the number of mutated k-mers among L (distance_utils.py:133-153)"
    if r1 == 0:
        return 0.0
    r1 = float(r1)
    if q is None:
        q = r1_to_q(k, r1)
    varN = (L * (1 - q) * (q * (2 * k + (2 / r1) - 1) - 2 * k)
            + k * (k - 1) * (1 - q) ** 2
            + (2 * (1 - q) / (r1 ** 2)) * ((1 + (k - 1) * (1 - q)) * r1 - q))
    if varN < 0.0:
        raise ValueError("Error: varN <0.0!")
    return float(varN)


def exp_n_mutated(L, k, r1):
    return L * r1_to_q(k, r1)


def exp_n_mutated_squared(L, k, p):
    return var_n_mutated(L, k, p) + exp_n_mutated(L, k, p) ** 2


def probit(p):
    from scipy.stats import norm
    return norm.ppf(p)


def handle_seqlen_nkmers(ksize, *, sequence_len_bp=None, n_unique_kmers=None):
    if n_unique_kmers is not None:
        return n_unique_kmers
    if sequence_len_bp is None:
        raise ValueError("Error: distance estimation requires input of either 'sequence_len_bp' or 'n_unique_kmers'")
    return sequence_len_bp - (ksize - 1)


def set_size_chernoff(set_size, scaled, *, relative_error=0.05):
    "two-sided Chernoff bound on P(|sketch_size*scaled - set_size| <= relative_error*set_size)"
    return 1 - 2 * np.exp(-(relative_error ** 2) * set_size / (scaled * 3))


def set_size_exact_prob(set_size, scaled, *, relative_error=0.05):
    "the same probability from the binomial CDF (sketch size ~ Binomial(set_size, 1/scaled))"
    from scipy.stats import binom
    lo_arg = -set_size / scaled * (relative_error - 1)
    hi_arg = set_size / scaled * (relative_error + 1)
    prob = binom.cdf(hi_arg, set_size, 1 / scaled) - binom.cdf(lo_arg, set_size, 1 / scaled)
    if lo_arg == int(lo_arg):            # include the lower edge when it is an attainable count
        prob = prob + binom.pmf(lo_arg, set_size, 1 / scaled)
    return prob


def get_expected_log_probability(n_unique_kmers, ksize, mutation_rate, scaled_fraction):
    exp_nmut = exp_n_mutated(n_unique_kmers, ksize, mutation_rate)
    try:
        return (n_unique_kmers - exp_nmut) * log(1.0 - scaled_fraction)
    except Exception:
        return float("-inf")


def get_exp_probability_nothing_common(mutation_rate, ksize, scaled, *, n_unique_kmers=None, sequence_len_bp=None):
    "expected probability that two FracMinHash sketches of sequences at this distance share no hash"
    n_unique_kmers = handle_seqlen_nkmers(ksize, sequence_len_bp=sequence_len_bp, n_unique_kmers=n_unique_kmers)
    if mutation_rate == 1.0:
        return 1.0
    if mutation_rate == 0.0:
        return 0.0
    return exp(get_expected_log_probability(n_unique_kmers, ksize, mutation_rate, 1.0 / float(scaled)))


def containment_to_distance(containment, ksize, scaled, *, n_unique_kmers=None, sequence_len_bp=None,
                            confidence=0.95, estimate_ci=False, prob_threshold=1e-3):
    "containment -> distance point estimate (1 - C^(1/k)) with an optional confidence interval"
    n = handle_seqlen_nkmers(ksize, sequence_len_bp=sequence_len_bp, n_unique_kmers=n_unique_kmers)
    sol_hi = sol_lo = None
    if containment == 0:
        point = sol_hi = sol_lo = 1.0
    elif containment == 1:
        point = sol_hi = sol_lo = 0.0
    else:
        point = 1.0 - containment ** (1.0 / ksize)
        if estimate_ci:
            from scipy.optimize import brentq
            try:
                z_alpha = probit(1 - (1 - confidence) / 2)
                f_scaled = 1.0 / scaled
                bias_factor = 1 - (1 - f_scaled) ** n
                term_1 = (1.0 - f_scaled) / (f_scaled * n ** 3 * bias_factor ** 2)

                def var_direct(p):
                    term_2 = n * exp_n_mutated(n, ksize, p) - exp_n_mutated_squared(n, ksize, p)
                    term_3 = var_n_mutated(n, ksize, p) / n ** 2
                    return term_1 * term_2 + term_3

                def upper(p):
                    return (1 - p) ** ksize + z_alpha * np.sqrt(var_direct(p)) - containment

                def lower(p):
                    return (1 - p) ** ksize - z_alpha * np.sqrt(var_direct(p)) - containment

                sol_hi = brentq(upper, 0.0000001, 0.9999999)
                sol_lo = brentq(lower, 0.0000001, 0.9999999)
            except ValueError:
                # tiny inputs: no sign change / negative variance -> no interval (the reference warns and continues)
                sol_hi = sol_lo = None
    p_nothing = get_exp_probability_nothing_common(point, ksize, scaled, n_unique_kmers=n)
    return ciANIResult(point, p_nothing, dist_low=sol_lo, dist_high=sol_hi, p_threshold=prob_threshold)


def jaccard_to_distance(jaccard, ksize, scaled, *, n_unique_kmers=None, sequence_len_bp=None, prob_threshold=1e-3,
                        err_threshold=1e-4):
    "Jaccard -> distance point estimate 1 - (2J/(1+J))^(1/k) and a lower bound of its approximation error"
    n = handle_seqlen_nkmers(ksize, sequence_len_bp=sequence_len_bp, n_unique_kmers=n_unique_kmers)
    if jaccard == 0:
        point, err = 1.0, 0.0
    elif jaccard == 1:
        point, err = 0.0, 0.0
    else:
        point = 1.0 - (2.0 * jaccard / float(1 + jaccard)) ** (1.0 / float(ksize))
        exp_n_mut = exp_n_mutated(n, ksize, point)
        var_n_mut = var_n_mutated(n, ksize, point)
        err = 1.0 * n * var_n_mut / (n + exp_n_mut) ** 3
    p_nothing = get_exp_probability_nothing_common(point, ksize, scaled, n_unique_kmers=n)
    return jaccardANIResult(point, p_nothing, jaccard_error=err, p_threshold=prob_threshold, je_threshold=err_threshold)
