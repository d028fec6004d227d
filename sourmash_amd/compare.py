"""All-vs-all comparison of signatures -- one batched GPU call.

API of src/sourmash/compare.py (compare_serial :14-64, compare_serial_containment
:67-108, compare_serial_max_containment :111-150, compare_serial_avg_containment
:153-187, compare_parallel :241-325, compare_all_pairs :328-358).  The reference
walks the N(N-1)/2 pairs in Python, cloning two sketches through the FFI per
pair; here the sketches are packed once into a CSR, the compare kernels
(csrc/compare.hip, bitindex.hip, sparse_pairs.hip) return the u32 common-hash matrix,
and Jaccard / containment / ANI are derived from it on whole arrays:
    jaccard[i][j]      = common / max(1, n_i + n_j - common)     (one IEEE divide, on the GPU)
    containment[i][j]  = debias(common, n_j)  with the host formula of minhash.py:819-841
    ani[i][j]          = 1 - (1 - containment^(1/k))             (host libm pow, distance_utils.py:276-283)
Abundance-weighted (angular) pairs, num sketches and collections with mixed scaled
values go through the per-pair GPU entry points, like the reference's loop.
"""
import ctypes as C
import itertools

import numpy as np

from ._lowlevel import lib
from .utils import rustcall

__all__ = ["compare_all_pairs", "compare_serial", "compare_parallel", "compare_serial_containment",
           "compare_serial_max_containment", "compare_serial_avg_containment", "common_matrix"]


def _uniform(mhs):
    "every sketch scaled, none bottom-k, one scaled value: the batched kernels apply as they are"
    return bool(mhs) and not any(mh.num for mh in mhs) and all(mh.scaled for mh in mhs) and len({mh.scaled for mh in mhs}) == 1


def common_matrix(mhs, want_jaccard=True):
    """u32 common[n][n] (+ f64 jaccard[n][n]) of flat scaled sketches: one GPU call
    (smgpu_compare_all_pairs).  Raises the compatibility error of the first mismatch."""
    n = len(mhs)
    common = np.zeros((n, n), dtype=np.uint32)
    jac = np.zeros((n, n), dtype=np.float64) if want_jaccard else None
    if n == 0:
        return common, jac
    ptrs = (C.c_void_p * n)(*[mh._get_objptr() for mh in mhs])
    rustcall(lib.smgpu_compare_all_pairs, ptrs, n, common.ctypes.data_as(C.POINTER(C.c_uint32)),
             jac.ctypes.data_as(C.POINTER(C.c_double)) if want_jaccard else None)
    return common, jac


def _pow(x, y):
    """x ** y element by element through the host libm's pow() -- the function CPython's float `**` ends in, so the
    bits equal the reference's per-pair Python arithmetic (NumPy's own vectorised pow may differ in the last place).
    y: scalar or array like x.  Threads split the array; the host float layer stays on the host."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    if np.ndim(y) == 0:
        yy, ny = np.array([y], dtype=np.float64), 1
    else:
        yy = np.ascontiguousarray(y, dtype=np.float64)
        ny = yy.size
        assert ny == x.size
    lib.smgpu_host_pow_f64(x.ctypes.data_as(C.c_void_p), yy.ctypes.data_as(C.c_void_p), ny,
                           out.ctypes.data_as(C.c_void_p), x.size, 0)
    return out


def compare_serial(siglist, ignore_abundance, *, downsample=False, return_ani=False):
    """Similarity matrix (compare.py:14-64): Jaccard, or angular similarity for the pairs whose two sketches both track
    abundance when it is not ignored."""
    n = len(siglist)
    mhs = [s.minhash for s in siglist]
    if return_ani or not _uniform(mhs):
        # per-pair semantics as they are: bottom-k sketches, mixed scaled values (each pair is compared at ITS coarser
        # scaled when downsample is set, and fails with MismatchScaled otherwise), Jaccard ANI
        sims = np.ones((n, n))
        for i, j in itertools.combinations(range(n), 2):
            if return_ani:
                ani = siglist[i].jaccard_ani(siglist[j], downsample=downsample).ani
                sims[i][j] = sims[j][i] = 0.0 if ani is None else ani
            else:
                sims[i][j] = sims[j][i] = siglist[i].similarity(siglist[j], ignore_abundance=ignore_abundance,
                                                                downsample=downsample)
        return sims
    weighted = [] if ignore_abundance else [i for i, mh in enumerate(mhs) if mh.track_abundance]
    if len(weighted) == n and n > 1:
        sims = np.ones((n, n))
    else:
        _, sims = common_matrix([mh.flatten() for mh in mhs], want_jaccard=True)
    for i, j in itertools.combinations(weighted, 2):           # minhash.rs:682-702 decides per pair
        sims[i][j] = sims[j][i] = siglist[i].similarity(siglist[j], ignore_abundance=False, downsample=downsample)
    return sims


def _bias_factors(sizes, scaled):
    "1 - (1 - 1/scaled)^(denom * scaled) for every denominator (minhash.py:832-834); Python floats, libm pow"
    return np.array([1.0 - (1.0 - 1.0 / scaled) ** float(int(d) * scaled) if d else 1.0 for d in sizes], dtype=np.float64)


def _debias(common, denom, bias):
    "count / (denom * bias_factor) clamped to [0, 1], 0 for an empty denominator (minhash.py:827-841); arrays broadcast"
    with np.errstate(divide="ignore", invalid="ignore"):
        c = common / (denom * bias)
    c = np.where(c >= 1, 1.0, np.where(c <= 0, 0.0, c))
    return np.where(denom == 0, 0.0, c)


def _debias_matrix(common, sizes, scaled, mode):
    """containment matrices from the common matrix with the reference's host arithmetic
    (src/sourmash/minhash.py:819-841,881-905,946-959), in its operation order: n pow calls for the bias factors, then
    IEEE multiplies and divides on whole arrays (exactly rounded, so the bits equal the per-pair Python floats)."""
    n = len(sizes)
    sz = np.asarray(sizes, dtype=np.float64)
    bias = _bias_factors(sizes, scaled)
    cm = np.asarray(common, dtype=np.float64)
    if mode == "containment":            # [i][j] = siglist[j].contained_by(siglist[i]): the denominator is |j|
        out = _debias(cm, sz[None, :], bias[None, :])
    elif mode == "max":                  # denominator min(|i|, |j|), its bias factor
        small_j = sz[None, :] <= sz[:, None]
        out = _debias(cm, np.where(small_j, sz[None, :], sz[:, None]), np.where(small_j, bias[None, :], bias[:, None]))
    else:                                # mean of the two directed containments
        out = (_debias(cm, sz[None, :], bias[None, :]) + _debias(cm, sz[:, None], bias[:, None])) / 2
    out[np.arange(n), np.arange(n)] = 1.0
    return out


def _ani_from_containment(cont, ksize):
    """1 - distance with distance = 1 - containment^(1/ksize), 1 for containment 0, 0 for containment 1
    (distance_utils.py:276-283 point estimate, ANIResult.ani = 1 - dist)"""
    point = 1.0 - _pow(cont, 1.0 / ksize)
    point = np.where(cont == 0, 1.0, np.where(cont == 1, 0.0, point))
    return 1 - point


def _containment(siglist, downsample, mode, return_ani):
    n = len(siglist)
    mhs = [s.minhash for s in siglist]
    if not all(mh.scaled for mh in mhs):
        raise TypeError("Error: can only calculate %s for scaled MinHashes" % ("ANI" if return_ani else "containment"))
    if not _uniform(mhs):
        return _containment_pairs(siglist, downsample, mode, return_ani)
    flat = [mh.flatten() for mh in mhs]
    common, _ = common_matrix(flat, want_jaccard=False)
    sizes = [len(mh) for mh in flat]
    scaled, ksize = flat[0].scaled, flat[0].ksize
    if not return_ani:
        return _debias_matrix(common, sizes, scaled, mode)
    # ANI (compare.py:67-187): the containment of every entry -> point estimate; an estimate is withheld (0 in the
    # matrix) when either sketch is too small for its size to be trusted (minhash.py:869-871)
    if mode == "avg":
        sz = np.asarray(sizes, dtype=np.float64)
        bias = _bias_factors(sizes, scaled)
        cm = np.asarray(common, dtype=np.float64)
        a1 = _ani_from_containment(_debias(cm, sz[None, :], bias[None, :]), ksize).reshape(n, n)
        a2 = _ani_from_containment(_debias(cm, sz[:, None], bias[:, None]), ksize).reshape(n, n)
        out = (a1 + a2) / 2
    else:
        out = _ani_from_containment(_debias_matrix(common, sizes, scaled, mode), ksize).reshape(n, n)
    trusted = np.array([mh.size_is_accurate() for mh in mhs], dtype=bool)
    out = np.where(trusted[:, None] & trusted[None, :], out, 0.0)
    out[np.arange(n), np.arange(n)] = 1.0
    return out


def _containment_pairs(siglist, downsample, mode, return_ani):
    "the reference's loops pair by pair (mixed scaled values: every pair at its own coarser scaled)"
    from .sketchcomparison import FracMinHashComparison
    n = len(siglist)
    out = np.ones((n, n))
    for i in range(n):
        for j in range(n):
            if i == j or (mode != "containment" and j < i):
                continue
            if mode == "containment":
                v = siglist[j].containment_ani(siglist[i], downsample=downsample).ani if return_ani \
                    else siglist[j].contained_by(siglist[i], downsample=downsample)
                out[i][j] = 0.0 if v is None else v
                continue
            if mode == "max":
                v = siglist[j].max_containment_ani(siglist[i], downsample=downsample).ani if return_ani \
                    else siglist[j].max_containment(siglist[i], downsample=downsample)
            else:
                v = FracMinHashComparison(siglist[j].minhash, siglist[i].minhash).avg_containment_ani if return_ani \
                    else siglist[j].avg_containment(siglist[i], downsample=downsample)
            out[i][j] = out[j][i] = 0.0 if v is None else v
    return out


def compare_serial_containment(siglist, *, downsample=False, return_ani=False):
    return _containment(siglist, downsample, "containment", return_ani)


def compare_serial_max_containment(siglist, *, downsample=False, return_ani=False):
    return _containment(siglist, downsample, "max", return_ani)


def compare_serial_avg_containment(siglist, *, downsample=False, return_ani=False):
    return _containment(siglist, downsample, "avg", return_ani)


def compare_parallel(siglist, ignore_abundance, *, downsample, n_jobs, return_ani=False):
    "n_jobs is accepted for API compatibility; the GPU call is already all-pairs."
    return compare_serial(siglist, ignore_abundance, downsample=downsample, return_ani=return_ani)


def compare_all_pairs(siglist, ignore_abundance, *, downsample=False, n_jobs=None, return_ani=False):
    "compare.py:328-358"
    return compare_serial(siglist, ignore_abundance, downsample=downsample, return_ani=return_ani)
