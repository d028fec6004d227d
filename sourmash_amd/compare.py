"""All-vs-all comparison of signatures -- one batched GPU call.

API of src/sourmash/compare.py (compare_serial :14-64, compare_serial_containment
:67-108, compare_serial_max_containment :111-150, compare_serial_avg_containment
:153-187, compare_parallel :241-325, compare_all_pairs :328-358).  The reference
walks the N(N-1)/2 pairs in Python, cloning two sketches through the FFI per
pair; here the sketches are packed once into a CSR, the LDS-tiled merge kernel
(csrc/compare.hip) returns the u32 common-hash matrix, and Jaccard / containment
are derived from it:
    jaccard[i][j]      = common / max(1, n_i + n_j - common)     (one IEEE divide, on the GPU)
    containment[i][j]  = debias(common, n_j)  with the host formula of minhash.py:819-841
Abundance-weighted (angular) comparison and num sketches go through the per-pair
GPU entry points, like the reference's loop.
"""
import ctypes as C
import itertools

import numpy as np

from ._lowlevel import lib
from .utils import rustcall

__all__ = ["compare_all_pairs", "compare_serial", "compare_parallel", "compare_serial_containment",
           "compare_serial_max_containment", "compare_serial_avg_containment", "common_matrix"]


def _flat_scaled_minhashes(siglist, downsample):
    "-> list of flat MinHash at one scaled (or raises like the reference would)."
    mhs = [s.minhash for s in siglist]
    if not mhs:
        return mhs
    scaleds = {mh.scaled for mh in mhs}
    if downsample and len(scaleds) > 1 and all(scaleds):
        mx = max(scaleds)
        mhs = [mh.downsample(scaled=mx) for mh in mhs]
    return [mh.flatten() for mh in mhs]


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


def _batchable(siglist, ignore_abundance):
    mhs = [s.minhash for s in siglist]
    if not mhs or any(mh.num for mh in mhs) or not all(mh.scaled for mh in mhs):
        return False
    if not ignore_abundance and all(mh.track_abundance for mh in mhs):
        return False            # angular similarity: per-pair path
    return True


def compare_serial(siglist, ignore_abundance, *, downsample=False, return_ani=False):
    "Similarity matrix (Jaccard, or angular when every sketch tracks abundance and it is not ignored)."
    n = len(siglist)
    if _batchable(siglist, ignore_abundance) and not return_ani:
        mhs = _flat_scaled_minhashes(siglist, downsample)
        _, jac = common_matrix(mhs, want_jaccard=True)
        return jac
    sims = np.ones((n, n))
    for i, j in itertools.combinations(range(n), 2):
        if return_ani:
            ani = siglist[i].jaccard_ani(siglist[j], downsample=downsample).ani
            sims[i][j] = sims[j][i] = 0.0 if ani is None else ani
        else:
            sims[i][j] = sims[j][i] = siglist[i].similarity(siglist[j], ignore_abundance=ignore_abundance,
                                                            downsample=downsample)
    return sims


def _debias_matrix(common, sizes, scaled, mode):
    """containment matrices from the common matrix with the reference's host math
    (src/sourmash/minhash.py:819-841,881-905): count / (denom * (1 - (1 - 1/scaled)**(denom*scaled))),
    clamped to [0, 1]; Python floats so `**` is the same libm pow the reference uses."""
    n = len(sizes)
    out = np.ones((n, n))

    def debias(count, denom):
        if not denom:
            return 0.0
        bias = 1.0 - (1.0 - 1.0 / scaled) ** float(denom * scaled)
        c = count / (denom * bias)
        return 1.0 if c >= 1 else 0.0 if c <= 0 else c

    for i in range(n):
        for j in range(n):
            if i == j:
                continue
            c = int(common[i, j])
            if mode == "containment":        # [i][j] = siglist[j].contained_by(siglist[i])
                out[i, j] = debias(c, sizes[j])
            elif mode == "max":
                out[i, j] = debias(c, min(sizes[i], sizes[j]))
            else:                            # avg of the two directed containments
                out[i, j] = (debias(c, sizes[j]) + debias(c, sizes[i])) / 2
    return out


def _containment(siglist, downsample, mode, return_ani):
    if return_ani:
        return _containment_ani(siglist, downsample, mode)
    mhs = [s.minhash for s in siglist]
    if not all(mh.scaled for mh in mhs):
        raise TypeError("Error: can only calculate containment for scaled MinHashes")
    mhs = _flat_scaled_minhashes(siglist, downsample)
    common, _ = common_matrix(mhs, want_jaccard=False)
    sizes = [len(mh) for mh in mhs]
    return _debias_matrix(common, sizes, mhs[0].scaled if mhs else 1, mode)


def _containment_ani(siglist, downsample, mode):
    """ANI matrices from the containment family (compare.py:67-180 of the reference): the counts behind every entry
    are GPU intersections, the ANI point estimates host floats per pair; a missing estimate is reported as 0."""
    from .sketchcomparison import FracMinHashComparison
    n = len(siglist)
    out = np.ones((n, n))
    for i in range(n):
        for j in range(n):
            if i == j or (mode != "containment" and j < i):
                continue
            if mode == "containment":
                ani = siglist[j].containment_ani(siglist[i], downsample=downsample).ani
                out[i][j] = 0.0 if ani is None else ani
                continue
            if mode == "max":
                ani = siglist[j].max_containment_ani(siglist[i], downsample=downsample).ani
            else:
                ani = FracMinHashComparison(siglist[j].minhash, siglist[i].minhash).avg_containment_ani
            out[i][j] = out[j][i] = 0.0 if ani is None else ani
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
