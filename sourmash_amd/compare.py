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
Bottom-k (num) sketches and abundance-weighted (angular) similarity have tile kernels of their own (csrc/compare_ext.hip);
collections with several scaled values are served one launch per value (every pair at ITS coarser scaled, like
similarity(downsample=True)); Jaccard- and containment-derived ANI and the containment matrices of mixed-scaled lists are
whole-array arithmetic on count matrices.  The per-pair loop remains only as the way a list with an incompatible pair raises
the reference's error from the reference's pair.
"""
import ctypes as C
import itertools

import numpy as np

from ._lowlevel import lib
from .utils import objptr_array, rustcall

__all__ = ["compare_all_pairs", "compare_serial", "compare_parallel", "compare_serial_containment",
           "compare_serial_max_containment", "compare_serial_avg_containment", "common_matrix", "num_matrix", "angular_matrix",
           "jaccard_ani_values"]


class _Views:
    """The first sketches of a list of signatures as BORROWED handles plus their parameters as arrays, from ONE C call
    (smgpu_signatures_sketch_views).  The reference's loop reads `sig.minhash` per pair, and that property clones the sketch
    through the FFI (signature.rs:167-182): for 10,000 signatures 400 MB of copies and a dozen FFI calls per sketch before any
    comparing starts.  The views keep the signature objects alive; nothing here is freed or modified."""

    def __init__(self, siglist=None, *, _from=None, _idx=None):
        from .minhash import _get_scaled_for_max_hash
        if _from is not None:                                      # a subset of another view (shares the signatures)
            idx = np.asarray(_idx, dtype=np.int64)
            self.sigs = [_from.sigs[i] for i in idx]
            self.n = len(idx)
            self.ptrs = (C.c_void_p * max(self.n, 1))(*[_from.ptrs[i] for i in idx])
            self.params = _from.params[idx].copy()
        else:
            self.sigs = list(siglist)
            self.n = n = len(self.sigs)
            self.ptrs = (C.c_void_p * max(n, 1))()
            self.params = np.zeros((max(n, 1), 8), dtype=np.uint64)
            if n:
                sp, _alive = objptr_array(self.sigs)
                rustcall(lib.smgpu_signatures_sketch_views, sp, n, self.ptrs, self.params.ctypes.data_as(C.c_void_p))
            self.params = self.params[:n]
        p = self.params
        self.ksize_raw, self.hf, self.seed, self.max_hash, self.num = p[:, 0], p[:, 1], p[:, 2], p[:, 3], p[:, 4]
        self.abund = p[:, 5] != 0
        self.size = p[:, 6].astype(np.int64)
        self.one_max_hash = bool(self.n == 0 or (self.max_hash == self.max_hash[0]).all())
        if self.n and self.one_max_hash:                             # the usual list: one scaled value
            m0 = int(self.max_hash[0])
            self.scaled = np.full(self.n, _get_scaled_for_max_hash(m0) if m0 else 0, dtype=np.int64)
        else:
            per = {int(m): (_get_scaled_for_max_hash(int(m)) if m else 0) for m in np.unique(self.max_hash)}
            self.scaled = np.array([per[int(m)] for m in self.max_hash], dtype=np.int64)

    @property
    def ksize(self):
        "MinHash.ksize of the first sketch (residues for protein / dayhoff / hp)"
        k = int(self.ksize_raw[0])
        return k if int(self.hf[0]) == 1 else k // 3

    def subset(self, idx):
        return _Views(_from=self, _idx=idx)

    def minhashes(self):
        "sketch OBJECTS (the cloning path): only for the corners that need host-side set operations"
        return [s.minhash for s in self.sigs]


class _Handles:
    "the same for a list of MinHash objects (common_matrix / num_matrix / angular_matrix as public helpers)"

    def __init__(self, mhs):
        self.keep = list(mhs)
        self.n = len(self.keep)
        self.ptrs, self._ptrs_alive = objptr_array(self.keep)


_PINNED_FROM = 32 << 20         # result arrays from this size up live in page-locked memory (SMG_PINNED_RESULTS=0: never)


class _PinnedBlock:
    "owner of one smgpu_host_alloc block: released (back to the library's cache) when the array built on it goes"

    def __init__(self, nbytes):
        self.ptr = rustcall(lib.smgpu_host_alloc, nbytes)
        self.buf = (C.c_char * nbytes).from_address(self.ptr)

    def __del__(self):
        ptr, self.ptr = getattr(self, "ptr", None), None
        if ptr:
            lib.smgpu_host_free(ptr)


def _result_array(shape, dtype):
    """np.empty for a result the library is about to fill.  Large ones are placed in page-locked memory: the device-to-host copy of
    an 800 MB matrix into fresh pageable memory runs at ~38 GB/s on the transfer ring (page faults under the copy-out threads),
    into page-locked memory at the link's ~55 GB/s with no threads at all; the block returns to the library's cache when the
    array is collected, so the next call of the same size pays nothing for it."""
    import os
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    if nbytes < _PINNED_FROM or os.environ.get("SMG_PINNED_RESULTS") == "0":
        return np.empty(shape, dtype=dtype)
    try:
        block = _PinnedBlock(nbytes)
    except Exception:                                             # noqa: BLE001 -- no page-locked memory to be had: pageable will do
        return np.empty(shape, dtype=dtype)
    arr = np.frombuffer(block.buf, dtype=dtype).reshape(shape)
    block.buf._owner = block                                      # the array keeps the ctypes buffer alive, the buffer its owner
    del block.buf                                                 # (no cycle: the owner must not hold the buffer)
    return arr


def _common_ptrs(ptrs, n, want_jaccard=True, want_common=True):
    "only the matrices asked for cross PCIe (the u32 matrix of 10,000 sketches is 400 MB, the f64 one 800 MB)"
    common = _result_array((n, n), np.uint32) if want_common else None
    jac = _result_array((n, n), np.float64) if want_jaccard else None
    if n:
        rustcall(lib.smgpu_compare_all_pairs, ptrs, n, common.ctypes.data_as(C.POINTER(C.c_uint32)) if want_common else None,
                 jac.ctypes.data_as(C.POINTER(C.c_double)) if want_jaccard else None)
    return common, jac


def common_matrix(mhs, want_jaccard=True):
    """u32 common[n][n] (+ f64 jaccard[n][n]) of flat scaled sketches: one GPU call
    (smgpu_compare_all_pairs).  Raises the compatibility error of the first mismatch."""
    h = _Handles(mhs)
    return _common_ptrs(h.ptrs, h.n, want_jaccard)


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


def _num_ptrs(ptrs, n, want_counts=False):
    jac = np.ones((n, n), dtype=np.float64)
    common = np.zeros((n, n), dtype=np.uint32) if want_counts else None
    union = np.zeros((n, n), dtype=np.uint32) if want_counts else None
    if n:
        rustcall(lib.smgpu_compare_num_all_pairs, ptrs, n,
                 common.ctypes.data_as(C.POINTER(C.c_uint32)) if want_counts else None,
                 union.ctypes.data_as(C.POINTER(C.c_uint32)) if want_counts else None,
                 jac.ctypes.data_as(C.POINTER(C.c_double)))
    return (jac, common, union) if want_counts else jac


def num_matrix(mhs, want_counts=False):
    """Jaccard of every pair of flat BOTTOM-K sketches in one launch (smgpu_compare_num_all_pairs, csrc/compare_ext.hip): the
    intersection is taken against the merged sketch truncated to `num` (minhash.rs:593-621), num being that of the sketch
    with the lower index, as in compare.py:39 `siglist[i].similarity(siglist[j])`.  -> f64 [n][n] (and u32 common, union)."""
    h = _Handles(mhs)
    return _num_ptrs(h.ptrs, h.n, want_counts)


def _angular_ptrs(ptrs, n):
    sims = np.ones((n, n), dtype=np.float64)
    if n:
        rustcall(lib.smgpu_compare_angular_all_pairs, ptrs, n, sims.ctypes.data_as(C.POINTER(C.c_double)), None, None)
    return sims


def angular_matrix(mhs):
    """Angular similarity of every pair of abundance-tracking sketches (minhash.rs:635-680): the integer sums -- sum over
    the common hashes of abund x abund, per sketch the sum of squares -- from one launch (csrc/compare_ext.hip), sqrt / acos
    from the host's libm in the reference's operation order (smgpu_compare_angular_all_pairs).  -> f64 [n][n], diagonal 1.0."""
    h = _Handles(mhs)
    return _angular_ptrs(h.ptrs, h.n)


def _batchable(v, downsample):
    """Can the pairs of this list be served by batched launches with the per-pair semantics intact?  Yes when every sketch
    is mutually compatible with every other: one (ksize, molecule, seed) (what check_compatible looks at besides the
    threshold, minhash.rs:886-912), and either all bottom-k, or all scaled with one scaled value -- or several, if the caller
    asked for downsampling (every pair is then compared at ITS coarser scaled, minhash.rs:688-696).  Otherwise some pair raises
    in the reference's loop, and the caller lets that very pair raise."""
    if v.n == 0:
        return True
    if (v.params[:, :3] != v.params[0, :3]).any():                 # (comparisons, not np.unique: a sort of 10,000 rows is milliseconds)
        return False
    if (v.num != 0).all():
        return True
    if (v.num != 0).any() or (v.max_hash == 0).any():
        return False
    return downsample or v.one_max_hash


def _by_scaled(mhs, downsample, block):
    """f64 [n][n], diagonal 1.0, assembled from one `block(sketches, scaled) -> matrix` call per distinct scaled value s of the
    list: the sketches with scaled <= s, downsampled to s (a prefix of their hashes, minhash.rs:777-798), give the entries of
    the pairs whose coarser scaled is s -- what similarity(downsample=True) does pair by pair (minhash.rs:688-696).  One
    value (or bottom-k sketches: scaled 0): one call on the sketches as they are.  (Host sketch OBJECTS: only the abundance-
    weighted similarity of a list with several scaled values still comes here; everything count-based goes through
    _by_scaled_counts.)"""
    n = len(mhs)
    scaleds = sorted({mh.scaled for mh in mhs})
    if len(scaleds) == 1:
        return block(mhs, scaleds[0])
    assert downsample
    sc = np.array([mh.scaled for mh in mhs], dtype=np.int64)
    out = np.ones((n, n), dtype=np.float64)
    for s in scaleds:
        idx = np.flatnonzero(sc <= s)
        if len(idx) < 2 or not (sc[idx] == s).any():
            continue
        sub = [mhs[i] if sc[i] == s else mhs[i].downsample(scaled=int(s)) for i in idx]
        m = block(sub, int(s))
        here = np.maximum(sc[idx][:, None], sc[idx][None, :]) == s
        view = out[np.ix_(idx, idx)]
        view[here] = m[here]
        out[np.ix_(idx, idx)] = view
    out[np.arange(n), np.arange(n)] = 1.0
    return out


def _mixed_common(v):
    """A list of scaled sketches with SEVERAL scaled values, every pair at its coarser scaled, in ONE call
    (smgpu_compare_all_pairs_mixed): the sketches travel once as they are, downsampling is a prefix cut on the device
    (minhash.rs:777-798), no downsampled host objects.  -> (common u32 [n][n], the distinct scaled values ascending,
    sizes_at int64 [value][sketch] = len(sketch downsampled to that value), class_of = index of every sketch's own value)."""
    n = v.n
    max_hashes = np.unique(v.max_hash)[::-1].copy()               # ascending SCALED = descending max_hash
    lookup = {int(m): c for c, m in enumerate(max_hashes)}
    class_of = np.array([lookup[int(m)] for m in v.max_hash], dtype=np.uint32)
    scaleds = [int(v.scaled[np.flatnonzero(class_of == c)[0]]) for c in range(len(max_hashes))]
    common = np.empty((n, n), dtype=np.uint32)
    sizes = np.zeros((len(max_hashes), n), dtype=np.uint64)
    mh_arr = np.ascontiguousarray(max_hashes, dtype=np.uint64)
    rustcall(lib.smgpu_compare_all_pairs_mixed, v.ptrs, n, class_of.ctypes.data_as(C.c_void_p), mh_arr.ctypes.data_as(C.c_void_p),
             len(max_hashes), common.ctypes.data_as(C.c_void_p), sizes.ctypes.data_as(C.c_void_p))
    return common, scaleds, sizes.astype(np.int64), class_of.astype(np.int64)


def _by_scaled_counts(v, block):
    """_by_scaled for quantities that are arithmetic on counts and sizes: block(common u32 [m][m], sizes int64 [m], scaled)
    -> f64 [m][m] is called once per scaled value of the list with the sketches that are as fine or finer, their sizes AT
    that value, and the counts of the pairs whose coarser scaled it is (0 elsewhere); the entries of those pairs are kept."""
    n = v.n
    if v.one_max_hash:
        return block(_common_ptrs(v.ptrs, n, want_jaccard=False)[0], v.size, int(v.scaled[0]))
    common, scaleds, sizes_at, class_of = _mixed_common(v)
    out = np.ones((n, n), dtype=np.float64)
    for c, s in enumerate(scaleds):
        idx = np.flatnonzero(class_of <= c)
        if len(idx) < 2 or not (class_of[idx] == c).any():
            continue
        here = np.maximum(class_of[idx][:, None], class_of[idx][None, :]) == c
        m = block(np.where(here, common[np.ix_(idx, idx)], 0).astype(np.uint32), sizes_at[c][idx], int(s))
        view = out[np.ix_(idx, idx)]
        view[here] = m[here]
        out[np.ix_(idx, idx)] = view
    out[np.arange(n), np.arange(n)] = 1.0
    return out


def _jaccard_from_counts(common, sizes):
    "common / max(1, n_i + n_j - common): ONE IEEE divide per entry like csrc/compare.hip: jaccard_from_counts_kernel (minhash.rs:624-631)"
    cm = common.astype(np.int64)
    uni = sizes[:, None] + sizes[None, :] - cm
    jac = cm.astype(np.float64) / np.maximum(uni, 1).astype(np.float64)
    jac[np.arange(len(sizes)), np.arange(len(sizes))] = 1.0
    return jac


def _jaccard_matrix(v, downsample):
    "Jaccard of every pair: bottom-k -> one launch; one scaled value -> one launch incl. the f64 matrix; several -> counts + arithmetic"
    if (v.num != 0).all():
        return _num_ptrs(v.ptrs, v.n)
    if v.one_max_hash:
        return _common_ptrs(v.ptrs, v.n, want_jaccard=True, want_common=False)[1]
    assert downsample
    return _by_scaled_counts(v, lambda cm, sz, s: _jaccard_from_counts(cm, sz))


def _angular(v, downsample):
    "abundance-weighted similarity of every pair (all sketches track abundance)"
    if (v.num != 0).all() or v.one_max_hash:
        return _angular_ptrs(v.ptrs, v.n)
    return _by_scaled(v.minhashes(), downsample, lambda sub, s: angular_matrix(sub))     # several scaled values: host objects


def _raise_like_the_loop(siglist, call):
    "the reference's loop meets an incompatible pair and raises: find the first one in its order and let it raise"
    for i, j in itertools.combinations(range(len(siglist)), 2):
        call(siglist[i], siglist[j])
    raise AssertionError("every pair went through, but the list did not look batchable")


def jaccard_ani_values(jaccard, ksize, n_unique_kmers, err_threshold=1e-4):
    """ANI point estimates from Jaccard on whole arrays: jaccard_to_distance (distance_utils.py:349-407) with r1_to_q,
    exp_n_mutated and var_n_mutated (:128-157) in the reference's operation order -- IEEE multiplies / divides / adds on
    arrays, every `**` through the host libm's pow (what CPython's float ** ends in; pow(x, 2.0) is NOT always x * x), so
    the bits are those of the per-pair Python floats.  -> (ani, withheld): ani = 1 - dist; withheld where the
    approximation-error bound exceeds err_threshold (jaccardANIResult.ani is None there).  ValueError like the reference
    when var_n_mutated turns negative."""
    j = np.ascontiguousarray(jaccard, dtype=np.float64)
    shape = j.shape
    j = j.reshape(-1)
    L = np.ascontiguousarray(n_unique_kmers, dtype=np.float64).reshape(-1)
    k = float(ksize)
    mid = (j != 0) & (j != 1)
    jm, Lm = j[mid], L[mid]
    r1 = 1.0 - _pow(2.0 * jm / (1 + jm), 1.0 / k)
    q = 1 - _pow(1 - r1, k)
    exp_n_mut = Lm * q
    with np.errstate(divide="ignore", invalid="ignore"):
        var_n = (Lm * (1 - q) * (q * (2 * k + (2 / r1) - 1) - 2 * k)
                 + k * (k - 1) * _pow(1 - q, 2.0)
                 + (2 * (1 - q) / _pow(r1, 2.0)) * ((1 + (k - 1) * (1 - q)) * r1 - q))
    var_n = np.where(r1 == 0, 0.0, var_n)
    if (var_n < 0.0).any():
        raise ValueError("Error: varN <0.0!")
    err = 1.0 * Lm * var_n / _pow(Lm + exp_n_mut, 3.0)
    dist = np.where(j == 0, 1.0, 0.0)
    dist[mid] = r1
    withheld = np.zeros(j.shape, dtype=bool)
    withheld[mid] = err > err_threshold
    return (1 - dist).reshape(shape), withheld.reshape(shape)


def _jaccard_ani_block(common, sizes, scaled, ksize):
    "jaccard_ani of every pair of sketches of ONE scaled value from their counts and sizes (minhash.py:749-785) -> f64 matrix, 0.0 = withheld"
    n = len(sizes)
    jac = _jaccard_from_counts(common, sizes)
    sizes = sizes.astype(np.float64)
    n_kmers = np.rint((sizes[:, None] + sizes[None, :]) / 2 * scaled)       # round(avg_sketch_kmers * scaled): half to even, both
    ani, withheld = jaccard_ani_values(jac, ksize, n_kmers)
    out = np.where(withheld, 0.0, ani)
    out[np.arange(n), np.arange(n)] = 1.0
    return out


def _sizes_trusted(sizes, scaled, relative_error=0.20, confidence=0.95):
    """size_is_accurate() of sketches known by their sizes (minhash.py:1099-1120: set_size_exact_prob of len * scaled); scaled: one
    value or one per sketch"""
    from .distance_utils import set_size_exact_prob
    sc = np.broadcast_to(np.asarray(scaled, dtype=np.int64), (len(sizes),))
    return np.array([set_size_exact_prob(int(n) * int(s), int(s), relative_error=relative_error) >= confidence for n, s in zip(sizes, sc)],
                    dtype=bool)


def compare_serial(siglist, ignore_abundance, *, downsample=False, return_ani=False):
    """Similarity matrix (compare.py:14-64).  Per pair the reference computes: Jaccard-derived ANI (return_ani); else the
    angular similarity when both sketches track abundance and it is not ignored, else Jaccard -- with the bottom-k rule for
    num sketches, and at the pair's coarser scaled when downsampling (minhash.rs:682-702).  Here every one of these is a
    batched launch over the whole list (or over the sketches sharing a scaled value); lists in which some pair is
    incompatible raise what the reference's loop raises, from the same pair.  The sketches are read through borrowed views
    (_Views): no clone per signature, one parameter call for the list."""
    siglist = list(siglist)
    n = len(siglist)
    if n < 2:
        return np.ones((n, n))
    v = _Views(siglist)
    if return_ani:
        if (v.max_hash == 0).any() or not _batchable(v, downsample):
            _raise_like_the_loop(siglist, lambda a, b: a.jaccard_ani(b, downsample=downsample))
        ksize = v.ksize
        out = _by_scaled_counts(v, lambda cm, sz, s: _jaccard_ani_block(cm, sz, s, ksize))
        trusted = _sizes_trusted(v.size, v.scaled)                  # of the sketches as given (minhash.py:783)
        out = np.where(trusted[:, None] & trusted[None, :], out, 0.0)
        out[np.arange(n), np.arange(n)] = 1.0
        return out
    if not _batchable(v, downsample):
        _raise_like_the_loop(siglist, lambda a, b: a.similarity(b, ignore_abundance=ignore_abundance, downsample=downsample))
    weighted = np.zeros(0, dtype=np.int64) if ignore_abundance else np.flatnonzero(v.abund)
    if len(weighted) == n:
        return _angular(v, downsample)
    sims = _jaccard_matrix(v, downsample)                            # (the count kernels read the hashes only: no flatten() copies)
    if len(weighted) > 1:                                         # minhash.rs:697-701 decides per pair: both track abundance
        sims[np.ix_(weighted, weighted)] = _angular(v.subset(weighted), downsample)
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


def _containment_block(common, sizes, scaled, ksize, mode, return_ani):
    "containment / max / avg containment (or their ANI point estimates, no trust masking) of sketches of ONE scaled value from their counts"
    n = len(sizes)
    sizes = [int(v) for v in sizes]
    if not return_ani:
        return _debias_matrix(common, sizes, scaled, mode)
    # ANI (compare.py:67-187): the containment of every entry -> point estimate
    if mode == "avg":
        sz = np.asarray(sizes, dtype=np.float64)
        bias = _bias_factors(sizes, scaled)
        cm = np.asarray(common, dtype=np.float64)
        a1 = _ani_from_containment(_debias(cm, sz[None, :], bias[None, :]), ksize).reshape(n, n)
        a2 = _ani_from_containment(_debias(cm, sz[:, None], bias[:, None]), ksize).reshape(n, n)
        return (a1 + a2) / 2
    return _ani_from_containment(_debias_matrix(common, sizes, scaled, mode), ksize).reshape(n, n)


def _containment_mixed(v, mode):
    """containment / max / avg containment of a list with SEVERAL scaled values, downsample = True, in the reference's
    (asymmetric) arithmetic: the count of a pair is taken at the pair's coarser scaled (count_common downsamples the finer
    sketch, minhash.rs:539-548), but the denominator and its bias factor are those of the sketch AS GIVEN -- len(self) and
    self.scaled (minhash.py:819-841), min(len(self), len(other)) with self.scaled for max containment (:881-905), self being
    the sketch with the higher index of the pair (compare.py:111-150).  ONE call for the counts (prefix cut on the device), then
    whole-array arithmetic; the bias factors are one libm pow per (sketch, scaled value)."""
    n = v.n
    cm = _mixed_common(v)[0].astype(np.float64)                   # every pair at its coarser scaled, one call
    sizes = [int(x) for x in v.size]
    sz = v.size.astype(np.float64)
    sc = [int(x) for x in v.scaled]
    own_bias = np.array([_bias_factors([sizes[i]], sc[i])[0] for i in range(n)], dtype=np.float64)
    if mode == "containment":            # [i][j] = siglist[j].contained_by(siglist[i]): j's size, j's scaled
        out = _debias(cm, sz[None, :], own_bias[None, :])
    elif mode == "avg":                  # (self.contained_by(other) + other.contained_by(self)) / 2
        out = (_debias(cm, sz[None, :], own_bias[None, :]) + _debias(cm, sz[:, None], own_bias[:, None])) / 2
    else:                                # max: min of the two sizes, the bias at the scaled of `self` = the higher index
        values = sorted(set(sc))
        bias_at = {s: _bias_factors(sizes, s) for s in values}       # [scaled][sketch]
        idx = np.arange(n)
        hi = np.maximum(idx[:, None], idx[None, :])                 # `self` of the pair
        lo = np.minimum(idx[:, None], idx[None, :])
        small = np.where(sz[hi] <= sz[lo], hi, lo)                  # whose size is min(len(self), len(other)) (ties: the value is the same)
        denom = sz[small]
        sc_arr = np.array(sc)
        bias = np.empty((n, n), dtype=np.float64)
        for s in values:
            m = sc_arr[hi] == s
            bias[m] = bias_at[s][small[m]]
        out = _debias(cm, denom, bias)
    out[np.arange(n), np.arange(n)] = 1.0
    return out


def _containment(siglist, downsample, mode, return_ani):
    siglist = list(siglist)
    n = len(siglist)
    v = _Views(siglist)
    if (v.max_hash == 0).any():
        raise TypeError("Error: can only calculate %s for scaled MinHashes" % ("ANI" if return_ani else "containment"))
    if n < 2:
        return np.ones((n, n))
    if not _batchable(v, downsample):
        return _containment_pairs(siglist, downsample, mode, return_ani)      # some pair raises: from the same pair as the reference's loop
    ksize = v.ksize
    if not return_ani:
        if v.one_max_hash:
            return _by_scaled_counts(v, lambda cm, sz, s: _containment_block(cm, sz, s, ksize, mode, False))
        return _containment_mixed(v, mode)
    # ANI: both sketches are downsampled to the pair's coarser scaled first, then everything is computed there
    # (minhash.py:843-879,907-944): the blocks of _by_scaled_counts are exactly that.  An estimate is withheld (0 in the matrix)
    # when a sketch is too small for its size to be trusted -- and WHICH sketch is asked differs by mode, as in the reference:
    # containment / max go through MinHash.{containment,max_containment}_ani, which ask the sketches AS GIVEN
    # (minhash.py:877-878,938-939); avg goes through FracMinHashComparison (compare.py:166-168), whose mh1_cmp / mh2_cmp are the
    # sketches already downsampled to the pair's scaled (sketchcomparison.py:53-70,143-170) -- a sketch trusted at its own
    # scaled but not after downsampling gives 0.0 there.
    def trust_mask(m, ok):
        ok = np.asarray(ok, dtype=bool)
        return np.where(ok[:, None] & ok[None, :], m, 0.0)
    if mode == "avg":
        out = _by_scaled_counts(v, lambda cm, sz, s: trust_mask(_containment_block(cm, sz, s, ksize, mode, True), _sizes_trusted(sz, s)))
    else:
        out = trust_mask(_by_scaled_counts(v, lambda cm, sz, s: _containment_block(cm, sz, s, ksize, mode, True)),
                         _sizes_trusted(v.size, v.scaled))
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
