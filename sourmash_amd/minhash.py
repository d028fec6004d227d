"""MinHash / FrozenMinHash -- the sketch object API of the reference
(src/sourmash/minhash.py:162-1258) over libsourmash_amd.so.

Same constructor, method names, argument meaning and error behaviour as the
reference class, so code and tests written against ``sourmash.MinHash`` read
the same here.  Differences are underneath: hashing k-mers and intersecting
sketches run in HIP kernels on the MI355X (no CPU fallback), and the object
also offers batch helpers (``add_sequence_buffer``) that hand whole buffers to
the GPU in one call.
"""
import ctypes as C
import os
import warnings
from collections.abc import Mapping

import numpy as np

from ._lowlevel import ffi, lib
from .utils import RustObject, rustcall, decode_str

__all__ = ["get_minhash_default_seed", "get_minhash_max_hash", "hash_murmur", "translate_codon", "MinHash",
           "FrozenMinHash"]

MINHASH_DEFAULT_SEED = 42
MINHASH_MAX_HASH = 0xFFFFFFFFFFFFFFFF


def get_minhash_default_seed():
    "Default MurmurHash seed."
    return MINHASH_DEFAULT_SEED


def get_minhash_max_hash():
    "Largest possible hash value (2**64 - 1)."
    return MINHASH_MAX_HASH


def _get_max_hash_for_scaled(scaled):
    # src/sourmash/minhash.py:53-60 (round(), not truncation; same value for every scaled used in practice)
    if scaled == 0:
        return 0
    if scaled == 1:
        return MINHASH_MAX_HASH
    return min(int(round(MINHASH_MAX_HASH / scaled, 0)), MINHASH_MAX_HASH)


def _get_scaled_for_max_hash(max_hash):
    # src/sourmash/minhash.py:63-67
    if max_hash == 0:
        return 0
    return min(int(round(MINHASH_MAX_HASH / max_hash, 0)), MINHASH_MAX_HASH)


def to_bytes(s):
    "str / bytes / int -> bytes (src/sourmash/minhash.py:70-85)"
    if isinstance(s, bytes):
        return s
    if isinstance(s, str):
        return s.encode("utf-8")
    if isinstance(s, int):
        return bytes([s])
    if isinstance(s, (bytearray, memoryview)):
        return bytes(s)
    raise TypeError("Requires a string-like sequence")


def hash_murmur(kmer, seed=MINHASH_DEFAULT_SEED):
    "MurmurHash3_x64_128 (low 64 bits) of a string; default seed 42."
    return lib.hash_murmur(to_bytes(kmer), seed)


def translate_codon(codon):
    "One codon (1-3 bases; a missing third base counts as N) -> residue letter; ValueError for other lengths."
    from .exceptions import SourmashError
    try:
        return rustcall(lib.sourmash_translate_codon, to_bytes(codon)).decode("utf-8")
    except SourmashError as e:
        raise ValueError(e.message)


def flatten_and_downsample_scaled(mh, *scaled_vals):
    "Flatten and downsample to the max of the given scaled values."
    assert mh.scaled
    assert all(x > 0 for x in scaled_vals)
    mh = mh.flatten()
    scaled = max(scaled_vals)
    return mh.downsample(scaled=scaled) if scaled > mh.scaled else mh


def flatten_and_downsample_num(mh, *num_vals):
    "Flatten and downsample to the min of the given num values."
    assert mh.num
    assert all(x > 0 for x in num_vals)
    mh = mh.flatten()
    num = min(num_vals)
    return mh.downsample(num=num) if num < mh.num else mh


def flatten_and_intersect_scaled(mh1, mh2):
    "Flatten + downsample two scaled sketches to a common scaled, then intersect."
    scaled = max(mh1.scaled, mh2.scaled)
    return mh1.flatten().downsample(scaled=scaled) & mh2.flatten().downsample(scaled=scaled)


_add_sequence_rc = lib.smgpu_minhash_add_sequence_rc


def _raise_last(code):
    from .exceptions import SourmashError, exceptions_by_code
    from .utils import decode_str
    raise exceptions_by_code.get(code, SourmashError)(decode_str(lib.sourmash_err_get_last_message()))


class _HashesWrapper(Mapping):
    "Read-only {hash: abundance} view."

    def __init__(self, h):
        self._data = h

    def __getitem__(self, key):
        return self._data[key]

    def __repr__(self):
        return repr(self._data)

    def __len__(self):
        return len(self._data)

    def __iter__(self):
        return iter(self._data)

    def __eq__(self, other):
        return list(self.items()) == list(other.items())

    def __setitem__(self, k, v):
        raise RuntimeError("cannot modify hashes directly; use 'add' methods")


_DNA_COMPLEMENT = str.maketrans("ACGT", "TGCA")


def _u64_array(values):
    arr = np.ascontiguousarray(values, dtype=np.uint64) if not isinstance(values, np.ndarray) \
        else np.ascontiguousarray(values.astype(np.uint64, copy=False))
    return arr, arr.ctypes.data_as(C.POINTER(C.c_uint64)), arr.size


class _AddSequencePython:
    "add_sequence through ctypes: what MinHash uses when the C-API entry (csrc/fastcall.c) has not been built."

    def add_sequence(self, sequence, force=False):
        """Add every k-mer of a DNA sequence (GPU).  The record is validated and queued; the library hashes the queue in
        one kernel launch when it is large or when the sketch is next looked at, so a loop over reads costs one C call
        per read and no launch.  Invalid DNA with force=False raises here, after the k-mers in front of the bad one
        were queued -- the reference's streaming order (signature.rs:48-54)."""
        b = to_bytes(sequence)
        code = _add_sequence_rc(self._get_objptr(), b, len(b), force)
        if code:
            _raise_last(code)


# The per-record call is the one place where the binding is the cost (ctypes: 0.65 us per 150-bp read, two thirds of it
# argument conversion).  csrc/fastcall.c is the same call as a C method; it is bound to the entry point of the library
# loaded above.  SMG_NO_FASTCALL=1 keeps the ctypes route (tests compare the two).
try:
    if os.environ.get("SMG_NO_FASTCALL") == "1":
        raise ImportError("disabled")
    from . import _fastcall
    _fastcall.bind(C.cast(_add_sequence_rc, C.c_void_p).value, _raise_last)
    _AddSequence = _fastcall.AddSequenceBase
except ImportError:
    _fastcall = None
    _AddSequence = _AddSequencePython


class MinHash(RustObject, _AddSequence):
    """The sketch object.

    ``MinHash(n, ksize, ...)`` builds a bottom-``n`` sketch, ``MinHash(0, ksize,
    scaled=s)`` a FracMinHash keeping every hash <= 2**64 / s.
    """

    __dealloc_func__ = lib.kmerminhash_free

    def __init__(self, n, ksize, *, is_protein=False, dayhoff=False, hp=False, track_abundance=False,
                 seed=MINHASH_DEFAULT_SEED, max_hash=0, mins=None, scaled=0):
        if max_hash:
            if scaled:
                raise ValueError("cannot set both max_hash and scaled")
            scaled = _get_scaled_for_max_hash(max_hash)
        if scaled and n:
            raise ValueError("cannot set both n and max_hash")
        if not n and not scaled:
            raise ValueError("cannot omit both n and scaled")
        if dayhoff or hp:
            is_protein = False
        if dayhoff:
            hash_function, ksize = lib.HASH_FUNCTIONS_MURMUR64_DAYHOFF, ksize * 3
        elif hp:
            hash_function, ksize = lib.HASH_FUNCTIONS_MURMUR64_HP, ksize * 3
        elif is_protein:
            hash_function, ksize = lib.HASH_FUNCTIONS_MURMUR64_PROTEIN, ksize * 3
        else:
            hash_function = lib.HASH_FUNCTIONS_MURMUR64_DNA
        self._objptr = lib.kmerminhash_new(scaled, ksize, hash_function, seed, track_abundance, n)
        if mins:
            if track_abundance:
                self.set_abundances(mins)
            else:
                self.add_many(mins)

    # ---- construction helpers ---------------------------------------------------------------
    def _like(self, *, num=None, track_abundance=None, max_hash=None):
        return MinHash(self.num if num is None else num, self.ksize, is_protein=self.is_protein,
                       dayhoff=self.dayhoff, hp=self.hp,
                       track_abundance=self.track_abundance if track_abundance is None else track_abundance,
                       seed=self.seed, max_hash=self._max_hash if max_hash is None else max_hash)

    def __copy__(self):
        a = self._like()
        a.merge(self)
        return a

    copy = __copy__

    def copy_and_clear(self):
        "An empty sketch with the same parameters."
        return self._like()

    def __getstate__(self):
        # same tuple layout as the reference (src/sourmash/minhash.py:276-293): pickles interoperate
        return (self.num, self.ksize if self.is_dna else self.ksize * 3, self.is_protein, self.dayhoff, self.hp,
                self.hashes, None, self.track_abundance, self._max_hash, self.seed)

    def __setstate__(self, tup):
        (n, ksize, is_protein, dayhoff, hp, mins, _, track_abundance, max_hash, seed) = tup
        self.__del__()
        hf = (lib.HASH_FUNCTIONS_MURMUR64_DAYHOFF if dayhoff else lib.HASH_FUNCTIONS_MURMUR64_HP if hp
              else lib.HASH_FUNCTIONS_MURMUR64_PROTEIN if is_protein else lib.HASH_FUNCTIONS_MURMUR64_DNA)
        self._shared = False
        self._objptr = lib.kmerminhash_new(_get_scaled_for_max_hash(max_hash), ksize, hf, seed, track_abundance, n)
        if track_abundance:
            MinHash.set_abundances(self, mins)     # explicit base calls: FrozenMinHash blocks the bound ones
        else:
            MinHash.add_many(self, mins)

    def __eq__(self, other):
        return self.__getstate__() == other.__getstate__()

    # ---- adding -------------------------------------------------------------------------------
    # add_sequence(sequence, force=False): inherited from _AddSequence (the C method, or its ctypes twin above)

    def add_sequence_buffer(self, buf, force=True):
        """Batch extension: sketch a whole buffer in one call.  Records are separated
        by any byte outside ACGTacgt (newline, '>', NUL ...); embedded NULs are fine."""
        b = to_bytes(buf)
        self._methodcall(lib.smgpu_minhash_add_buffer, b, len(b), force)

    def seq_to_hashes(self, sequence, *, force=False, bad_kmers_as_zeroes=False, is_protein=False):
        "Hashes of the k-mers of `sequence`, in order, without adding them."
        if is_protein and self.moltype not in ("protein", "dayhoff", "hp"):
            raise ValueError("cannot add protein sequence to DNA MinHash")
        if bad_kmers_as_zeroes and not force:
            raise ValueError("cannot represent invalid kmers as 0 while force is not set to True")
        b = to_bytes(sequence)
        size = ffi.new_size()
        ptr = self._methodcall(lib.kmerminhash_seq_to_hashes, b, len(b), force, bad_kmers_as_zeroes, is_protein,
                               C.byref(size))
        try:
            return ffi.unpack_u64(ptr, size.value)
        finally:
            lib.kmerminhash_slice_free(ptr, size.value)

    def kmers_and_hashes(self, sequence, *, force=False, is_protein=False):
        """Yield (k-mer, hash) for every k-mer without adding anything; invalid DNA k-mers give None when force
        is set.  DNA handed to a protein / dayhoff / hp sketch is translated: the k-mers are then the DNA
        windows of 3 x ksize bases in the order the hashes come (frame by frame, forward strand, then reverse
        complement)."""
        sequence = sequence.upper()
        hashvals = self.seq_to_hashes(sequence, force=force, is_protein=is_protein, bad_kmers_as_zeroes=force)
        if force:
            hashvals = [None if h == 0 else h for h in hashvals]
        translate = self.moltype != "DNA" and not is_protein
        if not translate:
            ksize = self.ksize
            assert len(hashvals) == max(len(sequence) - ksize + 1, 0)
            for i, h in enumerate(hashvals):
                yield sequence[i:i + ksize], h
            return
        span = self.ksize * 3
        assert len(hashvals) == max(len(sequence) - span + 1, 0) * 2
        revcomp = sequence.translate(_DNA_COMPLEMENT)[::-1]
        it = iter(hashvals)
        for frame in (0, 1, 2):
            for strand in (sequence, revcomp):
                for start in range(frame, len(strand) - span + 1, 3):
                    yield strand[start:start + span], next(it)

    def add_kmer(self, kmer):
        "Add one k-mer."
        want = self.ksize if self.is_dna else self.ksize * 3
        if len(kmer) != want:
            raise ValueError(f"kmer to add is not {want} in length")
        self.add_sequence(kmer)

    def add_many(self, hashes):
        "Add hashes from an iterable or another MinHash."
        if isinstance(hashes, MinHash):
            self._methodcall(lib.kmerminhash_add_from, hashes._get_objptr())
        else:
            arr, ptr, n = _u64_array(list(hashes) if not isinstance(hashes, np.ndarray) else hashes)
            self._methodcall(lib.kmerminhash_add_many, ptr, n)

    def remove_many(self, hashes):
        "Remove hashes given by an iterable or another MinHash."
        if isinstance(hashes, MinHash):
            self._methodcall(lib.kmerminhash_remove_from, hashes._get_objptr())
        else:
            arr, ptr, n = _u64_array(list(hashes) if not isinstance(hashes, np.ndarray) else hashes)
            self._methodcall(lib.kmerminhash_remove_many, ptr, n)

    def add_hash(self, h):
        return self._methodcall(lib.kmerminhash_add_hash, h)

    def add_hash_with_abundance(self, h, a):
        if not self.track_abundance:
            raise RuntimeError("Use track_abundance=True when constructing the MinHash to use add_hash_with_abundance.")
        return self._methodcall(lib.kmerminhash_add_hash_with_abundance, h, a)

    def add_protein(self, sequence):
        self._methodcall(lib.kmerminhash_add_protein, to_bytes(sequence))

    def clear(self):
        return self._methodcall(lib.kmerminhash_clear)

    def set_abundances(self, values, clear=True):
        "Set abundances from {hash: abund}; abundance 0 removes the hash."
        if not self.track_abundance:
            raise RuntimeError("Use track_abundance=True when constructing the MinHash to use set_abundances.")
        hashes, abunds = [], []
        for h, v in values.items():
            if v < 0:
                raise ValueError("Abundance cannot be set to a negative value.")
            hashes.append(h)
            abunds.append(v)
        ah, ph, n = _u64_array(hashes)
        aa, pa, _ = _u64_array(abunds)
        self._methodcall(lib.kmerminhash_set_abundances, ph, pa, n, clear)

    # ---- reading ------------------------------------------------------------------------------
    def __len__(self):
        return self._methodcall(lib.kmerminhash_get_mins_size)

    def _mins_array(self):
        "numpy u64 copy of the sorted hashes (batch helper)."
        size = ffi.new_size()
        ptr = self._methodcall(lib.kmerminhash_get_mins, C.byref(size))
        try:
            n = size.value
            return np.ctypeslib.as_array(ptr, shape=(n,)).copy() if n else np.zeros(0, dtype=np.uint64)
        finally:
            lib.kmerminhash_slice_free(ptr, size.value)

    @property
    def hashes(self):
        mins = self._mins_array().tolist()
        if self.track_abundance:
            size = ffi.new_size()
            ptr = self._methodcall(lib.kmerminhash_get_abunds, C.byref(size))
            try:
                assert size.value == len(mins)
                return _HashesWrapper(dict(zip(mins, ffi.unpack_u64(ptr, size.value))))
            finally:
                lib.kmerminhash_slice_free(ptr, size.value)
        return _HashesWrapper({k: 1 for k in mins})

    def get_mins(self, with_abundance=False):
        "Deprecated since the reference's 3.5 (src/sourmash/minhash.py:498-511): use .hashes."
        warnings.warn("get_mins is deprecated; use the .hashes property", DeprecationWarning, stacklevel=2)
        mins = self.hashes
        return mins if with_abundance else mins.keys()

    def get_hashes(self):
        "Deprecated since the reference's 3.5 (src/sourmash/minhash.py:513-521): use .hashes."
        warnings.warn("get_hashes is deprecated; use the .hashes property", DeprecationWarning, stacklevel=2)
        return self.hashes.keys()

    @property
    def seed(self):
        return self._methodcall(lib.kmerminhash_seed)

    @property
    def num(self):
        return self._methodcall(lib.kmerminhash_num)

    @property
    def scaled(self):
        mx = self._methodcall(lib.kmerminhash_max_hash)
        return _get_scaled_for_max_hash(mx) if mx else 0

    @property
    def is_dna(self):
        return not (self.is_protein or self.dayhoff or self.hp)

    @property
    def is_protein(self):
        return self._methodcall(lib.kmerminhash_is_protein)

    @property
    def dayhoff(self):
        return self._methodcall(lib.kmerminhash_dayhoff)

    @property
    def hp(self):
        return self._methodcall(lib.kmerminhash_hp)

    @property
    def ksize(self):
        k = self._methodcall(lib.kmerminhash_ksize)
        if not self.is_dna:
            assert k % 3 == 0
            k //= 3
        return k

    @property
    def max_hash(self):
        return self._methodcall(lib.kmerminhash_max_hash)

    @property
    def _max_hash(self):
        return self._methodcall(lib.kmerminhash_max_hash)

    @property
    def track_abundance(self):
        return self._methodcall(lib.kmerminhash_track_abundance)

    @track_abundance.setter
    def track_abundance(self, b):
        if self.track_abundance == b:
            return
        if b is False:
            self._methodcall(lib.kmerminhash_disable_abundance)
        elif len(self) > 0:
            raise RuntimeError("Can only set track_abundance=True if the MinHash is empty")
        else:
            self._methodcall(lib.kmerminhash_enable_abundance)

    @property
    def moltype(self):
        return "protein" if self.is_protein else "dayhoff" if self.dayhoff else "hp" if self.hp else "DNA"

    def md5sum(self):
        return decode_str(self._methodcall(lib.kmerminhash_md5sum))

    # ---- set operations (GPU intersections) -----------------------------------------------------
    def count_common(self, other, downsample=False):
        "Number of shared hashes; optionally downsample to the coarser scaled."
        if not isinstance(other, MinHash):
            raise TypeError("Must be a MinHash!")
        return self._methodcall(lib.kmerminhash_count_common, other._get_objptr(), downsample)

    def intersection_and_union_size(self, other):
        if not isinstance(other, MinHash):
            raise TypeError("Must be a MinHash!")
        if not self.is_compatible(other):
            raise TypeError("incompatible MinHash objects")
        usize = ffi.new_u64()
        common = self._methodcall(lib.kmerminhash_intersection_union_size, other._get_objptr(), C.byref(usize))
        return common, usize.value

    def downsample(self, *, num=None, scaled=None):
        "Copy with fewer hashes: a smaller num, or a larger scaled."
        if num is None and scaled is None:
            raise ValueError("must specify either num or scaled to downsample")
        if num is not None and scaled is not None:
            raise ValueError("cannot specify both num and scaled")
        if num is not None:
            if self.scaled:
                raise ValueError("cannot downsample a scaled MinHash using num")
            if self.num < num:
                raise ValueError("new sample num is higher than current sample num")
            max_hash = 0
        else:
            if self.num:
                raise ValueError("cannot downsample a num MinHash using scaled")
            if self.scaled > scaled:
                raise ValueError(f"new scaled {scaled} is lower than current sample scaled {self.scaled}")
            max_hash = _get_max_hash_for_scaled(scaled)
            num = 0
        a = self._like(num=num, max_hash=max_hash)
        if self.track_abundance:
            a.set_abundances(self.hashes)
        else:
            a.add_many(self)
        return a

    def flatten(self):
        "Drop abundances (returns self if there are none)."
        if not self.track_abundance:
            return self
        a = self._like(track_abundance=False)
        a.add_many(self)
        return a

    def jaccard(self, other, downsample=False):
        if self.num != other.num:
            raise TypeError(f"must have same num: {self.num} != {other.num}")
        return self._methodcall(lib.kmerminhash_similarity, other._get_objptr(), True, downsample)

    def similarity(self, other, ignore_abundance=False, downsample=False):
        "Jaccard, or angular similarity when both sketches track abundance."
        return self._methodcall(lib.kmerminhash_similarity, other._get_objptr(), ignore_abundance, downsample)

    def angular_similarity(self, other):
        if not (self.track_abundance and other.track_abundance):
            raise TypeError("Error: Angular (cosine) similarity requires both sketches to track hash abundance.")
        return self._methodcall(lib.kmerminhash_angular_similarity, other._get_objptr())

    def is_compatible(self, other):
        return self._methodcall(lib.kmerminhash_is_compatible, other._get_objptr())

    @staticmethod
    def _debias(count, denom, scaled):
        # src/sourmash/minhash.py:819-841: count / (denom * (1 - (1 - 1/scaled)^(denom*scaled))), clamped to [0, 1]
        bias_factor = 1.0 - (1.0 - 1.0 / scaled) ** float(denom * scaled)
        c = count / (denom * bias_factor)
        return 1.0 if c >= 1 else 0.0 if c <= 0 else c

    def contained_by(self, other, downsample=False):
        "Fraction of self's hashes found in other (de-biased)."
        if not (self.scaled and other.scaled):
            raise TypeError("Error: can only calculate containment for scaled MinHashes")
        denom = len(self)
        if not denom:
            return 0.0
        return self._debias(self.count_common(other, downsample), denom, self.scaled)

    def max_containment(self, other, downsample=False):
        if not (self.scaled and other.scaled):
            raise TypeError("Error: can only calculate containment for scaled MinHashes")
        min_denom = min(len(self), len(other))
        if not min_denom:
            return 0.0
        return self._debias(self.count_common(other, downsample), min_denom, self.scaled)

    def avg_containment(self, other, *, downsample=False):
        if not (self.scaled and other.scaled):
            raise TypeError("Error: can only calculate containment for scaled MinHashes")
        return (self.contained_by(other, downsample) + other.contained_by(self, downsample)) / 2

    # ---- ANI (host float layer, src/sourmash/minhash.py:749-976) ----------------------------------
    def _ani_pair(self, other, downsample):
        if not (self.scaled and other.scaled):
            raise TypeError("Error: can only calculate ANI for scaled MinHashes")
        a, b, scaled = self, other, self.scaled
        if downsample:
            scaled = max(a.scaled, b.scaled)
            a, b = a.downsample(scaled=scaled), b.downsample(scaled=scaled)
        return a, b, scaled

    def jaccard_ani(self, other, *, downsample=False, jaccard=None, prob_threshold=1e-3, err_threshold=1e-4):
        from .distance_utils import jaccard_to_distance
        a, b, scaled = self._ani_pair(other, downsample)
        if jaccard is None:
            jaccard = a.similarity(b, ignore_abundance=True)
        avg_n_kmers = round((len(a) + len(b)) / 2 * scaled)
        res = jaccard_to_distance(jaccard, a.ksize, scaled, n_unique_kmers=avg_n_kmers,
                                  prob_threshold=prob_threshold, err_threshold=err_threshold)
        if not self.size_is_accurate() or not other.size_is_accurate():
            res.size_is_inaccurate = True
        return res

    def containment_ani(self, other, *, downsample=False, containment=None, confidence=0.95, estimate_ci=False,
                        prob_threshold=1e-3):
        from .distance_utils import containment_to_distance
        a, b, scaled = self._ani_pair(other, downsample)
        if containment is None:
            containment = a.contained_by(b)
        res = containment_to_distance(containment, a.ksize, a.scaled, n_unique_kmers=len(a) * scaled,
                                      confidence=confidence, estimate_ci=estimate_ci, prob_threshold=prob_threshold)
        if not self.size_is_accurate() or not other.size_is_accurate():
            res.size_is_inaccurate = True
        return res

    def max_containment_ani(self, other, *, downsample=False, max_containment=None, confidence=0.95,
                            estimate_ci=False, prob_threshold=1e-3):
        from .distance_utils import containment_to_distance
        a, b, scaled = self._ani_pair(other, downsample)
        if max_containment is None:
            max_containment = a.max_containment(b)
        res = containment_to_distance(max_containment, a.ksize, scaled,
                                      n_unique_kmers=min(len(a), len(b)) * scaled, confidence=confidence,
                                      estimate_ci=estimate_ci, prob_threshold=prob_threshold)
        if not self.size_is_accurate() or not other.size_is_accurate():
            res.size_is_inaccurate = True
        return res

    def avg_containment_ani(self, other, *, downsample=False, prob_threshold=1e-3):
        a1 = self.containment_ani(other, downsample=downsample, prob_threshold=prob_threshold).ani
        a2 = other.containment_ani(self, downsample=downsample, prob_threshold=prob_threshold).ani
        if a1 is None or a2 is None:
            return None
        return (a1 + a2) / 2

    # ---- combining ------------------------------------------------------------------------------------
    def __add__(self, other):
        if not isinstance(other, MinHash):
            raise TypeError("can only add MinHash objects to MinHash objects!")
        if self.num and other.num and self.num != other.num:
            raise TypeError(f"incompatible num values: self={self.num} other={other.num}")
        new_obj = self.to_mutable()
        new_obj += other
        return new_obj

    __or__ = __add__

    def __iadd__(self, other):
        if not isinstance(other, MinHash):
            raise TypeError("can only add MinHash objects to MinHash objects!")
        self._methodcall(lib.kmerminhash_merge, other._get_objptr())
        return self

    def merge(self, other):
        if not isinstance(other, MinHash):
            raise TypeError("can only add MinHash objects to MinHash objects!")
        self._methodcall(lib.kmerminhash_merge, other._get_objptr())

    def intersection(self, other):
        "New sketch holding the shared hashes (flat sketches only)."
        if not isinstance(other, MinHash):
            raise TypeError("can only intersect MinHash objects")
        if self.track_abundance or other.track_abundance:
            raise TypeError("can only intersect flat MinHash objects")
        return MinHash._from_objptr(self._methodcall(lib.kmerminhash_intersection, other._get_objptr()))

    __and__ = intersection

    def to_mutable(self):
        return self.__copy__()

    def to_frozen(self):
        new_mh = self.__copy__()
        new_mh.__class__ = FrozenMinHash
        return new_mh

    def into_frozen(self):
        self.__class__ = FrozenMinHash

    def inflate(self, from_mh):
        "Copy abundances for this flat sketch's hashes from `from_mh`."
        if not self.track_abundance and from_mh.track_abundance:
            orig = from_mh.hashes
            abunds = {h: orig.get(h, 0) for h in self.hashes}     # abundance 0 removes the hash
            abund_mh = from_mh.copy_and_clear()
            abund_mh.set_abundances(abunds)
            return abund_mh
        raise ValueError("inflate operates on a flat MinHash and takes a MinHash object with track_abundance=True")

    @property
    def sum_abundances(self):
        return sum(self.hashes.values()) if self.track_abundance else None

    @property
    def mean_abundance(self):
        return float(np.mean(list(self.hashes.values()))) if self.track_abundance else None

    @property
    def median_abundance(self):
        return float(np.median(list(self.hashes.values()))) if self.track_abundance else None

    @property
    def std_abundance(self):
        return float(np.std(list(self.hashes.values()))) if self.track_abundance else None

    @property
    def unique_dataset_hashes(self):
        "Estimated number of distinct k-mers behind this sketch (scaled only)."
        if not self.scaled:
            raise TypeError("can only approximate unique_dataset_hashes for scaled MinHashes")
        return len(self) * self.scaled

    def size_is_accurate(self, relative_error=0.20, confidence=0.95):
        "Is the sketch large enough for len*scaled to estimate the k-mer count well?"
        from .distance_utils import set_size_exact_prob
        if not self.scaled:
            raise TypeError("Error: can only estimate dataset size for scaled MinHashes")
        if any(not 0 <= v <= 1 for v in (relative_error, confidence)):
            raise ValueError("Error: relative error and confidence values must be between 0 and 1.")
        probability = set_size_exact_prob(self.unique_dataset_hashes, self.scaled, relative_error=relative_error)
        return probability >= confidence


def _read_only(*_a, **_k):
    raise TypeError("FrozenMinHash does not support modification")


class FrozenMinHash(MinHash):
    "Read-only MinHash (src/sourmash/minhash.py:1152-1258)."
    add_sequence = add_sequence_buffer = add_kmer = add_many = remove_many = add_hash = _read_only
    add_hash_with_abundance = clear = set_abundances = add_protein = __iadd__ = merge = _read_only

    def downsample(self, *, num=None, scaled=None):
        if scaled and self.scaled == scaled:
            return self
        if num and self.num == num:
            return self
        return MinHash.downsample(self, num=num, scaled=scaled).to_frozen()

    def flatten(self):
        if not self.track_abundance:
            return self
        return MinHash.flatten(self).to_frozen()

    def to_mutable(self):
        mut = self._like()                 # same parameters, empty; then one C-level sorted merge
        MinHash.merge(mut, self)
        return mut

    def to_frozen(self):
        return self

    def into_frozen(self):
        pass

    def __setstate__(self, tup):
        MinHash.__setstate__(self, tup)

    def __copy__(self):
        return self

    copy = __copy__
