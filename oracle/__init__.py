"""ctypes front-end of the CPU oracle (oracle/oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of oracle.c.  Allowed importers:
tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  The product
package (sourmash_amd/) never imports this module.

Parity status: pinned (tests/test_oracle.py checks it against the reference's
golden fixtures under tests/golden/ and its known-answer tests).
"""
import ctypes as C
import gzip
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("ORACLE_LIBRARY") or os.path.join(_HERE, "liboracle.so")   # ORACLE_LIBRARY: the sanitizer build

u64 = C.c_uint64
u64p = C.POINTER(C.c_uint64)


def build(force=False):
    """Compile oracle.c -> liboracle.so (gcc, see oracle/Makefile)."""
    src = os.path.join(_HERE, "oracle.c")
    if os.environ.get("ORACLE_LIBRARY"):
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "clean"])
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    vp = C.c_void_p
    sig = {
        "orc_hash_murmur": (u64, [C.c_char_p, u64, u64]),
        "orc_max_hash_for_scaled": (u64, [u64]),
        "orc_scaled_for_max_hash": (u64, [u64]),
        "orc_seq_to_hashes_dna": (C.c_int64, [C.c_char_p, u64, C.c_uint32, u64, C.c_int, vp]),
        "orc_mh_new": (vp, [u64, C.c_uint32, C.c_uint32, u64, C.c_int, C.c_uint32]),
        "orc_mh_free": (None, [vp]),
        "orc_mh_clone": (vp, [vp]),
        "orc_mh_size": (u64, [vp]),
        "orc_mh_mins": (u64p, [vp]),
        "orc_mh_abunds": (u64p, [vp]),
        "orc_mh_max_hash": (u64, [vp]),
        "orc_mh_clear": (None, [vp]),
        "orc_mh_remove_hash": (None, [vp, u64]),
        "orc_mh_add_hash_with_abundance": (None, [vp, u64, u64]),
        "orc_mh_add_hash": (None, [vp, u64]),
        "orc_mh_add_many": (None, [vp, vp, u64]),
        "orc_mh_remove_many": (None, [vp, vp, u64]),
        "orc_mh_add_sequence": (C.c_int64, [vp, C.c_char_p, u64, C.c_int]),
        "orc_mh_add_protein": (None, [vp, C.c_char_p, u64]),
        "orc_translate_codon": (C.c_uint8, [C.c_uint8, C.c_uint8, C.c_uint8]),
        "orc_aa_to_dayhoff": (C.c_uint8, [C.c_uint8]),
        "orc_aa_to_hp": (C.c_uint8, [C.c_uint8]),
        "orc_seq_to_hashes_protein": (u64, [C.c_char_p, u64, C.c_uint32, C.c_uint32, u64, C.c_int, vp]),
        "orc_mh_check_compatible": (C.c_uint32, [vp, vp]),
        "orc_mh_merge": (C.c_uint32, [vp, vp]),
        "orc_intersection_size": (u64, [vp, u64, vp, u64, u64p]),
        "orc_intersection": (u64, [vp, u64, vp, u64, vp]),
        "orc_mh_downsample_scaled": (vp, [vp, u64, C.POINTER(C.c_uint32)]),
        "orc_mh_count_common": (u64, [vp, vp, C.c_int, C.POINTER(C.c_uint32)]),
        "orc_mh_intersection_size": (u64, [vp, vp, u64p, C.POINTER(C.c_uint32)]),
        "orc_mh_jaccard": (C.c_double, [vp, vp, C.POINTER(C.c_uint32)]),
        "orc_mh_angular_similarity": (C.c_double, [vp, vp, C.POINTER(C.c_uint32)]),
        "orc_mh_similarity": (C.c_double, [vp, vp, C.c_int, C.c_int, C.POINTER(C.c_uint32)]),
        "orc_md5_hex": (None, [C.c_char_p, u64, C.c_char_p]),
        "orc_md5sum_hashes": (None, [C.c_uint32, vp, u64, C.c_char_p]),
        "orc_mh_md5sum": (None, [vp, C.c_char_p]),
        "orc_sketch_dna_bulk": (u64, [vp, u64, C.c_uint32, u64, u64, C.c_int, C.POINTER(u64p)]),
        "orc_free": (None, [vp]),
        "orc_splitmix64": (u64, [u64]),
        "orc_synth_dna": (None, [vp, u64, u64, u64, u64]),
        "orc_compare_all_pairs": (None, [vp, vp, u64, vp, vp, C.c_int]),
        "orc_gather": (u64, [vp, u64, vp, vp, u64, u64, u64, vp, vp, u64]),
        "orc_gather_mt": (u64, [vp, u64, vp, vp, u64, u64, u64, vp, vp, u64, C.c_int]),
        "orc_contained_by": (C.c_double, [u64, u64, u64]),
        "orc_max_containment": (C.c_double, [u64, u64, u64, u64]),
        "orc_avg_containment": (C.c_double, [u64, u64, u64, u64]),
        "orc_containment_to_distance_point": (C.c_double, [C.c_double, C.c_uint32]),
        "orc_similarity_matrix": (None, [vp, u64, C.c_int, C.c_int, vp, C.POINTER(C.c_uint32), C.c_int]),
        "orc_jaccard_to_distance": (C.c_double, [C.c_double, C.c_uint32, u64, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def usable_cpus():
    """Threads this process can really run at once: the scheduler affinity mask, capped by the cgroup CPU quota
    (cpu.max "quota period" on cgroup v2, cpu.cfs_quota_us / cpu.cfs_period_us on v1).  os.cpu_count() is the
    machine's figure and oversubscribes a container that was given fewer."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(p)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _as_bytes(s):
    return s.encode("utf-8") if isinstance(s, str) else bytes(s)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# --------------------------------------------------------------------------- #
# thin functional API
# --------------------------------------------------------------------------- #
def hash_murmur(kmer, seed=42):
    b = _as_bytes(kmer)
    return lib().orc_hash_murmur(b, len(b), seed)


def max_hash_for_scaled(scaled):
    return lib().orc_max_hash_for_scaled(scaled)


def scaled_for_max_hash(max_hash):
    return lib().orc_scaled_for_max_hash(max_hash)


class InvalidDNA(ValueError):
    def __init__(self, kmer):
        super().__init__(f"invalid DNA character in input k-mer: {kmer}")
        self.kmer = kmer


def seq_to_hashes(seq, ksize, seed=42, force=False, bad_kmers_as_zeroes=False):
    """Per-k-mer hashes of a DNA sequence (ffi/minhash.rs:63-99 semantics)."""
    b = _as_bytes(seq)
    if len(b) < ksize:
        return []
    out = np.zeros(len(b) - ksize + 1, dtype=np.uint64)
    r = lib().orc_seq_to_hashes_dna(b, len(b), ksize, seed, int(force), _ptr(out))
    if r < 0:
        i = -1 - r
        raise InvalidDNA(b[i:i + ksize].upper().decode("latin-1"))
    hs = out[:r].tolist()
    if force and bad_kmers_as_zeroes:
        return hs
    return [h for h in hs if h != 0]


HF_BY_MOLTYPE = {"dna": 1, "protein": 2, "dayhoff": 3, "hp": 4}


def seq_to_hashes_protein(seq, ksize, moltype="protein", seed=42, is_protein=True):
    """Hashes of every residue k-mer (ksize = residues; signature.rs:307-393): `seq` holds residues
    (is_protein) or DNA that is translated in six frames."""
    b = _as_bytes(seq)
    hf = HF_BY_MOLTYPE[moltype.lower()]
    n = lib().orc_seq_to_hashes_protein(b, len(b), ksize * 3, hf, seed, int(is_protein), None)
    out = np.zeros(max(n, 1), dtype=np.uint64)
    lib().orc_seq_to_hashes_protein(b, len(b), ksize * 3, hf, seed, int(is_protein), _ptr(out))
    return out[:n]


def translate_codon(codon):
    c = _as_bytes(codon).upper()
    if len(c) == 1:
        return "X"                       # encodings.rs:309-311
    if len(c) == 2:
        c += b"N"
    if len(c) != 3:
        raise ValueError(f"{len(c)}")
    return chr(lib().orc_translate_codon(c[0], c[1], c[2]))


def sketch_dna_bulk(buf, ksize, seed=42, max_hash=0, scaled=None, nthreads=1):
    """Sorted unique kept hashes of a whole buffer; any non-ACGT byte separates
    records (force=True semantics)."""
    if scaled is not None:
        max_hash = max_hash_for_scaled(scaled)
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    a = np.ascontiguousarray(a)
    out = u64p()
    n = lib().orc_sketch_dna_bulk(_ptr(a), a.size, ksize, seed, max_hash, nthreads, C.byref(out))
    res = np.ctypeslib.as_array(out, shape=(n,)).copy() if n else np.zeros(0, dtype=np.uint64)
    lib().orc_free(out)
    return res


def synth_dna(start, n, seed=42, record_len=0):
    out = np.empty(n, dtype=np.uint8)
    lib().orc_synth_dna(_ptr(out), start, n, seed, record_len)
    return out


def splitmix64(x):
    return lib().orc_splitmix64(x & 0xFFFFFFFFFFFFFFFF)


def intersection_size(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    u = u64(0)
    c = lib().orc_intersection_size(_ptr(a), a.size, _ptr(b), b.size, C.byref(u))
    return c, u.value


def md5sum_hashes(ksize, mins):
    a = np.ascontiguousarray(mins, dtype=np.uint64)
    out = C.create_string_buffer(33)
    lib().orc_md5sum_hashes(ksize, _ptr(a), a.size, out)
    return out.value.decode()


def md5_hex(data):
    out = C.create_string_buffer(33)
    lib().orc_md5_hex(data, len(data), out)
    return out.value.decode()


def make_csr(sketches):
    """list of sorted u64 arrays -> (hashes, offsets) CSR."""
    offsets = np.zeros(len(sketches) + 1, dtype=np.uint64)
    for i, s in enumerate(sketches):
        offsets[i + 1] = offsets[i] + len(s)
    hashes = (np.concatenate([np.asarray(s, dtype=np.uint64) for s in sketches])
              if len(sketches) and int(offsets[-1]) else np.zeros(0, dtype=np.uint64))
    return np.ascontiguousarray(hashes), offsets


def compare_all_pairs(hashes, offsets, nthreads=1):
    n = len(offsets) - 1
    common = np.zeros((n, n), dtype=np.uint32)
    jac = np.zeros((n, n), dtype=np.float64)
    hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    lib().orc_compare_all_pairs(_ptr(hashes), _ptr(offsets), n, _ptr(common), _ptr(jac), nthreads)
    return common, jac


def gather(query, hashes, offsets, threshold_bp=0, scaled=1000, max_rounds=None, nthreads=1):
    query = np.ascontiguousarray(query, dtype=np.uint64)
    hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    ndb = len(offsets) - 1
    if max_rounds is None:
        max_rounds = ndb
    idx = np.zeros(max(max_rounds, 1), dtype=np.uint64)
    isz = np.zeros(max(max_rounds, 1), dtype=np.uint64)
    r = lib().orc_gather_mt(_ptr(query), query.size, _ptr(hashes), _ptr(offsets), ndb, threshold_bp,
                            scaled, _ptr(idx), _ptr(isz), max_rounds, nthreads)
    return [(int(idx[i]), int(isz[i])) for i in range(r)]


def contained_by(common, denom, scaled):
    "minhash.py:819-841 on counts: |A ∩ B| / (|A| * bias(|A|, scaled)), clamped"
    return float(lib().orc_contained_by(int(common), int(denom), int(scaled)))


def max_containment(common, n_self, n_other, scaled):
    return float(lib().orc_max_containment(int(common), int(n_self), int(n_other), int(scaled)))


def avg_containment(common, n_self, n_other, scaled):
    return float(lib().orc_avg_containment(int(common), int(n_self), int(n_other), int(scaled)))


def containment_to_distance_point(containment, ksize):
    "distance_utils.py:276-283: the point estimate of containment_to_distance"
    return float(lib().orc_containment_to_distance_point(float(containment), int(ksize)))


def jaccard_to_distance(jaccard, ksize, n_unique_kmers):
    "distance_utils.py:349-407: (point estimate, lower bound of the approximation error); ValueError where the reference raises"
    err, bad = C.c_double(0.0), C.c_int(0)
    d = lib().orc_jaccard_to_distance(float(jaccard), int(ksize), int(n_unique_kmers), C.byref(err), C.byref(bad))
    if bad.value:
        raise ValueError("Error: varN <0.0!")
    return float(d), float(err.value)


def similarity_matrix(mhs, ignore_abundance=False, downsample=False, nthreads=1):
    "compare.py:14-64 over OracleMinHash objects (num rule, angular similarity, per-pair downsampling) -> f64 [n][n]"
    n = len(mhs)
    out = np.zeros((n, n), dtype=np.float64)
    ptrs = (C.c_void_p * max(n, 1))(*[m._p for m in mhs])
    e = C.c_uint32(0)
    lib().orc_similarity_matrix(ptrs, n, int(ignore_abundance), int(downsample), _ptr(out), C.byref(e), nthreads)
    if e.value:
        raise OracleError(e.value)
    return out


# --------------------------------------------------------------------------- #
# sketch object
# --------------------------------------------------------------------------- #
class OracleError(Exception):
    def __init__(self, code):
        super().__init__(f"oracle error code {code}")
        self.code = code


class OracleMinHash:
    """Vec-backed KmerMinHash restatement (src/core/src/sketch/minhash.rs:36-913)."""

    def __init__(self, n, ksize, *, scaled=0, seed=42, track_abundance=False, hash_function=1, _ptr_=None):
        self._p = _ptr_ if _ptr_ is not None else lib().orc_mh_new(
            scaled, ksize, hash_function, seed, int(track_abundance), n)
        self.ksize = ksize
        self.seed = seed
        self.num = n
        self.track_abundance = track_abundance

    def __del__(self):
        if getattr(self, "_p", None):
            lib().orc_mh_free(self._p)
            self._p = None

    def __len__(self):
        return lib().orc_mh_size(self._p)

    @property
    def max_hash(self):
        return lib().orc_mh_max_hash(self._p)

    @property
    def mins(self):
        n = len(self)
        if not n:
            return np.zeros(0, dtype=np.uint64)
        return np.ctypeslib.as_array(lib().orc_mh_mins(self._p), shape=(n,)).copy()

    @property
    def abunds(self):
        n = len(self)
        p = lib().orc_mh_abunds(self._p)
        if not p or not n:
            return None if not self.track_abundance else np.zeros(0, dtype=np.uint64)
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    def add_hash(self, h):
        lib().orc_mh_add_hash(self._p, h)

    def add_hash_with_abundance(self, h, a):
        lib().orc_mh_add_hash_with_abundance(self._p, h, a)

    def add_many(self, hs):
        a = np.ascontiguousarray(hs, dtype=np.uint64)
        lib().orc_mh_add_many(self._p, _ptr(a), a.size)

    def remove_many(self, hs):
        a = np.ascontiguousarray(hs, dtype=np.uint64)
        lib().orc_mh_remove_many(self._p, _ptr(a), a.size)

    def add_protein(self, seq):
        b = _as_bytes(seq)
        lib().orc_mh_add_protein(self._p, b, len(b))

    def add_sequence(self, seq, force=False):
        b = _as_bytes(seq)
        r = lib().orc_mh_add_sequence(self._p, b, len(b), int(force))
        if r < 0:
            i = -1 - r
            raise InvalidDNA(b[i:i + self.ksize].upper().decode("latin-1"))

    def merge(self, other):
        e = lib().orc_mh_merge(self._p, other._p)
        if e:
            raise OracleError(e)

    def _err(self):
        return C.c_uint32(0)

    def count_common(self, other, downsample=False):
        e = self._err()
        r = lib().orc_mh_count_common(self._p, other._p, int(downsample), C.byref(e))
        if e.value:
            raise OracleError(e.value)
        return r

    def intersection_and_union_size(self, other):
        e = self._err()
        u = u64(0)
        c = lib().orc_mh_intersection_size(self._p, other._p, C.byref(u), C.byref(e))
        if e.value:
            raise OracleError(e.value)
        return c, u.value

    def jaccard(self, other):
        e = self._err()
        r = lib().orc_mh_jaccard(self._p, other._p, C.byref(e))
        if e.value:
            raise OracleError(e.value)
        return r

    def similarity(self, other, ignore_abundance=False, downsample=False):
        e = self._err()
        r = lib().orc_mh_similarity(self._p, other._p, int(ignore_abundance), int(downsample), C.byref(e))
        if e.value:
            raise OracleError(e.value)
        return r

    def angular_similarity(self, other):
        e = self._err()
        r = lib().orc_mh_angular_similarity(self._p, other._p, C.byref(e))
        if e.value:
            raise OracleError(e.value)
        return r

    def downsample_scaled(self, scaled):
        e = self._err()
        p = lib().orc_mh_downsample_scaled(self._p, scaled, C.byref(e))
        if e.value:
            raise OracleError(e.value)
        return OracleMinHash(self.num, self.ksize, seed=self.seed, track_abundance=self.track_abundance, _ptr_=p)

    def md5sum(self):
        out = C.create_string_buffer(33)
        lib().orc_mh_md5sum(self._p, out)
        return out.value.decode()


# --------------------------------------------------------------------------- #
# fixture readers used by the tests (plain stdlib; independent of the product)
# --------------------------------------------------------------------------- #
def read_fasta(path):
    """Yield (name, sequence) with newlines stripped (screed semantics used at
    src/sourmash/command_sketch.py:697,746)."""
    op = gzip.open if str(path).endswith(".gz") else open
    name, chunks = None, []
    with op(path, "rt") as fh:
        for line in fh:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if name is not None:
                    yield name, "".join(chunks)
                name, chunks = line[1:], []
            elif name is not None:
                chunks.append(line)
    if name is not None:
        yield name, "".join(chunks)


def read_sig_json(path):
    """-> list of dict(name, filename, ksize, num, seed, max_hash, molecule, md5sum, mins, abundances)"""
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "rt") as fh:
        data = json.load(fh)
    out = []
    for rec in data:
        for sk in rec["signatures"]:
            d = dict(sk)
            d["name"] = rec.get("name", "")
            d["filename"] = rec.get("filename", "")
            d["mins"] = np.array(sk["mins"], dtype=np.uint64)
            if "abundances" in sk:
                d["abundances"] = np.array(sk["abundances"], dtype=np.uint64)
            out.append(d)
    return out
