"""CPU check of the sketch kernel's per-lane byte plumbing
(sourmash_amd/csrc/kmer_core.hpp compiled for the host with v_perm_b32 /
v_alignbyte_b32 emulated) against the oracle.  No GPU needed."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "kmer_core_emul.cpp")
SO = os.path.join(HERE, "native", "libkmer_core_emul.so")
HDRS = [os.path.join(HERE, "..", "sourmash_amd", "csrc", h) for h in ("kmer_core.hpp", "murmur3.hpp")]


@pytest.fixture(scope="module")
def emul():
    newest = max(os.path.getmtime(p) for p in [SRC] + HDRS)
    if not os.path.exists(SO) or os.path.getmtime(SO) < newest:
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", SO, SRC])
    lib = C.CDLL(SO)
    lib.emul_sketch.restype = C.c_uint64
    lib.emul_sketch.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64,
                                C.c_void_p, C.c_uint64]

    def run(buf, k, p=16, seed=42, thr=2**64 - 1):
        a = np.frombuffer(bytes(buf), dtype=np.uint8)
        out = np.zeros(max(len(a), 1), dtype=np.uint64)
        n = lib.emul_sketch(a.ctypes.data, len(a), k, p, seed, thr, out.ctypes.data, len(out))
        assert n != 2**64 - 1, "k/p combination not instantiated"
        return np.sort(out[:n])
    run.lib = lib
    return run


def _oracle_all(buf, k, seed=42, thr=2**64 - 1):
    hs = oracle.seq_to_hashes(bytes(buf), k, seed=seed, force=True)   # drops bad k-mers and zeros
    return np.sort(np.array([h for h in hs if 1 <= h <= thr], dtype=np.uint64))


def _rand_dna(rng, n, alphabet=b"ACGT"):
    return bytes(rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=n))


def test_every_dispatched_ksize_matches_oracle(emul):
    "sketch.hip / sketch_long.hip instantiate the register-window kernel for every k = 1 .. 88 (P = 16; the template itself holds to k = 128): each one against the oracle"
    rng = np.random.default_rng(2024)
    for k in range(1, 129):
        for n in (k - 1, k, k + 17, 700):
            s = _rand_dna(rng, n, alphabet=b"ACGTacgtN" if n == 700 else b"ACGT")
            assert np.array_equal(emul(s, k, 16), _oracle_all(s, k)), (k, n)


@pytest.mark.parametrize("k,p", [(31, 16), (31, 8), (21, 16), (51, 16), (4, 16), (3, 16), (5, 16), (10, 16),
                                 (16, 16), (32, 16), (17, 8), (15, 4), (33, 16), (63, 16), (1, 16), (8, 16), (9, 16)])
def test_all_kmers_match_oracle(emul, k, p):
    rng = np.random.default_rng(k * 100 + p)
    for n in (0, 1, k - 1, k, k + 1, 100, 1000, 4097):
        if n < 0:
            continue
        s = _rand_dna(rng, n)
        got = emul(s, k, p)
        want = _oracle_all(s, k)
        assert np.array_equal(got, want), (k, p, n)


def test_invalid_lowercase_and_palindromes(emul):
    rng = np.random.default_rng(7)
    # N every 89th (src/core/benches/compute.rs:22-26), IUPAC, lowercase, separators, NUL, high bytes
    s = bytearray(_rand_dna(rng, 20000, b"ACGTacgt"))
    for i in range(1, len(s), 89):
        s[i] = ord("N")
    for i, c in zip(range(500, 20000, 997), b"RYKMSWBDHVnU\n>\x00\xff{[@`"):
        s[i] = c
    for k in (31, 21, 51, 4):
        assert np.array_equal(emul(bytes(s), k), _oracle_all(bytes(s), k)), k
    # long runs where forward == prefix of reverse complement (tie-break beyond the first 8 bytes)
    pal = b"ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT" * 5 + b"AAAAAAAAAAAAAAAATTTTTTTTTTTTTTTT" * 4
    pal += b"ATATATATATATATATATATATATATATATATATAT" + b"GCGCGCGCGCGCGCGCGCGCGCGCGCGCGCGCGC" + b"AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA"
    for k in (31, 32, 21, 16, 10, 51, 33):
        assert np.array_equal(emul(pal, k), _oracle_all(pal, k)), k


def test_threshold_and_seed(emul):
    rng = np.random.default_rng(11)
    s = _rand_dna(rng, 50000)
    thr = oracle.max_hash_for_scaled(100)
    got = emul(s, 31, 16, 42, thr)
    assert np.array_equal(got, _oracle_all(s, 31, 42, thr)) and 300 < len(got) < 700
    assert np.array_equal(emul(s, 21, 16, 7, thr), _oracle_all(s, 21, 7, thr))


def test_open_form_of_the_hash(emul):
    "fmix64 split around its last multiply: same value, top dword known to within the carry"
    emul.lib.emul_open_form_violations.restype = C.c_uint64
    emul.lib.emul_open_form_violations.argtypes = [C.c_uint64, C.c_uint64]
    assert emul.lib.emul_open_form_violations(2_000_000, 42) == 0


def test_threshold_boundaries(emul):
    """The early reject works on the top dword of the hash: thresholds sitting exactly on, just below and just above
    real hash values, and on dword boundaries, must keep exactly the hashes <= thr."""
    rng = np.random.default_rng(11)
    s = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 6000))
    allh = _oracle_all(s, 31)
    assert len(allh) > 5000
    picks = [allh[0], allh[1], allh[len(allh) // 2], allh[-1]]
    thrs = set()
    for h in map(int, picks):
        t = h >> 32
        thrs.update([h, h - 1, h + 1, (t << 32), (t << 32) - 1, (t << 32) | 0xffffffff, ((t + 1) << 32),
                     ((t - 1) << 32) | 0xffffffff if t else 0])
    thrs.update([1, 0xffffffff, 1 << 32, (1 << 32) - 1, (0xfffffffe << 32) | 5, (0xffffffff << 32), 2**64 - 2])
    for thr in sorted(x for x in thrs if 0 < x < 2**64):
        got = emul(s, 31, 16, 42, thr)
        assert np.array_equal(got, allh[allh <= np.uint64(thr)]), hex(thr)
