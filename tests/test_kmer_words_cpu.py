"""CPU check of the long-k sketch kernel's per-position logic (sourmash_amd/csrc/kmer_words.hpp compiled for the host, staged the
way sketch_words.hip stages a stretch) against the oracle (signature.rs:246-306).  No GPU needed."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "kmer_words_emul.cpp")
SO = os.path.join(HERE, "native", "libkmer_words_emul.so")
HDRS = [os.path.join(HERE, "..", "sourmash_amd", "csrc", h) for h in ("kmer_words.hpp", "kmer_core.hpp", "murmur3.hpp")]


@pytest.fixture(scope="module")
def emul():
    newest = max(os.path.getmtime(p) for p in [SRC] + HDRS)
    if not os.path.exists(SO) or os.path.getmtime(SO) < newest:
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", SO, SRC])
    lib = C.CDLL(SO)
    lib.emul_words_dense.restype = C.c_uint64
    lib.emul_words_dense.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64]

    def run(buf, k, tile=4096, skip=0, seed=42):
        a = np.frombuffer(bytes(buf), dtype=np.uint8)
        n = max(len(a) - k + 1, 0)
        out = np.zeros(max(n, 1), dtype=np.uint64)
        lib.emul_words_dense(a.ctypes.data if len(a) else None, len(a), k, tile, skip, seed, out.ctypes.data, n)
        return out[:n]
    return run


def _oracle_dense(buf, k, seed=42):
    "one hash per start position, 0 where the k-mer holds a byte outside ACGT (ffi/minhash.rs:63-99 with bad_kmers_as_zeroes)"
    if len(buf) < k:
        return np.zeros(0, dtype=np.uint64)
    return np.array(oracle.seq_to_hashes(bytes(buf), k, seed=seed, force=True, bad_kmers_as_zeroes=True), dtype=np.uint64)


def _rand_dna(rng, n, alphabet=b"ACGT"):
    return bytes(rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=n))


def test_every_tail_length_and_alignment_matches_the_oracle(emul):
    rng = np.random.default_rng(5)
    for k in list(range(129, 162)) + [176, 191, 192, 193, 200, 255, 256, 257, 300, 511, 1000]:
        for n in (k, k + 1, k + 37):
            s = _rand_dna(rng, n)
            for skip in (0, 5):
                assert np.array_equal(emul(s, k, tile=32, skip=skip), _oracle_dense(s, k)), (k, n, skip)


def test_stretch_seams_invalid_bytes_and_case(emul):
    rng = np.random.default_rng(6)
    s = bytearray(_rand_dna(rng, 9000, b"ACGTacgt"))
    for i in range(700, 9000, 1307):
        s[i] = ord("N")
    for i, c in zip(range(100, 9000, 2111), b"RY\n\x00\xff"):
        s[i] = c
    for k in (129, 200, 256, 333):
        want = _oracle_dense(bytes(s), k)
        assert want.any()
        for tile, skip in ((4096, 0), (4096, 11), (64, 3), (160, 0)):
            assert np.array_equal(emul(bytes(s), k, tile=tile, skip=skip), want), (k, tile, skip)


def test_palindromes_and_long_ties(emul):
    "forward == reverse complement for many blocks: the tie walk goes past block 0, into the tail, or never ends (a palindrome)"
    half = b"ACGGTCATTGCA" * 20
    rc = bytes(half[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA")))
    pal = half + rc                       # an exact palindrome of 480 bases
    rng = np.random.default_rng(8)
    near = bytearray(pal)
    near[300] = ord("A") if near[300] != ord("A") else ord("C")     # differs deep inside
    poly = b"A" * 200 + b"T" * 200 + _rand_dna(rng, 50) + b"AT" * 150 + b"GC" * 150
    for s in (pal, bytes(near), poly):
        for k in (129, 160, 161, 200, 240, 300):
            if len(s) >= k:
                assert np.array_equal(emul(s, k, tile=128), _oracle_dense(s, k)), (k, len(s))


def test_seeds(emul):
    rng = np.random.default_rng(9)
    s = _rand_dna(rng, 1000)
    for seed in (0, 1, 43, 2**32 - 1):
        assert np.array_equal(emul(s, 150, seed=seed), _oracle_dense(s, 150, seed=seed)), seed


def test_very_long_kmers(emul):
    "k in the thousands, up to the kernel's limit: many blocks, a stretch dominated by its halo"
    rng = np.random.default_rng(10)
    s = bytearray(_rand_dna(rng, 61_500))
    for k in (4999, 60_000):
        assert np.array_equal(emul(bytes(s), k), _oracle_dense(bytes(s), k)), k
    s[30_000] = ord("N")
    assert np.array_equal(emul(bytes(s), 20_000), _oracle_dense(bytes(s), 20_000))
