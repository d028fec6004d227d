"""CPU check of the residue-window kernel's per-lane logic (sourmash_amd/csrc/residue_core.hpp compiled for the host) against the
naive definition and the oracle (signature.rs:307-393).  No GPU needed."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "residue_core_emul.cpp")
SO = os.path.join(HERE, "native", "libresidue_core_emul.so")
HDRS = [os.path.join(HERE, "..", "sourmash_amd", "csrc", h) for h in ("residue_core.hpp", "murmur3.hpp", "residues.hpp", "translate_core.hpp")]


@pytest.fixture(scope="module")
def emul():
    newest = max(os.path.getmtime(p) for p in [SRC] + HDRS)
    if not os.path.exists(SO) or os.path.getmtime(SO) < newest:
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", SO, SRC])
    lib = C.CDLL(SO)
    for f in (lib.emul_residue_windows, lib.naive_residue_windows):
        f.restype = C.c_uint64
        f.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]

    def run(fn, buf, k, seed=42):
        a = np.frombuffer(bytes(buf), dtype=np.uint8)
        cap = max(len(a), 1)
        st, hs = np.zeros(cap, dtype=np.uint64), np.zeros(cap, dtype=np.uint64)
        n = fn(a.ctypes.data if len(a) else None, len(a), k, seed, st.ctypes.data, hs.ctypes.data, cap)
        assert n != 2**64 - 1
        order = np.argsort(st[:n], kind="stable")
        return st[:n][order], hs[:n][order]
    run.fast = lambda buf, k, seed=42: run(lib.emul_residue_windows, buf, k, seed)
    run.naive = lambda buf, k, seed=42: run(lib.naive_residue_windows, buf, k, seed)
    return run


AA = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWYX*", dtype=np.uint8)


def test_every_window_length_matches_the_naive_definition_and_the_oracle(emul):
    rng = np.random.default_rng(3)
    for k in list(range(1, 80)):
        for n in (k - 1, k, k + 1, k + 7, k + 8, k + 9, 257):
            if n < 0:
                continue
            s = bytes(rng.choice(AA, size=n))
            fs, fh = emul.fast(s, k)
            ns, nh = emul.naive(s, k)
            assert np.array_equal(fs, ns) and np.array_equal(fh, nh), (k, n)
            if n >= k and k <= 60:
                want = oracle.seq_to_hashes_protein(s.replace(b"*", b"A").replace(b"X", b"A"), k, "protein")
                got = emul.fast(s.replace(b"*", b"A").replace(b"X", b"A"), k)[1]
                assert np.array_equal(got, want), (k, n)


def test_separators_cut_windows_exactly(emul):
    """0xFF bytes (the marks between the six translations of a record, protein.hip) at every alignment: a window exists iff none
    of its k bytes is a separator -- also when the separator sits in the tail bytes, in a neighbouring lane's word, first, last."""
    rng = np.random.default_rng(5)
    for k in (1, 2, 7, 8, 9, 10, 15, 16, 17, 24, 31, 32, 33, 42, 48, 63, 64, 65, 79):
        for trial in range(12):
            n = int(rng.integers(k, 400))
            s = bytearray(rng.choice(AA, size=n).tobytes())
            for p in rng.integers(0, n, size=int(rng.integers(1, 5))):
                s[int(p)] = 0xff
            if trial % 3 == 0:
                s[0] = 0xff
                s[-1] = 0xff
            fs, fh = emul.fast(bytes(s), k)
            ns, nh = emul.naive(bytes(s), k)
            assert np.array_equal(fs, ns) and np.array_equal(fh, nh), (k, trial)


def test_seed_and_high_bytes(emul):
    rng = np.random.default_rng(9)
    s = bytes(rng.integers(0, 255, size=1000, dtype=np.uint8))      # any byte value but 0xFF
    for k in (5, 12, 20, 37):
        for seed in (0, 1, 42, 2**32 - 1):
            assert np.array_equal(emul.fast(s, k, seed)[1], emul.naive(s, k, seed)[1]), (k, seed)


def test_translate_tables_equal_the_scalar_functions(emul):
    "protein.hip's translate kernel reads byte -> code and code triple -> residue tables: every byte triple, three alphabets, both strands"
    lib = C.CDLL(SO)
    lib.check_translate_tables.restype = C.c_uint64
    assert lib.check_translate_tables() == 0


def test_translate_words_equal_the_per_byte_definition_and_the_oracle(emul):
    """protein.hip's translate kernel gives every lane one aligned word of the six-segment output (translate_core.hpp): against the
    per-byte definition for every length 3 .. 80 and longer ones (segment starts at every alignment, tails, separators), three
    alphabets, junk bytes; and the hashes of the translated windows against the oracle's walk over the DNA itself."""
    lib = C.CDLL(SO)
    lib.emul_translate.restype = C.c_uint64
    lib.emul_translate.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(17)
    alphabet = np.frombuffer(b"ACGTacgtNnRY\x00\xfe", dtype=np.uint8)
    for n in list(range(3, 81)) + [255, 256, 257, 1000, 4099]:
        for hf in (2, 3, 4):
            s = rng.choice(alphabet[:8] if n % 3 else alphabet, size=n).astype(np.uint8)
            total = sum((n - f) // 3 + 1 for f in (0, 0, 1, 1, 2, 2))
            fast, naive = np.zeros(total + 8, dtype=np.uint8), np.zeros(total + 8, dtype=np.uint8)
            got = lib.emul_translate(s.ctypes.data, n, hf, fast.ctypes.data, naive.ctypes.data)
            assert got == total and np.array_equal(fast[:total], naive[:total]), (n, hf)
    # end to end on the host: translated windows hashed by the register-window lane == the oracle on the DNA
    s = rng.choice(alphabet[:8], size=3000).astype(np.uint8)
    total = sum((3000 - f) // 3 + 1 for f in (0, 0, 1, 1, 2, 2))
    fast, naive = np.zeros(total + 8, dtype=np.uint8), np.zeros(total + 8, dtype=np.uint8)
    lib.emul_translate(s.ctypes.data, 3000, 2, fast.ctypes.data, naive.ctypes.data)
    hashes = emul.fast(bytes(fast[:total]), 10)[1]
    want = oracle.seq_to_hashes_protein(bytes(s), 10, "protein", is_protein=False)
    assert np.array_equal(hashes, want)
