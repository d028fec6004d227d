"""Host pieces of the per-record path that need no GPU: the C method behind MinHash.add_sequence (csrc/fastcall.c) --
that it is built, bound to the loaded library, parses arguments like the Python signature it replaces and routes the
library's error through the usual exception classes -- and the vectorised validity scan that decides where
add_sequence(force=False) must raise (capi.cpp: first_invalid_byte; encodings.rs:370-377 of the reference)."""
import ctypes as C
import random

import pytest

import sourmash_amd
from sourmash_amd import minhash as mhmod
from sourmash_amd._lowlevel import lib


def test_c_method_is_built_and_bound():
    assert mhmod._fastcall is not None, "sourmash_amd/_fastcall*.so is not built (make -C sourmash_amd/csrc)"
    assert mhmod._AddSequence is mhmod._fastcall.AddSequenceBase
    assert type(sourmash_amd.MinHash.add_sequence).__name__ == "method_descriptor"
    assert "force=False" in sourmash_amd.MinHash.add_sequence.__text_signature__
    assert sourmash_amd.MinHash.add_sequence.__doc__.startswith("Add every k-mer of a DNA sequence")
    # the ctypes twin keeps the same signature
    assert mhmod._AddSequencePython.add_sequence.__defaults__ == (False,)


def test_argument_errors_match_the_python_signature():
    mh = sourmash_amd.MinHash(0, 21, scaled=1000)
    for args, kwargs in (((), {}), ((b"A", True, 3), {}), ((b"A",), {"foo": 1}), ((b"A",), {"sequence": b"C"}),
                         ((b"A", True), {"force": False})):
        with pytest.raises(TypeError):
            mh.add_sequence(*args, **kwargs)
    for wrong in (3.5, None, ["ACGT"], object()):
        with pytest.raises(TypeError, match="string-like"):
            mh.add_sequence(wrong)
    with pytest.raises(ValueError):
        mh.add_sequence(256)                                      # bytes([256]) in the ctypes twin
    closed = sourmash_amd.MinHash(0, 21, scaled=1000)
    ptr, closed._objptr = closed._objptr, None
    try:
        with pytest.raises(RuntimeError, match="closed"):
            closed.add_sequence(b"ACGT" * 10)
    finally:
        closed._objptr = ptr


def test_library_errors_arrive_as_the_usual_exceptions():
    if sourmash_amd.gpu_available():
        pytest.skip("a GPU is present: the call succeeds (covered by tests/test_gpu_deferred.py)")
    mh = sourmash_amd.MinHash(0, 21, scaled=1000)
    with pytest.raises(sourmash_amd.exceptions.SourmashError, match="no HIP device"):
        mh.add_sequence("ACGT" * 10)                              # no CPU path for hashing: the library says so
    with pytest.raises(sourmash_amd.exceptions.SourmashError, match="no HIP device"):
        mhmod._AddSequencePython.add_sequence(mh, "ACGT" * 10)
    mh.add_sequence("ACGT")                                       # shorter than k: nothing to do, no device needed


def _ref(b):
    ok = set(b"ACGTacgt")
    for i, c in enumerate(b):
        if c not in ok:
            return i
    return C.c_size_t(-1).value


def test_validity_scan_every_byte_value_at_every_position():
    f = lib.smgpu_first_invalid_dna_byte
    base = b"ACGTacgt" * 8
    for length in range(0, 50):                                   # the 16-byte steps, their tail, and every boundary between
        for pos in range(length):
            for v in range(256):
                b = bytearray(base[:length])
                b[pos] = v
                assert f(bytes(b), length) == _ref(b), (length, pos, v)
    random.seed(1)
    alphabet = b"ACGTacgtNn\x00\xff\x80AAAACCCC"
    for _ in range(5000):
        length = random.randint(0, 300)
        b = bytes(random.choice(alphabet) for _ in range(length))
        assert f(b, length) == _ref(b)
    assert f(None, 0) == C.c_size_t(-1).value
