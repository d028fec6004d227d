"""The many-thread gzip reader of the ingest path (csrc/pargz.hpp) on the host alone: whatever the file looks like, the bytes it
returns are the bytes zlib's sequential stream returns (same length, same CRC-32, byte for byte where compared), or the call fails
the way the sequential reader fails.  The reference reads .gz input through one zlib stream (screed / niffler:
src/sourmash/command_sketch.py:697, src/core/benches/compute.rs:35-38)."""
import ctypes as C
import gzip
import os
import zlib

import numpy as np
import pytest

from sourmash_amd._lowlevel import lib


def _gunzip(path, threads=8, span=256 << 10, keep=True):
    size_hint = 64 << 20
    out = (C.c_uint8 * size_hint)() if keep else None
    crc, par = C.c_uint32(0), C.c_bool(False)
    lib.sourmash_err_clear()
    n = lib.smgpu_gunzip_file(str(path).encode(), threads, span, out, size_hint if keep else 0, C.byref(crc), C.byref(par))
    code = lib.sourmash_err_get_last_code()
    lib.sourmash_err_clear()
    return n, (bytes(out[:n]) if keep and code == 0 else None), crc.value, par.value, code


def _fasta(n_bases, seed, line=70, lower=False, records=1):
    rng = np.random.default_rng(seed)
    parts = []
    for r in range(records):
        seq = rng.choice(np.frombuffer(b"acgt" if lower else b"ACGT", dtype=np.uint8), n_bases // records)
        body = b"\n".join(bytes(seq[i:i + line]) for i in range(0, len(seq), line))
        parts.append(b">record_%d a description\n" % r + body + b"\n")
    return b"".join(parts)


def _repetitive(n_bases, seed):
    "long exact repeats: matches that reach 32 KB back, so bytes copied out of the unknown window travel far into a span"
    rng = np.random.default_rng(seed)
    unit = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 20_000))
    out = bytearray(b">rep\n")
    while len(out) < n_bases:
        out += unit
        out += bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(rng.integers(10, 3000))))
    return bytes(out)


def _fastq(n_reads, seed):
    rng = np.random.default_rng(seed)
    seqs = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), (n_reads, 100), p=[.249, .249, .249, .249, .004])
    quals = rng.integers(33, 74, (n_reads, 100), dtype=np.uint8)
    return b"".join(b"@read%d/1\n%s\n+\n%s\n" % (i, bytes(seqs[i]), bytes(quals[i])) for i in range(n_reads))


@pytest.mark.parametrize("level", [1, 6, 9])
def test_parallel_equals_sequential_on_text(tmp_path, level):
    for name, data in (("fasta", _fasta(6_000_000, 1)), ("lower_multi", _fasta(5_000_000, 2, lower=True, records=40)),
                       ("repeats", _repetitive(12_000_000, 3)), ("fastq", _fastq(20_000, 4))):
        p = tmp_path / f"{name}_{level}.gz"
        with gzip.open(p, "wb", compresslevel=level) as f:
            f.write(data)
        n1, out1, crc1, par1, code1 = _gunzip(p, threads=1)
        assert code1 == 0 and not par1 and out1 == data
        size = os.path.getsize(p)
        for threads, span in ((8, size // 13), (3, size // 6), (8, size // 29)):      # spans sized to the file: always several of them
            n, out, crc, par, code = _gunzip(p, threads=threads, span=span)
            assert code == 0, (name, level, threads, span)
            assert par, (name, level, span, "the parallel form should carry a plain text member cut into this many spans")
            assert n == len(data) and crc == zlib.crc32(data) and out == data, (name, level, threads, span)


def test_small_files_and_other_framings_take_the_sequential_stream(tmp_path):
    data = _fasta(3_000_000, 7)
    small = tmp_path / "small.gz"
    with gzip.open(small, "wb") as f:
        f.write(data[:2000])
    n, out, _, par, code = _gunzip(small)
    assert code == 0 and not par and out == data[:2000]
    empty = tmp_path / "empty.gz"
    with gzip.open(empty, "wb") as f:
        pass
    assert _gunzip(empty)[:2] == (0, b"")
    # two members back to back (what bgzip / `cat a.gz b.gz` produce): the first member ends inside a span -> the stream
    # is handed to zlib, which reads members one after the other
    multi = tmp_path / "multi.gz"
    with open(multi, "wb") as f:
        f.write(gzip.compress(data[:1_500_000], 6))
        f.write(gzip.compress(data[1_500_000:], 6))
    n, out, crc, par, code = _gunzip(multi, span=128 << 10)
    assert code == 0 and not par and out == data and crc == zlib.crc32(data)
    # stored blocks only (level 0): there is no dynamic block to find
    stored = tmp_path / "stored.gz"
    with gzip.open(stored, "wb", compresslevel=0) as f:
        f.write(data)
    n, out, _, par, code = _gunzip(stored, span=128 << 10)
    assert code == 0 and not par and out == data


def test_input_that_is_not_7_bit_text_is_still_right(tmp_path):
    rng = np.random.default_rng(9)
    text = _fasta(2_000_000, 11)
    # bytes >= 0x80 from the very start / only deep inside the file
    blob = bytes(rng.integers(0, 256, 600_000, dtype=np.uint8)) * 3
    for name, data in (("binary", blob + text), ("late", text + b">x\n" + bytes([0xC3, 0xA9]) * 50_000 + text)):
        p = tmp_path / f"{name}.gz"
        with gzip.open(p, "wb", compresslevel=6) as f:
            f.write(data)
        n, out, crc, par, code = _gunzip(p, span=128 << 10)
        # the reader notices BEFORE it delivers the span that holds such a byte (a data byte >= 0x80 reads alike in both marker
        # passes; the third pass against zeros tells it from a marker) and carries on with the plain stream: never an error,
        # never other bytes -- the reference reads such files through one zlib stream without complaint
        assert code == 0 and out == data and not par and crc == zlib.crc32(data), name


def test_utf8_in_a_late_header_followed_by_stored_blocks(tmp_path):
    """text, two UTF-8 bytes three megabytes in, text, 900 KB of stored blocks (no block start to be found there),
    text.  Round 3 delivered 'caf' + the window's bytes for the two data bytes, met the structural fallback afterwards and
    skipped the delivered prefix unchecked: return code 0 and wrong bytes.  Now the span with the data bytes is refused before
    delivery and a skipped prefix is CRC-checked against what was delivered."""
    rng = np.random.default_rng(21)
    text = _fasta(3_000_000, 31)
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    raw = co.compress(text + b">caf" + bytes([0xC3, 0xA9]) + b" strain\n" + _fasta(1_500_000, 32))
    raw += co.flush(zlib.Z_FULL_FLUSH)
    noise = bytes(rng.integers(0, 128, 900_000, dtype=np.uint8))        # incompressible 7-bit bytes: zlib stores them
    raw += co.compress(noise)
    raw += co.flush(zlib.Z_FULL_FLUSH)
    tail = _fasta(2_000_000, 33)
    raw += co.compress(tail) + co.flush()
    data = gzip.decompress(raw)
    assert bytes([0xC3, 0xA9]) in data
    p = tmp_path / "cafe.gz"
    p.write_bytes(raw)
    for threads, span in ((8, 128 << 10), (4, 300 << 10), (8, 1 << 20)):
        if os.path.getsize(p) < 4 * span:
            continue
        n, out, crc, par, code = _gunzip(p, threads=threads, span=span)
        assert code == 0 and n == len(data) and out == data and crc == zlib.crc32(data), (threads, span)
        assert not par


def test_position_code_keeps_the_ambiguous_markers_out_of_reach():
    """The two marker bytes of a window position are 0x80 | code >> 8 and code & 0xff.  They coincide for 128 codes; the code is
    an involution on 0..32767 that hands exactly those codes to window positions 0..127 -- further back than zlib-family
    encoders ever reach (32768 - 262) -- so that text files never need the third pass."""
    codes = [lib.smgpu_gunzip_position_code(p) for p in range(32768)]
    assert sorted(codes) == list(range(32768))
    assert all(codes[codes[p]] == p for p in range(32768))
    ambiguous = [p for p in range(32768) if (0x80 | (codes[p] >> 8)) == (codes[p] & 0xff)]
    assert ambiguous == list(range(128))


def test_corrupt_and_truncated_files_fail(tmp_path):
    data = _fasta(4_000_000, 13)
    raw = gzip.compress(data, 6)
    cut = tmp_path / "cut.gz"
    cut.write_bytes(raw[: len(raw) * 2 // 3])
    assert _gunzip(cut, span=128 << 10)[4] != 0
    assert _gunzip(cut, threads=1)[4] != 0
    flipped = bytearray(raw)
    flipped[len(raw) // 2] ^= 0x5A
    bad = tmp_path / "bad.gz"
    bad.write_bytes(bytes(flipped))
    n, out, crc, par, code = _gunzip(bad, span=128 << 10)
    assert code != 0 or out == data                         # (a flipped bit inside a stored literal run could only be caught by the CRC: it is)
    assert not os.path.exists(tmp_path / "nope.gz") and _gunzip(tmp_path / "nope.gz")[4] != 0
