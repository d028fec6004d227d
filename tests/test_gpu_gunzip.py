"""gzip members inflated on the GPU (csrc/gunzip.hip, csrc/gunzip.hpp) against zlib, and the .gz ingest on both of its paths.
The same cases as tests/test_inflate_core_cpu.py runs through the host build of the shared decoder.  Run with -m gpu."""
import gzip
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

from conftest import ROOT
from test_inflate_core_cpu import CASES, fasta, fastq, gz

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sk():
    import torch  # noqa: F401
    import sourmash_amd
    assert sourmash_amd.gpu_available()
    from sourmash_amd import sketch
    return sketch


def _write(tmp_path, name, blob):
    p = tmp_path / name
    p.write_bytes(blob)
    return str(p)


@pytest.mark.parametrize("name", sorted(CASES))
def test_members_of_text_inflate_to_what_zlib_gives(sk, tmp_path, name):
    rng = np.random.default_rng(abs(hash(name)) % 2**32)
    blob = CASES[name](rng)
    (got,), stats = sk.gunzip_files([_write(tmp_path, "a.gz", blob)])
    assert got == zlib.decompress(blob, 31)
    assert stats["runs"] >= 2


def test_small_and_odd_members_in_one_batch(sk, tmp_path):
    rng = np.random.default_rng(5)
    datas = [b"", b"A", b"ACGT" * 3, b">x\nACGTACGTAC\n", bytes(rng.integers(0, 256, size=70_000, dtype=np.uint8)),
             b"A" * 1_000_000, (b"ACGTTGCA" * 40 + b"\n") * 5000, bytes(rng.integers(128, 256, size=300_000, dtype=np.uint8) % 7 + 200)]
    paths, want = [], []
    for i, data in enumerate(datas):
        for level in (0, 1, 6, 9):
            paths.append(_write(tmp_path, f"f{i}_{level}.gz", gz(data, level)))
            want.append(data)
    got, stats = sk.gunzip_files(paths)
    for p, g, w in zip(paths, got, want):
        if len(open(p, "rb").read()) < 26:
            assert g is None                                               # (shorter than any gzip file the framing parser takes)
        else:
            assert g == w, p


def test_flush_points_strategies_windows(sk, tmp_path):
    rng = np.random.default_rng(11)
    data = fasta(rng, 900_000)
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    blob = b""
    for i in range(0, len(data), 130_000):
        blob += co.compress(data[i:i + 130_000]) + co.flush(zlib.Z_SYNC_FLUSH if (i // 130_000) % 2 else zlib.Z_FULL_FLUSH)
    blob += co.flush()
    paths, want = [_write(tmp_path, "flush.gz", blob)], [data]
    for strategy in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
        paths.append(_write(tmp_path, f"s{strategy}.gz", gz(data[:300_000], 6, strategy=strategy)))
        want.append(data[:300_000])
    for wbits, memlevel in ((9 + 16, 8), (12 + 16, 1), (15 + 16, 9), (15 + 16, 1)):
        co = zlib.compressobj(6, zlib.DEFLATED, wbits, memlevel)
        paths.append(_write(tmp_path, f"w{wbits}_{memlevel}.gz", co.compress(data[:500_000]) + co.flush()))
        want.append(data[:500_000])
    got, _ = sk.gunzip_files(paths)
    assert got == want


def test_damaged_and_multi_member_files_are_refused_not_misread(sk, tmp_path):
    rng = np.random.default_rng(13)
    data = fasta(rng, 400_000)
    blob = gz(data, 6)
    bad = bytearray(blob)
    bad[len(bad) // 2] ^= 0x55
    paths = [_write(tmp_path, "two.gz", blob + blob), _write(tmp_path, "flip.gz", bytes(bad)),
             _write(tmp_path, "cut.gz", blob[:len(blob) // 2] + blob[-8:]), _write(tmp_path, "text.gz", b"not a gzip file at all, but long enough"),
             _write(tmp_path, "good.gz", blob)]
    got, _ = sk.gunzip_files(paths, capacity=4 * len(data))
    assert got[0] is None and got[2] is None and got[3] is None
    assert got[1] is None                                              # the CRC-32 (or the chain) catches the flipped byte
    assert got[4] == data                                              # ... and the good member of the same batch is served


def test_a_larger_member_and_its_stages(sk, tmp_path):
    "120 MB of FASTA at level 1 (the shape of bench.py's ingest_gz, smaller): thousands of runs, tails in order, 64 KB pieces, CRC chunks"
    rng = np.random.default_rng(21)
    data = fasta(rng, 120_000_000)
    blob = gz(data, 1)
    (got,), stats = sk.gunzip_files([_write(tmp_path, "big.gz", blob)])
    assert len(got) == len(data) and zlib.crc32(got) == zlib.crc32(data) and got == data
    assert stats["runs"] > 500 and stats["candidates"] >= stats["runs"]


def test_ingest_takes_the_device_path_and_gives_the_same_sketch(sk, tmp_path):
    rng = np.random.default_rng(31)
    data = fasta(rng, 3_000_000) + fasta(rng, 50_000)
    plain, packed = _write(tmp_path, "g.fa", data), _write(tmp_path, "g.fa.gz", gz(data, 6))
    fq = fastq(rng, 20_000)
    fq_plain, fq_packed = _write(tmp_path, "r.fq", fq), _write(tmp_path, "r.fq.gz", gz(fq, 6))
    two = _write(tmp_path, "two.fa.gz", gz(data, 6) + gz(data[:100_000], 6))      # two members: the host inflater's
    before = sk.gunzip_counters()
    for a, b in ((plain, packed), (fq_plain, fq_packed)):
        sa, = sk.sketch_file(a, "k=21,k=31,scaled=100,abund")
        sb, = sk.sketch_file(b, "k=21,k=31,scaled=100,abund")
        assert [m.md5sum() for m in sa.minhashes()] == [m.md5sum() for m in sb.minhashes()]
        assert all(len(m) > 0 for m in sb.minhashes())
    mid = sk.gunzip_counters()
    assert mid[0] - before[0] == 2 and mid[1] == before[1]
    s2, = sk.sketch_file(two, "k=31,scaled=100")
    after = sk.gunzip_counters()
    assert after[1] - mid[1] == 1                                      # refused by the device, served by the host: both members' k-mers
    both = _write(tmp_path, "both.fa", data + data[:100_000])
    s3, = sk.sketch_file(both, "k=31,scaled=100")
    assert s2.minhash.md5sum() == s3.minhash.md5sum()
    many = sk.sketch_files([packed, fq_packed, two, packed], "k=31,scaled=100", threads=3)
    assert many[0].minhash.md5sum() == many[3].minhash.md5sum() == sk.sketch_file(plain, "k=31,scaled=100")[0].minhash.md5sum()
    assert many[2].minhash.md5sum() == s3.minhash.md5sum()


def test_ingest_suite_on_the_host_inflater():
    "the .gz cases of tests/test_gpu_ingest.py once more with the device inflater switched off (SMG_GUNZIP_DEVICE=0): the fallback stays a tested path"
    env = dict(os.environ, SMG_GUNZIP_DEVICE="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_ingest.py"), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_many_small_files_batched_give_the_per_file_sketches(sk, tmp_path):
    "smgpu_sketch_files inflates AND sketches its gzip members per batch (one tagged sort for all lists): every signature must equal the per-file one"
    rng = np.random.default_rng(41)
    paths = []
    for i in range(37):
        kind = i % 6
        if kind == 0:
            blob = gz(fastq(rng, 300 + 7 * i), 6)
        elif kind == 1:
            blob = fasta(rng, 20_000 + 1000 * i)                       # not compressed: the per-file path
        elif kind == 2:
            blob = gz(fasta(rng, 5_000), 6) + gz(fasta(rng, 3_000), 6)  # two members: the host inflater
        elif kind == 3 and i == 3:
            blob = gz(b"", 6)                                          # an empty file
        else:
            blob = gz(fasta(rng, 30_000 + 4_000 * i) + fasta(rng, 1_000), 1 + i % 9)
        paths.append(_write(tmp_path, f"m{i}.{'fq' if kind == 0 else 'fa'}{'' if kind == 1 else '.gz'}", blob))
    for params in ("k=21,k=31,scaled=100,abund", "k=31,scaled=1", "k=31,num=500", "k=21,k=31,k=51,scaled=1000"):
        many = sk.sketch_files(paths, params, threads=3)
        for p, sig in zip(paths, many):
            one, = sk.sketch_file(p, params)
            assert [m.md5sum() for m in sig.minhashes()] == [m.md5sum() for m in one.minhashes()], (params, p)
            if "abund" in params:
                for a, b in zip(sig.minhashes(), one.minhashes()):
                    assert a.hashes == b.hashes
    one_thread = sk.sketch_files(paths, "k=31,scaled=100", threads=1)      # one batch of all the files
    assert [s.minhash.md5sum() for s in one_thread] == [sk.sketch_file(p, "k=31,scaled=100")[0].minhash.md5sum() for p in paths]
