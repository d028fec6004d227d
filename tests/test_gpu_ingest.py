"""GPU parity of the native FASTA/FASTQ(.gz) ingest path (smgpu_signature_add_file) against the
oracle and the golden genome sketches.  Run with -m gpu."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from conftest import golden, ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sm():
    import torch  # noqa: F401
    import sourmash_amd
    assert sourmash_amd.gpu_available()
    return sourmash_amd


def _write_fasta(path, records, width=70, crlf=False, gz=False):
    eol = "\r\n" if crlf else "\n"
    text = "".join(f">{n}{eol}" + eol.join(s[i:i + width] for i in range(0, len(s), width)) + eol for n, s in records)
    (gzip.open if gz else open)(path, "wb").write(text.encode())


def _records(rng, n=7):
    out = []
    for i in range(n):
        L = int(rng.integers(10, 40_000))
        s = bytearray(rng.choice(np.frombuffer(b"ACGTacgt", dtype=np.uint8), size=L).tobytes())
        for j in range(5, L, 501):
            s[j] = ord("N")
        out.append((f"rec{i} some description", s.decode()))
    out.append(("short", "ACG"))
    out.append(("empty", ""))
    return out


def _oracle_sig(records, k, scaled=0, num=0, abund=False):
    mh = oracle.OracleMinHash(num, k, scaled=scaled, track_abundance=abund)
    for _, s in records:
        mh.add_sequence(s, force=True)
    return mh


def test_golden_genomes_through_native_ingest(sm):
    from sourmash_amd.sketch import sketch_file
    fa = golden("ecoli", "GCF_000005845.2_ASM584v2_genomic.fna.gz")
    want = {s["ksize"]: s["md5sum"] for s in oracle.read_sig_json(fa + ".sig")}
    sig, = sketch_file(fa, "k=21,k=31,k=51,scaled=1000")
    assert {mh.ksize: mh.md5sum() for mh in sig.minhashes()} == want
    assert sig.filename == fa
    fa = golden("num", "genome-s10.fa.gz")                       # multi-record, num sketches
    for w in [s for s in oracle.read_sig_json(fa + ".sig") if s["molecule"].lower() == "dna"]:
        sig, = sketch_file(fa, f"k={w['ksize']},num={w['num']}")
        assert sig.md5sum() == w["md5sum"]


@pytest.mark.parametrize("crlf,gz", [(False, False), (True, False), (False, True)])
def test_fasta_variants_vs_oracle(sm, tmp_path, crlf, gz):
    from sourmash_amd.sketch import sketch_file
    recs = _records(np.random.default_rng(3))
    path = str(tmp_path / ("x.fa.gz" if gz else "x.fa"))
    _write_fasta(path, recs, crlf=crlf, gz=gz)
    sig, = sketch_file(path, "k=21,k=31,scaled=50,abund")
    for mh in sig.minhashes():
        want = _oracle_sig(recs, mh.ksize, scaled=50, abund=True)
        assert np.array_equal(mh._mins_array(), want.mins), mh.ksize
        assert list(mh.hashes.values()) == want.abunds.tolist()


def test_fastq_and_chunk_boundaries(sm, tmp_path):
    """Many tiny chunks (SMG_INGEST_CHUNK) so that records and k-mers straddle chunk boundaries; run in a
    subprocess because the chunk size is read when the library first ingests."""
    recs = _records(np.random.default_rng(5), n=5)
    fq = str(tmp_path / "r.fastq")
    with open(fq, "w") as fh:
        for n, s in recs:
            fh.write(f"@{n}\n{s}\n+\n{'I' * len(s)}\n")
    fa = str(tmp_path / "r.fa")
    _write_fasta(fa, recs, width=61)
    code = f"""
import sys, numpy as np
sys.path.insert(0, {ROOT!r})
import torch
from sourmash_amd.sketch import sketch_file
for path in ({fq!r}, {fa!r}):
    sig, = sketch_file(path, "k=21,k=51,scaled=20,abund")
    for mh in sig.minhashes():
        np.save(path + f".k{{mh.ksize}}.npy", np.array([list(mh.hashes.keys()), list(mh.hashes.values())], dtype=np.uint64))
    sig, = sketch_file(path, "k=31,num=300")
    np.save(path + ".num.npy", sig.minhash._mins_array())
"""
    env = dict(os.environ, SMG_INGEST_CHUNK="1000")
    subprocess.check_call([sys.executable, "-c", code], env=env)
    for path in (fq, fa):
        for k in (21, 51):
            got = np.load(path + f".k{k}.npy")
            want = _oracle_sig(recs, k, scaled=20, abund=True)
            assert np.array_equal(got[0], want.mins), (path, k)
            assert np.array_equal(got[1], want.abunds), (path, k)          # no k-mer hashed twice at a boundary
        assert np.array_equal(np.load(path + ".num.npy"), _oracle_sig(recs, 31, num=300).mins)


def test_record_parser_over_many_blocks_and_the_fused_ksize_pass(sm, tmp_path):
    """Round 6: the record structure comes from per-block summaries (csrc/fastx.hip: a block's effect on the line state, its kept
    bytes as a function of the state that enters it), and k = 21 / 31 / 51 are hashed in one pass (csrc/sketch_multi.hip).  A FASTQ
    of several hundred 8 KiB blocks whose quality lines start with '@', '>' and '+' (line counting, not first characters, decides),
    a CRLF FASTA with '>' inside sequence lines and empty lines, whole and in chunks that are not multiples of the block -- with
    abundances, so a k-mer hashed twice at a chunk border (the shorter ksizes start behind the longest one's halo) would show."""
    rng = np.random.default_rng(77)
    recs = []
    for i in range(1500):
        L = int(rng.integers(30, 2500))
        s = bytearray(rng.choice(np.frombuffer(b"ACGTacgt", dtype=np.uint8), size=L).tobytes())
        if i % 7 == 0:
            s[L // 2] = ord("N")
        recs.append((f"read{i}/1 len={L}", s.decode()))
    fq = str(tmp_path / "big.fastq")
    with open(fq, "w") as fh:
        for i, (n, s) in enumerate(recs):
            q = ("@>+I"[i % 4] + "I" * (len(s) - 1)) if s else ""
            fh.write(f"@{n}\n{s}\n+{n if i % 3 == 0 else ''}\n{q}\n")
    fa = str(tmp_path / "big.fa")
    with open(fa, "wb") as fh:
        for i, (n, s) in enumerate(recs):
            body = "\r\n".join(s[j:j + 80] for j in range(0, len(s), 80))
            fh.write((f">{n} a>b\r\n{body}\r\n" + ("\r\n" if i % 11 == 0 else "")).encode())
    assert os.path.getsize(fq) > 300 * 8192
    code = f"""
import sys, numpy as np
sys.path.insert(0, {ROOT!r})
import torch
from sourmash_amd.sketch import sketch_file
for path in ({fq!r}, {fa!r}):
    sig, = sketch_file(path, "k=21,k=31,k=51,scaled=50,abund")
    for mh in sig.minhashes():
        np.save(path + f".k{{mh.ksize}}.npy", np.array([list(mh.hashes.keys()), list(mh.hashes.values())], dtype=np.uint64))
"""
    wants = {k: _oracle_sig(recs, k, scaled=50, abund=True) for k in (21, 31, 51)}
    for chunk in (None, "100003", "8192", "70001"):
        env = dict(os.environ) if chunk is None else dict(os.environ, SMG_INGEST_CHUNK=chunk)
        subprocess.check_call([sys.executable, "-c", code], env=env)
        for path in (fq, fa):
            for k in (21, 31, 51):
                got = np.load(path + f".k{k}.npy")
                assert np.array_equal(got[0], wants[k].mins), (chunk, path, k)
                assert np.array_equal(got[1], wants[k].abunds), (chunk, path, k)


def test_long_kmers_across_chunk_boundaries(sm, tmp_path):
    """k = 300 and 1,500 through the streaming ingest with chunks of 1,000 bytes: the halo between chunks is k - 1 bytes -- longer
    than a chunk for the second (rounds 1-4 kept a fixed 256-byte halo and refused k > 256)"""
    rng = np.random.default_rng(6)
    recs = []
    for i, L in enumerate((9_000, 1_499, 1_500, 23_456, 2_000)):
        s = bytearray(rng.choice(np.frombuffer(b"ACGTacgt", dtype=np.uint8), size=L).tobytes())
        for j in range(4_000, L, 7_001):
            s[j] = ord("N")
        recs.append((f"rec{i}", s.decode()))
    fa = str(tmp_path / "long.fa")
    _write_fasta(fa, recs, width=80)
    code = f"""
import sys, numpy as np
sys.path.insert(0, {ROOT!r})
import torch
from sourmash_amd.sketch import sketch_file
sig, = sketch_file({fa!r}, "k=21,k=300,k=1500,scaled=5,abund")
for mh in sig.minhashes():
    np.save({fa!r} + f".k{{mh.ksize}}.npy", np.array([list(mh.hashes.keys()), list(mh.hashes.values())], dtype=np.uint64))
"""
    subprocess.check_call([sys.executable, "-c", code], env=dict(os.environ, SMG_INGEST_CHUNK="1000"))
    for k in (21, 300, 1500):
        got = np.load(fa + f".k{k}.npy")
        want = _oracle_sig(recs, k, scaled=5, abund=True)
        assert len(want.mins) > 0 and np.array_equal(got[0], want.mins), k
        assert np.array_equal(got[1], want.abunds), k                      # no k-mer hashed twice at a boundary


def test_missing_file_raises(sm):
    from sourmash_amd.sketch import sketch_file
    with pytest.raises(sm.exceptions.SourmashError):
        sketch_file("/nonexistent/file.fa")


def test_many_files_in_parallel(sm, tmp_path):
    "smgpu_sketch_files: independent pipelines, one signature per file in input order, same sketches as file by file"
    from sourmash_amd.sketch import sketch_file, sketch_files
    recs = _records(np.random.default_rng(9), n=4)
    fa, fq = str(tmp_path / "a.fa"), str(tmp_path / "b.fastq")
    _write_fasta(fa, recs, width=60)
    with open(fq, "w") as fh:
        for n, s in recs[::-1]:
            fh.write(f"@{n}\n{s}\n+\n{'I' * len(s)}\n")
    paths = [golden("ecoli", "GCF_000005845.2_ASM584v2_genomic.fna.gz"), fa, golden("num", "genome-s10.fa.gz"), fq,
             golden("scaled100", "GCF_000006945.1_ASM694v1_genomic.fna.gz")] * 3
    params = "k=21,k=31,k=51,scaled=1000,abund"
    one_by_one = {p: sketch_file(p, params)[0] for p in set(paths)}
    for threads in (1, 4, 0):
        sigs = sketch_files(paths, params, threads=threads)
        assert [s.filename for s in sigs] == paths
        for p, sig in zip(paths, sigs):
            want = {mh.ksize: mh for mh in one_by_one[p].minhashes()}
            got = {mh.ksize: mh for mh in sig.minhashes()}
            assert sorted(got) == [21, 31, 51]
            for k in got:
                assert got[k] == want[k], (p, k, threads)
    flat = sketch_files(paths[:1], "k=31,scaled=1000")[0]
    assert flat.minhash.md5sum() == "0a8632c67e6d88f737ddb510bef90337"          # the reference's E. coli k=31 sketch
    assert sketch_files([]) == []
    with pytest.raises(sm.exceptions.SourmashError):
        sketch_files([fa, "/nonexistent/file.fa"], params)


def test_repeated_reads_keep_more_hashes_than_the_statistical_estimate(sm, tmp_path):
    """The number of kept hashes a chunk produces depends on multiplicity, not on distinct k-mers: a deep amplicon whose one
    k-mer falls under max_hash keeps a hash per READ.  The ingest path sizes its output for the positions of the chunk, so
    this is ordinary input (round 1 sized it from slen / scaled and raised 'sketch output overflow': ADVICE.md)."""
    from sourmash_amd.sketch import sketch_file
    rng = np.random.default_rng(31)
    read = None
    for _ in range(200_000):                                      # a 31-mer that scaled=1000 keeps
        cand = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=31)).decode()
        mh = oracle.OracleMinHash(0, 31, scaled=1000)
        mh.add_sequence(cand.encode(), force=True)
        if len(mh.mins):
            read = cand
            break
    assert read is not None
    copies = 60_000                                               # 1.9 MB of sequence: the estimate was ~7,000 kept hashes
    fq = str(tmp_path / "amplicon.fastq")
    with open(fq, "w") as fh:
        fh.write(f"@r\n{read}\n+\n{'I' * 31}\n" * copies)
    sig, = sketch_file(fq, "k=31,scaled=1000,abund")
    mh = sig.minhash
    assert len(mh) == 1 and list(mh.hashes.values()) == [copies]
    sig, = sketch_file(fq, "k=31,scaled=1000")
    assert len(sig.minhash) == 1


def test_one_large_gzip_member_inflated_on_many_threads(sm, tmp_path):
    """A single big .fna.gz: the reader cuts the deflate stream into spans, finds block starts by search, inflates the spans on
    all host threads and fills window references in afterwards (csrc/pargz.hpp).  Same sketch as from the plain file and as the
    oracle's; SMG_GUNZIP_SEQUENTIAL=1 (one zlib stream, what the reference does: command_sketch.py:697) agrees too."""
    import ctypes as C
    import zlib
    from sourmash_amd._lowlevel import lib
    from sourmash_amd.sketch import sketch_file
    n = 48_000_000
    seq = oracle.synth_dna(0, n, seed=77, record_len=3_999_999)            # 12 records + separators
    recs = bytes(seq).split(b"\n")
    plain = tmp_path / "big.fa"
    with open(plain, "wb") as fh:
        for i, r in enumerate(recs):
            if not r:
                continue
            a = np.frombuffer(r, dtype=np.uint8)
            full = (len(a) // 60) * 60
            fh.write(b">big_%d\n" % i)
            fh.write(np.concatenate([a[:full].reshape(-1, 60), np.full((full // 60, 1), 10, dtype=np.uint8)], axis=1).tobytes())
            if len(a) > full:
                fh.write(a[full:].tobytes() + b"\n")
    gz = tmp_path / "big.fa.gz"
    co = zlib.compressobj(1, zlib.DEFLATED, 31)
    with open(plain, "rb") as fi, open(gz, "wb") as fo:
        while True:
            block = fi.read(8 << 20)
            if not block:
                break
            fo.write(co.compress(block))
        fo.write(co.flush())
    par = C.c_bool(False)
    total = lib.smgpu_gunzip_file(str(gz).encode(), 0, 1 << 20, None, 0, None, C.byref(par))
    assert total == plain.stat().st_size and par.value, "the file should be large enough for the many-thread form"
    want = oracle.sketch_dna_bulk(np.frombuffer(seq, dtype=np.uint8), 31, scaled=1000, nthreads=8)
    sig_plain, = sketch_file(str(plain), "k=31,scaled=1000")
    sig_gz, = sketch_file(str(gz), "k=31,scaled=1000")
    assert np.array_equal(sig_plain.minhash._mins_array(), want)
    assert np.array_equal(sig_gz.minhash._mins_array(), want)
    assert sig_gz.md5sum() == sig_plain.md5sum()
