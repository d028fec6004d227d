"""GPU parity: the HIP sketch path, called through the C-ABI, against the oracle,
the reference's known-answer tests and its golden fixtures.  Run with -m gpu."""
import numpy as np
import pytest

import oracle
from conftest import golden

pytestmark = pytest.mark.gpu

MAX_HASH_1000 = 18446744073709552


@pytest.fixture(scope="module")
def sm():
    import torch  # noqa: F401
    import sourmash_amd
    assert sourmash_amd.gpu_available(), "these tests need a real GPU"
    return sourmash_amd


def _rand_dna(rng, n, alphabet=b"ACGT"):
    return bytes(rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=n))


# ---- the reference's own KATs, through MinHash ---------------------------------------------------
def test_kats(sm):
    assert sm.hash_murmur("ACG") == 1731421407650554201           # tests/test_minhash.py:1239-1264
    mh = sm.MinHash(1, 4)
    mh.add_sequence("ATGC")
    assert list(mh.hashes) == [12415348535738636339]               # tests/test_minhash.py:98-112
    mh = sm.MinHash(20, 5, track_abundance=True)
    mh.add_sequence("AAAAA")
    assert dict(mh.hashes) == {2110480117637990133: 1}             # tests/test_minhash.py:1267-1276
    mh.add_sequence("AAAAA")
    assert dict(mh.hashes) == {2110480117637990133: 2}
    a, b = sm.MinHash(20, 10), sm.MinHash(20, 10)                   # src/core/tests/minhash.rs:29-54
    a.add_sequence("TGCCGCCCAGCA"); b.add_sequence("TGCCGCCCAGCA")
    a.add_sequence("GTCCGCCCAGTGA"); b.add_sequence("GTCCGCCCAGTGG")
    a.merge(b)
    assert list(a.hashes) == [2996412506971915891, 4448613756639084635, 8373222269469409550, 9390240264282449587,
                              11085758717695534616, 11668188995231815419, 11760449009842383350,
                              14682565545778736889]


def test_invalid_dna_semantics(sm):
    a = sm.MinHash(20, 3)
    a.add_sequence("AAANNCCCTN", True)                              # src/core/tests/minhash.rs:56-66
    assert len(a) == 3
    b = sm.MinHash(20, 3)
    b.add_sequence("NAAA", True)
    assert len(b) == 1
    mh = sm.MinHash(1, 4)
    with pytest.raises(ValueError) as e:                             # tests/test_minhash.py:711-719
        mh.add_sequence("ATGR")
    assert "invalid DNA character in input k-mer: ATGR" in str(e.value)
    # streaming semantics: the k-mers before the first bad one are already in (signature.rs:48-54)
    mh = sm.MinHash(0, 4, scaled=1)
    om = oracle.OracleMinHash(0, 4, scaled=1)
    with pytest.raises(ValueError) as e:
        mh.add_sequence("ATGCATGRACGTN")
    with pytest.raises(ValueError) as eo:
        om.add_sequence("ATGCATGRACGTN")
    assert str(e.value) == str(eo.value)
    assert np.array_equal(mh._mins_array(), om.mins) and len(mh) > 0
    # shorter than k: silently nothing, even with junk (tests/test_minhash.py:1232-1236)
    mh = sm.MinHash(0, 31, scaled=1)
    mh.add_sequence("ACGTN")
    assert len(mh) == 0
    # lowercase == uppercase (signature.rs:214)
    u, l = sm.MinHash(0, 21, scaled=1), sm.MinHash(0, 21, scaled=1)
    s = _rand_dna(np.random.default_rng(3), 500).decode()
    u.add_sequence(s); l.add_sequence(s.lower())
    assert u == l and len(u) > 400


def test_seq_to_hashes(sm):
    rng = np.random.default_rng(5)
    seq = bytearray(_rand_dna(rng, 3000, b"ACGTacgt"))
    for i in range(1, len(seq), 89):
        seq[i] = ord("N")
    seq = bytes(seq)
    for k in (31, 21, 5, 25):                                       # 25: generic-k kernel
        mh = sm.MinHash(0, k, scaled=1)
        got = mh.seq_to_hashes(seq, force=True, bad_kmers_as_zeroes=True)
        want = oracle.seq_to_hashes(seq, k, force=True, bad_kmers_as_zeroes=True)
        assert got == want, k
        assert mh.seq_to_hashes(seq, force=True) == [h for h in want if h]
        with pytest.raises(ValueError) as e:
            mh.seq_to_hashes(seq)
        with pytest.raises(ValueError) as eo:
            oracle.seq_to_hashes(seq, k)
        assert str(e.value) == str(eo.value)
        assert len(mh) == 0                                          # seq_to_hashes never adds
    with pytest.raises(ValueError):
        sm.MinHash(0, 5, scaled=1).seq_to_hashes("ACGTN", bad_kmers_as_zeroes=True)


@pytest.mark.parametrize("k", [31, 21, 51, 4, 16, 25, 33, 64])
def test_random_vs_oracle(sm, k):
    rng = np.random.default_rng(100 + k)
    for n, scaled in ((0, 1), (k - 1, 1), (k, 1), (5000, 1), (300_000, 50), (1_000_003, 1000)):
        s = bytearray(_rand_dna(rng, n, b"ACGTacgt"))
        for i in range(7, n, 997):
            s[i] = ord("N")
        s = bytes(s)
        mh = sm.MinHash(0, k, scaled=scaled)
        mh.add_sequence_buffer(s)
        want = oracle.sketch_dna_bulk(s, k, scaled=scaled, nthreads=4) if n >= k else np.zeros(0, np.uint64)
        assert np.array_equal(mh._mins_array(), want), (k, n, scaled)


def test_every_ksize_on_the_gpu(sm):
    """k = 1 .. 88 dispatch to instantiations of the register-window kernel -- the appending form (sketch.hip, sketch_long.hip) and
    the per-position form (sketch_dense.hip: kmerminhash_seq_to_hashes) alike -- longer k-mers to the run-time-k kernel of
    sketch_words.hip (any tail length, past the old limit of 256): all vs the oracle"""
    rng = np.random.default_rng(77)
    s = bytearray(_rand_dna(rng, 40_000, b"ACGTacgt"))
    for i in range(11, len(s), 1013):
        s[i] = ord("N")
    s = bytes(s)
    for k in list(range(1, 129)) + [129, 130, 143, 144, 145, 160, 200, 255, 256, 257, 300]:
        mh = sm.MinHash(0, k, scaled=4)
        mh.add_sequence_buffer(s)
        assert np.array_equal(mh._mins_array(), oracle.sketch_dna_bulk(s, k, scaled=4, nthreads=4)), k
        ordered = mh.seq_to_hashes(s[:4200].decode(), force=True, bad_kmers_as_zeroes=True)     # per-position output (two tiles of the kernel)
        assert ordered == [h or 0 for h in oracle.seq_to_hashes(s[:4200], k, force=True, bad_kmers_as_zeroes=True)], k


def test_long_kmers_palindromes_and_seams(sm):
    """k > 128 (sketch_words.hip): k-mers equal to their reverse complement for many 16-byte blocks, bad bytes next to the seams of
    the 4,096-position stretches, k of a thousand and more -- vs the oracle"""
    half = b"ACGGTCATTGCA" * 40
    pal = half + bytes(half[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA")))
    rng = np.random.default_rng(78)
    s = bytearray(_rand_dna(rng, 3000) + pal + b"A" * 300 + b"T" * 300 + _rand_dna(rng, 12_000, b"ACGTacgt") + b"AT" * 200)
    for i in (4095, 4096, 4097, 8191 + 150, 12_288):
        s[i] = ord("N")
    s = bytes(s)
    for k in (129, 161, 240, 480, 1000, 5000):
        mh = sm.MinHash(0, k, scaled=3)
        mh.add_sequence_buffer(s[3:])
        assert np.array_equal(mh._mins_array(), oracle.sketch_dna_bulk(s[3:], k, scaled=3, nthreads=4)), k
        ordered = mh.seq_to_hashes(s[:9000].decode(), force=True, bad_kmers_as_zeroes=True)
        assert ordered == [h or 0 for h in oracle.seq_to_hashes(s[:9000], k, force=True, bad_kmers_as_zeroes=True)], k


def test_long_kmers_random_cases(sm):
    "forty random (k, length, alphabet, scaled) cases above the unrolled kernel's range: kept hashes and per-position hashes vs the oracle"
    rng = np.random.default_rng(81)
    for case in range(40):
        k = int(rng.integers(89, 420))
        n = int(rng.integers(k, 12_000))
        alphabet = (b"ACGT", b"ACGTacgt", b"ACGTN", b"ACGTacgtRY\n")[case % 4]
        weights = None if case % 4 < 2 else [0.97 / 4] * 4 + [0.03 / (len(alphabet) - 4)] * (len(alphabet) - 4) if len(alphabet) == 5 else None
        s = bytes(rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=n, p=weights))
        if case % 4 == 3:                                        # rare bad bytes instead of the uniform draw
            b = bytearray(_rand_dna(rng, n, b"ACGTacgt"))
            for i in rng.integers(0, n, size=max(1, n // 900)):
                b[int(i)] = alphabet[int(rng.integers(8, len(alphabet)))]
            s = bytes(b)
        scaled = int(rng.choice([1, 2, 7, 50]))
        mh = sm.MinHash(0, k, scaled=scaled)
        mh.add_sequence_buffer(s)
        assert np.array_equal(mh._mins_array(), oracle.sketch_dna_bulk(s, k, scaled=scaled, nthreads=2)), (case, k, n, scaled)
        ordered = mh.seq_to_hashes(s.decode("latin-1"), force=True, bad_kmers_as_zeroes=True)
        assert ordered == [h or 0 for h in oracle.seq_to_hashes(s, k, force=True, bad_kmers_as_zeroes=True)], (case, k, n)


def test_long_kmers_every_hash_kept(sm):
    "scaled = 1 and bottom-k sketches at k > 88: a stretch keeps more hashes than the kernel's LDS buffer holds (the spill path)"
    rng = np.random.default_rng(80)
    s = _rand_dna(rng, 30_000)
    for k in (89, 130, 300):
        mh = sm.MinHash(0, k, scaled=1)
        mh.add_sequence_buffer(s)
        assert np.array_equal(mh._mins_array(), oracle.sketch_dna_bulk(s, k, scaled=1, nthreads=4)), k
        mh = sm.MinHash(500, k)
        om = oracle.OracleMinHash(500, k)
        mh.add_sequence(s.decode()); om.add_sequence(s)
        assert np.array_equal(mh._mins_array(), om.mins), k


def test_the_longest_kmer_the_lds_holds(sm):
    "k = 60,000 (sketch_words.hip: 2.25 x (4,096 + k) bytes of LDS + the kept-hash buffer = 160 KB) vs the oracle; one more is refused"
    rng = np.random.default_rng(79)
    s = _rand_dna(rng, 70_000)
    mh = sm.MinHash(0, 60_000, scaled=2)
    mh.add_sequence_buffer(s)
    want = oracle.sketch_dna_bulk(s, 60_000, scaled=2, nthreads=8)
    assert len(want) > 3000 and np.array_equal(mh._mins_array(), want)
    mh = sm.MinHash(0, 60_001, scaled=2)
    with pytest.raises(Exception, match="longest DNA k-mer this device"):       # said in words at the entry point, per-record call as well
        mh.add_sequence_buffer(s)
        len(mh)                                                  # (buffers may be hashed when the sketch is next looked at)
    with pytest.raises(Exception, match="longest DNA k-mer this device"):
        sm.MinHash(0, 60_001, scaled=2).add_sequence(s.decode() if isinstance(s, (bytes, bytearray)) else bytes(s).decode(), True)


def test_abundance_and_num(sm):
    rng = np.random.default_rng(9)
    s = _rand_dna(rng, 2000) * 3 + _rand_dna(rng, 5000)
    mh = sm.MinHash(0, 21, scaled=20, track_abundance=True)
    om = oracle.OracleMinHash(0, 21, scaled=20, track_abundance=True)
    mh.add_sequence(s.decode()); om.add_sequence(s)
    assert np.array_equal(mh._mins_array(), om.mins)
    assert list(mh.hashes.values()) == om.abunds.tolist() and max(mh.hashes.values()) >= 3
    for num in (1, 50, 500):
        mh = sm.MinHash(num, 21)
        om = oracle.OracleMinHash(num, 21)
        for piece in (s[:3000], s[3000:]):
            mh.add_sequence(piece.decode()); om.add_sequence(piece)
        assert np.array_equal(mh._mins_array(), om.mins) and len(mh) == num


# ---- golden genomes -----------------------------------------------------------------------------------
def test_ecoli_golden_all_ksizes(sm):
    from sourmash_amd.sketch import sketch_file
    fa = golden("ecoli", "GCF_000005845.2_ASM584v2_genomic.fna.gz")
    want = {s["ksize"]: s for s in oracle.read_sig_json(fa + ".sig")}
    sig, = sketch_file(fa, "k=21,k=31,k=51,scaled=1000")
    got = {mh.ksize: mh for mh in sig.minhashes()}
    assert sorted(got) == [21, 31, 51]
    for k, mh in got.items():
        assert mh._max_hash == MAX_HASH_1000
        assert np.array_equal(mh._mins_array(), np.sort(want[k]["mins"])), k
        assert mh.md5sum() == want[k]["md5sum"]
    assert len(got[31]) == 4476 and got[31].md5sum() == "0a8632c67e6d88f737ddb510bef90337"
    # JSON round trip equals the reference's file content sketch for sketch
    back = {s.minhash.ksize: s for s in sm.load_signatures_from_json(sm.save_signatures_to_json([sig]))}
    assert all(back[k].minhash == got[k] for k in got)


def test_scaled100_and_num_golden(sm):
    from sourmash_amd.sketch import sketch_file
    fa = golden("scaled100", "GCF_000006945.1_ASM694v1_genomic.fna.gz")
    want, = oracle.read_sig_json(golden("scaled100", "GCF_000006945.1_ASM694v1_genomic.fna.gz.sig.gz"))
    sig, = sketch_file(fa, "k=21,scaled=100")
    assert len(sig.minhash) == 48504 and sig.md5sum() == want["md5sum"]
    fa = golden("num", "genome-s10.fa.gz")
    for want in [s for s in oracle.read_sig_json(fa + ".sig") if s["molecule"].lower() == "dna"]:
        sig, = sketch_file(fa, f"k={want['ksize']},num={want['num']}")
        assert sig.md5sum() == want["md5sum"], want["ksize"]
    # check_sequence (force=False) path on a clean genome gives the same sketch
    sig2, = sketch_file(fa, "k=21,num=500", check_sequence=True)
    assert sig2.md5sum() == [s for s in oracle.read_sig_json(fa + ".sig") if s["ksize"] == 21 and s["molecule"].lower() == "dna"][0]["md5sum"]


# ---- device-resident path -------------------------------------------------------------------------------
def test_device_synth_and_sketch(sm):
    import torch
    from sourmash_amd import device as smd
    n = 20_000_037
    seq = smd.synth_dna(n, seed=42, record_len=99_999, start=12345)
    host = oracle.synth_dna(12345, n, seed=42, record_len=99_999)
    assert np.array_equal(seq.cpu().numpy(), host)
    for k, scaled in ((31, 1000), (21, 100), (51, 1000)):
        sk = smd.DeviceSketcher(ksize=k, scaled=scaled)
        got = sk.sketch(seq).cpu().numpy().view(np.uint64)
        want = oracle.sketch_dna_bulk(host, k, scaled=scaled, nthreads=8)
        assert np.array_equal(got, want), (k, scaled)
    # pathological density: poly-A keeps one hash n times -> capacity retry path, one unique hash
    polya = torch.full((3_000_000,), ord("A"), dtype=torch.uint8, device="cuda")
    h = int(oracle.sketch_dna_bulk(b"A" * 31, 31, scaled=1)[0])
    sk = smd.DeviceSketcher(ksize=31, scaled=max(2, int(2**64 / (h + 1)) - 1))
    assert h <= sk.max_hash
    got = sk.sketch(polya).cpu().numpy().view(np.uint64)
    assert got.tolist() == [h]


def test_full_size_properties(sm):
    """BASELINE config C2 scale (per-GPU 10 GB is bench.py's job; here 2e9 bases): properties that
    do not need the oracle at full size."""
    import torch
    from sourmash_amd import device as smd
    rec = 10_000_000
    n = 200 * (rec + 1)
    seq = smd.synth_dna(n, seed=42, record_len=rec)
    sk = smd.DeviceSketcher(ksize=31, scaled=1000)
    h = sk.sketch(seq)
    hs = h.cpu().numpy().view(np.uint64)
    assert np.all(hs[1:] > hs[:-1]) and hs[0] > 0 and hs[-1] <= MAX_HASH_1000      # sorted, unique, kept range
    n_kmers = 200 * (rec - 30)
    assert abs(len(hs) - n_kmers / 1000) < 6 * (n_kmers / 1000) ** 0.5             # binomial count
    # additivity: sketch(A + B) == union(sketch(A), sketch(B)) when cut at a record boundary
    cut = 77 * (rec + 1)
    a = sk.sketch(seq[:cut]).cpu().numpy().view(np.uint64).copy()
    b = sk.sketch(seq[cut:]).cpu().numpy().view(np.uint64).copy()
    assert np.array_equal(np.union1d(a, b), hs)
    # idempotence + oracle on a slice of the same buffer
    assert np.array_equal(sk.sketch(seq).cpu().numpy().view(np.uint64), hs)
    sl = seq[cut:cut + 30_000_000]
    assert np.array_equal(sk.sketch(sl).cpu().numpy().view(np.uint64),
                          oracle.sketch_dna_bulk(sl.cpu().numpy(), 31, scaled=1000, nthreads=8))
