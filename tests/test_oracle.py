"""Pin the CPU oracle (oracle/oracle.c) to the reference's own known-answer
tests and golden fixtures.  CPU only."""
import glob
import os

import numpy as np
import pytest

import oracle
from oracle import OracleMinHash
from conftest import golden

MAX_HASH_1000 = 18446744073709552


# ---- murmur / parameter KATs -------------------------------------------------
def test_hash_murmur_kat():
    # tests/test_minhash.py:1239-1264
    assert oracle.hash_murmur("ACG") == 1731421407650554201
    assert oracle.hash_murmur("ACG", 42) == 1731421407650554201
    assert oracle.hash_murmur("ACG", 43) != 1731421407650554201


def test_max_hash_for_scaled():
    # src/core/tests/minhash.rs:177-180 ; tests/test-data/47.fa.sig max_hash
    assert oracle.max_hash_for_scaled(100) == 184467440737095520
    assert oracle.max_hash_for_scaled(1000) == MAX_HASH_1000
    assert oracle.max_hash_for_scaled(0) == 0
    assert oracle.max_hash_for_scaled(1) == 2**64 - 1
    assert oracle.scaled_for_max_hash(MAX_HASH_1000) == 1000
    assert oracle.scaled_for_max_hash(184467440737095520) == 100


def test_single_kmer_kats():
    # tests/test_minhash.py:98-112,1466-1474
    mh = OracleMinHash(1, 4)
    mh.add_sequence("ATGC")
    assert mh.mins.tolist() == [12415348535738636339]
    # tests/test_minhash.py:1267-1276
    mh = OracleMinHash(20, 5, track_abundance=True)
    mh.add_sequence("AAAAA")
    assert mh.mins.tolist() == [2110480117637990133] and mh.abunds.tolist() == [1]
    mh.add_sequence("AAAAA")
    assert mh.abunds.tolist() == [2]


def test_merge_kat():
    # src/core/tests/minhash.rs:29-54
    a = OracleMinHash(20, 10)
    b = OracleMinHash(20, 10)
    a.add_sequence("TGCCGCCCAGCA")
    b.add_sequence("TGCCGCCCAGCA")
    a.add_sequence("GTCCGCCCAGTGA")
    b.add_sequence("GTCCGCCCAGTGG")
    a.merge(b)
    assert a.mins.tolist() == [
        2996412506971915891, 4448613756639084635, 8373222269469409550, 9390240264282449587,
        11085758717695534616, 11668188995231815419, 11760449009842383350, 14682565545778736889]


def test_invalid_dna():
    # src/core/tests/minhash.rs:56-66
    a = OracleMinHash(20, 3)
    a.add_sequence("AAANNCCCTN", force=True)
    assert len(a) == 3
    b = OracleMinHash(20, 3)
    b.add_sequence("NAAA", force=True)
    assert len(b) == 1
    # src/core/tests/minhash.rs:19-27 ; tests/test_minhash.py:711-719
    mh = OracleMinHash(1, 4)
    with pytest.raises(ValueError) as e:
        mh.add_sequence("ATGR")
    assert "invalid DNA character in input k-mer: ATGR" in str(e.value)
    # streaming semantics: hashes of earlier valid k-mers are already in (signature.rs:48-54)
    mh = OracleMinHash(0, 4, scaled=1)
    with pytest.raises(ValueError):
        mh.add_sequence("ATGCATGR")
    assert len(mh) > 0


def test_seq_to_hashes_semantics():
    # tests/test_minhash.py:206-219,265-282
    seq = "ATGCAGTGCATGNACGTAGCT"
    k = 5
    hs = oracle.seq_to_hashes(seq, k, force=True, bad_kmers_as_zeroes=True)
    assert len(hs) == len(seq) - k + 1
    for i, h in enumerate(hs):
        kmer = seq[i:i + k]
        if "N" in kmer:
            assert h == 0
        else:
            mh = OracleMinHash(0, k, scaled=1)
            mh.add_sequence(kmer)
            assert mh.mins.tolist() == [h]
    assert oracle.seq_to_hashes(seq, k, force=True) == [h for h in hs if h]
    assert oracle.seq_to_hashes("ATG", 5) == []
    # lowercase sketches identically (signature.rs:214)
    assert oracle.seq_to_hashes(seq.lower(), k, force=True) == oracle.seq_to_hashes(seq, k, force=True)


def test_keep_rule_inclusive():
    # tests/test_minhash.py:475-490: max_hash == 35 keeps 10,20,30 drops 40; inclusive bound
    scaled = oracle.scaled_for_max_hash(35)
    mh = OracleMinHash(0, 4, scaled=scaled)
    mx = mh.max_hash
    for h in (10, 20, 30, mx, mx + 1, mx + 5):
        mh.add_hash(h)
    assert mh.mins.tolist() == sorted({10, 20, 30, mx})


def test_md5_kat():
    assert oracle.md5_hex(b"") == "d41d8cd98f00b204e9800998ecf8427e"
    assert oracle.md5_hex(b"abc") == "900150983cd24fb0d6963f7d28e17f72"
    assert oracle.md5_hex(b"a" * 1000) == "cabe45dcc9ae5b66ba86600cca6b8ba8"


# ---- whole-genome golden sketches -------------------------------------------
def _sketch_fasta(path, ksize, scaled=0, num=0, nthreads=4):
    mh_all = OracleMinHash(num, ksize, scaled=scaled)
    for _, seq in oracle.read_fasta(path):
        if num:
            mh_all.add_sequence(seq, force=True)
        else:
            hs = oracle.sketch_dna_bulk(seq.encode(), ksize, scaled=scaled, nthreads=nthreads)
            mh_all.add_many(hs)
    return mh_all


@pytest.mark.parametrize("ksize", [21, 31, 51])
def test_ecoli_golden(ksize):
    fa = golden("ecoli", "GCF_000005845.2_ASM584v2_genomic.fna.gz")
    sigs = {s["ksize"]: s for s in oracle.read_sig_json(fa + ".sig")}
    want = sigs[ksize]
    mh = _sketch_fasta(fa, ksize, scaled=1000)
    assert want["max_hash"] == mh.max_hash == MAX_HASH_1000
    assert np.array_equal(mh.mins, np.sort(want["mins"]))
    assert mh.md5sum() == want["md5sum"]
    if ksize == 31:
        assert len(mh) == 4476 and mh.md5sum() == "0a8632c67e6d88f737ddb510bef90337"


def test_ecoli_streaming_equals_bulk():
    """the slow per-record reference walk and the bulk path agree (k=31)."""
    fa = golden("ecoli", "GCF_000005845.2_ASM584v2_genomic.fna.gz")
    (_, seq), = list(oracle.read_fasta(fa))
    part = seq[:300000]
    mh = OracleMinHash(0, 31, scaled=1000)
    mh.add_sequence(part, force=True)
    assert np.array_equal(mh.mins, oracle.sketch_dna_bulk(part.encode(), 31, scaled=1000, nthreads=3))


def test_scaled100_golden():
    fa = golden("scaled100", "GCF_000006945.1_ASM694v1_genomic.fna.gz")
    want, = oracle.read_sig_json(golden("scaled100", "GCF_000006945.1_ASM694v1_genomic.fna.gz.sig.gz"))
    assert want["ksize"] == 21
    mh = _sketch_fasta(fa, 21, scaled=100)
    assert len(mh) == 48504
    assert np.array_equal(mh.mins, np.sort(want["mins"]))
    assert mh.md5sum() == want["md5sum"]


def test_num_golden():
    fa = golden("num", "genome-s10.fa.gz")
    sigs = [s for s in oracle.read_sig_json(fa + ".sig") if s["molecule"].lower() == "dna"]
    assert sigs
    for want in sigs:
        mh = _sketch_fasta(fa, want["ksize"], num=want["num"])
        assert np.array_equal(mh.mins, np.sort(want["mins"])), want["ksize"]
        assert mh.md5sum() == want["md5sum"]


def test_knowngood_genes():
    # tests/test_sourmash_compute.py:858-897 (k=21, num=500, --singleton; second record by name)
    recs = sorted(oracle.read_fasta(golden("genes", "ecoli.genes.fna")), key=lambda r: r[0])
    good, = oracle.read_sig_json(golden("genes", "benchmark.dna.sig"))
    mh = OracleMinHash(good["num"], good["ksize"])
    mh.add_sequence(recs[1][1], force=True)
    assert np.array_equal(mh.mins, np.sort(good["mins"]))


# ---- compare -----------------------------------------------------------------
def _load_mh(d):
    scaled = oracle.scaled_for_max_hash(d["max_hash"]) if d["max_hash"] else 0
    num = 0 if d["max_hash"] else d["num"]          # minhash.rs:150
    ab = "abundances" in d
    mh = OracleMinHash(num, d["ksize"], scaled=scaled, seed=d["seed"], track_abundance=ab)
    if ab:
        for h, a in sorted(zip(d["mins"].tolist(), d["abundances"].tolist())):
            mh.add_hash_with_abundance(h, a)
    else:
        mh.add_many(np.sort(d["mins"]))
    return mh


def test_compare_demo_matrix():
    # tests/test_compare.py:48-63 (num=500 sketches; exact k/500 values)
    files = sorted(glob.glob(golden("demo", "*.sig")))
    mhs = [_load_mh(oracle.read_sig_json(f)[0]) for f in files]
    n = len(mhs)
    got = np.ones((n, n))
    for i in range(n):
        for j in range(i + 1, n):
            got[i, j] = got[j, i] = mhs[i].similarity(mhs[j])
    want = np.array([
        [1.0, 0.356, 0.078, 0.086, 0.0, 0.0, 0.0],
        [0.356, 1.0, 0.072, 0.078, 0.0, 0.0, 0.0],
        [0.078, 0.072, 1.0, 0.074, 0.0, 0.0, 0.0],
        [0.086, 0.078, 0.074, 1.0, 0.0, 0.0, 0.0],
        [0.0, 0.0, 0.0, 0.0, 1.0, 0.382, 0.364],
        [0.0, 0.0, 0.0, 0.0, 0.382, 1.0, 0.386],
        [0.0, 0.0, 0.0, 0.0, 0.364, 0.386, 1.0]])
    np.testing.assert_array_equal(got, want)


def test_scaled_on_real_data():
    # tests/test_jaccard.py:207-232
    a = _load_mh(oracle.read_sig_json(golden("scaled100", "GCF_000005845.2_ASM584v2_genomic.fna.gz.sig.gz"))[0])
    b = _load_mh(oracle.read_sig_json(golden("scaled100", "GCF_000006945.1_ASM694v1_genomic.fna.gz.sig.gz"))[0])
    assert round(a.similarity(b), 5) == 0.01644
    a2, b2 = a.downsample_scaled(1000), b.downsample_scaled(1000)
    assert round(a2.similarity(b2), 5) == 0.01874
    assert round(b2.similarity(a2), 5) == 0.01874
    a3, b3 = a2.downsample_scaled(10000), b2.downsample_scaled(10000)
    assert a3.similarity(b3) == 0.01
    # downsample=True path agrees (minhash.rs:688-696)
    assert round(a.similarity(b2, downsample=True), 5) == 0.01874
    with pytest.raises(oracle.OracleError) as e:
        a.similarity(b2)
    assert e.value.code == 103       # MismatchScaled


def test_csr_compare_matches_pairwise():
    a = _load_mh(oracle.read_sig_json(golden("pairs", "47.fa.sig"))[0])
    b = _load_mh(oracle.read_sig_json(golden("pairs", "63.fa.sig"))[0])
    hashes, offsets = oracle.make_csr([a.mins, b.mins, a.mins, np.zeros(0, np.uint64)])
    common, jac = oracle.compare_all_pairs(hashes, offsets, nthreads=2)
    assert common[0, 1] == a.count_common(b) == common[1, 0]
    assert jac[0, 1] == a.jaccard(b)
    assert jac[0, 2] == 1.0 and common[0, 2] == len(a)
    assert jac[0, 3] == 0.0 and jac[3, 3] == 1.0      # compare.py:33 diagonal of ones


def test_angular_similarity_real_data():
    a = _load_mh(oracle.read_sig_json(golden("pairs", "track_abund_47.fa.sig"))[0])
    b = _load_mh(oracle.read_sig_json(golden("pairs", "track_abund_63.fa.sig"))[0])
    s = a.similarity(b)
    assert 0.0 < s < 1.0
    assert a.similarity(a) == 1.0
    assert a.similarity(b, ignore_abundance=True) == a.jaccard(b)


# ---- gather --------------------------------------------------------------------
GOLDEN_GATHER = [("NC_003198.1", 487), ("NC_000853.1", 192), ("NC_011978.1", 169), ("NC_002163.1", 157),
                 ("NC_003197.2", 152), ("NC_009486.1", 92), ("NC_006905.1", 76), ("NC_011080.1", 59),
                 ("NC_011274.1", 42), ("NC_006511.1", 31), ("NC_011294.1", 7), ("NC_004631.1", 2)]


def test_gather_golden():
    # tests/test_index_protocol.py:1057-1097
    q = [s for s in oracle.read_sig_json(golden("gather", "combined.sig")) if s["ksize"] == 21][0]
    files = sorted(glob.glob(golden("gather", "GCF_*.sig")))
    db = [[s for s in oracle.read_sig_json(f) if s["ksize"] == 21][0] for f in files]
    scaled = oracle.scaled_for_max_hash(q["max_hash"])
    assert scaled == 10000 and len(q["mins"]) == 1466
    hashes, offsets = oracle.make_csr([np.sort(d["mins"]) for d in db])
    res = oracle.gather(np.sort(q["mins"]), hashes, offsets, threshold_bp=0, scaled=scaled)
    got = [(db[i]["name"].split()[0], n) for i, n in res]
    assert got == GOLDEN_GATHER


def test_gather_counter_trace():
    # tests/test_index.py:1581-1679: query 0..19, matches 0-9 / 7-14 / 13-16 -> 10, 5, 2
    q = np.arange(0, 20, dtype=np.uint64)
    m = [np.arange(0, 10), np.arange(7, 15), np.arange(13, 17)]
    hashes, offsets = oracle.make_csr(m)
    assert oracle.gather(q, hashes, offsets, threshold_bp=0, scaled=1) == [(0, 10), (1, 5), (2, 2)]
    # threshold: stop when best overlap < threshold_bp / scaled
    assert oracle.gather(q, hashes, offsets, threshold_bp=5, scaled=1) == [(0, 10), (1, 5)]
    assert oracle.gather(q, hashes, offsets, threshold_bp=6, scaled=1) == [(0, 10)]
    # ties -> first inserted
    hashes, offsets = oracle.make_csr([np.arange(10, 15), np.arange(0, 5), np.arange(0, 3)])
    assert oracle.gather(q, hashes, offsets, scaled=1) == [(0, 5), (1, 5)]


def test_synth_dna_is_reproducible():
    a = oracle.synth_dna(0, 1000, seed=42, record_len=99)
    b = oracle.synth_dna(500, 500, seed=42, record_len=99)
    assert np.array_equal(a[500:], b)
    assert set(np.unique(a).tolist()) <= set(b"ACGT\n")
    assert (a[99::100] == ord("\n")).all() and (a[:99] != ord("\n")).all()


# ---- protein / dayhoff / hp: the oracle's restatement, pinned -----------------
def _prot_mh(moltype, k, **kw):
    "OracleMinHash with the STORED ksize (3 x residues), like MinHash(..., is_protein=True) does (minhash.py:225-241)"
    return oracle.OracleMinHash(kw.pop("n", 0), k * 3, hash_function=oracle.HF_BY_MOLTYPE[moltype], **kw)


def test_protein_known_answers():
    # tests/test_minhash.py:290-452 of the reference
    for moltype, want in (("protein", 4), ("dayhoff", 4), ("hp", 1)):
        mh = _prot_mh(moltype, 2, n=10)
        for _ in range(3):
            mh.add_protein("AGYYG")
        assert len(mh) == want, moltype
    mh = _prot_mh("dayhoff", 7, scaled=1)
    mh.add_protein("CADHIFC")
    assert list(mh.mins) == [oracle.hash_murmur("abcdefa")]
    assert list(oracle.seq_to_hashes_protein("CADHIFC", 7, "dayhoff")) == [oracle.hash_murmur("abcdefa")]
    assert list(oracle.seq_to_hashes_protein("CADHIF*", 7, "dayhoff")) == [oracle.hash_murmur("abcdef*")]     # stop codon kept
    assert list(oracle.seq_to_hashes_protein("ANA", 3, "hp")) == [oracle.hash_murmur("hph")]
    assert list(oracle.seq_to_hashes_protein("AN*", 3, "hp")) == [oracle.hash_murmur("hp*")]
    assert list(oracle.seq_to_hashes_protein("ag", 9, "protein")) == []                                    # :454-458 short
    # translation: tests/test_minhash.py:363-410
    assert [oracle.translate_codon(c) for c in ("TCT", "TC", "T", "TCN", "TAA", "TGA", "TGG", "ATG", "ATN", "NTC", "TC?")] == \
        ["S", "S", "X", "S", "*", "*", "W", "M", "X", "X", "X"]
    for moltype in ("protein", "dayhoff", "hp"):
        mh = _prot_mh(moltype, 2, n=10)
        mh.add_sequence("ACTGAC")                     # frames: TD / LT* ... -> 2 distinct windows overall (:372-388)
        assert len(mh) == 2, moltype
    # ACTGAC: forward frame 0 = T D, reverse complement GTCAGT frame 0 = V S
    want = {oracle.hash_murmur("TD"), oracle.hash_murmur("VS")}
    assert set(oracle.seq_to_hashes_protein("ACTGAC", 2, "protein", is_protein=False).tolist()) == want
    assert list(oracle.seq_to_hashes_protein("ACTGA", 2, "dayhoff", is_protein=False)) == []                # :283-287 shorter than 3k


def test_protein_golden_sketches():
    # six-frame translation of a genome: the reference's own sketches (k=21/30 -> 7/10 residues, num=500)
    recs = list(oracle.read_fasta(golden("num", "genome-s10.fa.gz")))
    golden_prot = [s for s in oracle.read_sig_json(golden("num", "genome-s10.fa.gz.sig")) if s["molecule"] == "protein"]
    assert sorted(s["ksize"] for s in golden_prot) == [21, 30]
    for want in golden_prot:
        mh = oracle.OracleMinHash(want["num"], want["ksize"], hash_function=2)
        for _, seq in recs:
            mh.add_sequence(seq, force=True)
        assert mh.md5sum() == want["md5sum"] and np.array_equal(mh.mins, np.sort(want["mins"]))
    # tests/test_sourmash_compute.py:811-930: protein input and translated genes vs independent restatements
    faa = list(oracle.read_fasta(golden("genes", "ecoli.faa")))
    fna = list(oracle.read_fasta(golden("genes", "ecoli.genes.fna")))
    def sketch(records, translate):
        out = {}
        for name, seq in records:
            mh = oracle.OracleMinHash(500, 21, hash_function=2)
            (mh.add_sequence if translate else mh.add_protein)(seq)
            out[name.split()[0]] = mh
        return out
    aa, tr = sketch(faa, False), sketch(fna, True)
    good_aa = next(s for s in oracle.read_sig_json(golden("genes", "benchmark.input_prot.sig")))
    good_tr = next(s for s in oracle.read_sig_json(golden("genes", "benchmark.prot.sig")))
    assert np.array_equal(aa["NP_414543.1"].mins, np.sort(good_aa["mins"]))
    assert np.array_equal(tr["gi|556503834:337-2799"].mins, np.sort(good_tr["mins"]))
    def jac(a, b):
        c, u = oracle.intersection_size(a.mins, b.mins)
        merged = np.union1d(a.mins, b.mins)[:500]                    # num rule (minhash.rs:593-621)
        common = np.intersect1d(np.intersect1d(a.mins, b.mins), merged).size
        return common / max(1, min(500, merged.size))
    assert round(jac(aa["NP_414544.1"], tr["gi|556503834:2801-3733"]), 3) == 0.166
    assert round(jac(aa["NP_414543.1"], tr["gi|556503834:337-2799"]), 3) == 0.174
    assert jac(aa["NP_414543.1"], tr["gi|556503834:2801-3733"]) == 0.0
