"""protein / dayhoff / hp sketches on the GPU against the reference's known answers
(tests/test_minhash.py:221-460), its golden sketches (genome-s10 translated; benchmark.*prot*.sig) and the oracle.
Run with -m gpu."""
import numpy as np
import pytest

import oracle
from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sm():
    import torch  # noqa: F401
    import sourmash_amd
    assert sourmash_amd.gpu_available()
    return sourmash_amd


@pytest.fixture(params=[True, False])
def track_abundance(request):
    return request.param


def _mh(sm, moltype, k, n=0, **kw):
    return sm.MinHash(n, k, is_protein=moltype == "protein", dayhoff=moltype == "dayhoff", hp=moltype == "hp", **kw)


def test_known_answers(sm, track_abundance):
    # tests/test_minhash.py:290-452
    for moltype, want in (("protein", 4), ("dayhoff", 4), ("hp", 1)):
        mh = _mh(sm, moltype, 2, n=10, track_abundance=track_abundance)
        assert mh.moltype == moltype
        mh.add_protein("AGYYG")
        mh.add_protein("AGYYG")
        mh.add_protein(b"AGYYG")
        assert len(mh.hashes) == want, moltype
    mh = _mh(sm, "dayhoff", 7, scaled=1, track_abundance=track_abundance)
    mh.add_protein("CADHIFC")
    assert list(mh.hashes) == [sm.hash_murmur("abcdefa")]
    assert list(mh.seq_to_hashes("CADHIFC", is_protein=True)) == [sm.hash_murmur("abcdefa")]
    mh = mh.copy_and_clear()
    mh.add_protein("CADHIF*")                                      # stop codons are residues like any other
    assert list(mh.hashes) == [sm.hash_murmur("abcdef*")] == list(mh.seq_to_hashes("CADHIF*", is_protein=True))
    mh = _mh(sm, "hp", 3, scaled=1, track_abundance=track_abundance)
    mh.add_protein("ANA")
    assert list(mh.hashes) == [sm.hash_murmur("hph")] == list(mh.seq_to_hashes("ANA", is_protein=True))
    mh = mh.copy_and_clear()
    mh.add_protein("AN*")
    assert list(mh.hashes) == [sm.hash_murmur("hp*")]
    short = _mh(sm, "protein", 9, n=10)
    short.add_protein("AG")                                        # :454-460
    assert len(short) == 0
    # DNA into residue sketches: six frames (:372-430)
    prot = _mh(sm, "protein", 2, n=10, track_abundance=track_abundance)
    prot.add_sequence("ACTGAC")
    assert set(prot.hashes) == {sm.hash_murmur("TD"), sm.hash_murmur("VS")}
    for moltype in ("dayhoff", "hp"):
        mh = _mh(sm, moltype, 2, n=10, track_abundance=track_abundance)
        mh.add_sequence("ACTGAC")
        assert len(mh.hashes) == 2 and set(mh.hashes) != set(prot.hashes)
    assert _mh(sm, "dayhoff", 2, scaled=1).seq_to_hashes("ACTGA") == []    # :283-287: shorter than 3 k
    # module-level helpers (:363-370)
    from sourmash_amd.minhash import translate_codon
    assert [translate_codon(c) for c in ("TCT", "TC", "T", "tcn", "TAA", "ATN")] == ["S", "S", "X", "S", "*", "X"]
    for bad in ("", "TCTA"):
        with pytest.raises(ValueError):
            translate_codon(bad)
    with pytest.raises(ValueError):
        sm.MinHash(0, 21, scaled=1).add_protein("CADHIFCADHIFCADHIFCADHIF")   # a DNA sketch has no residue alphabet
    with pytest.raises(ValueError):
        sm.MinHash(0, 21, scaled=1).seq_to_hashes("CADHIFCADHIFCADHIFCADHIF", is_protein=True)


@pytest.mark.parametrize("moltype", ["protein", "dayhoff", "hp"])
def test_seq_to_hashes_order_and_kmers(sm, moltype):
    # tests/test_minhash.py:221-282 + kmers_and_hashes for residues and for translated DNA
    rng = np.random.default_rng(11)
    dna = "".join(rng.choice(list("ACGTN"), p=[.24, .24, .24, .24, .04], size=400))
    aa = "".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY*XBZ"), size=300))
    k = {"protein": 7, "dayhoff": 11, "hp": 21}[moltype]
    mh = _mh(sm, moltype, k, scaled=1)
    assert mh.seq_to_hashes(aa, is_protein=True) == oracle.seq_to_hashes_protein(aa, k, moltype).tolist()
    assert mh.seq_to_hashes(aa.lower(), is_protein=True) == oracle.seq_to_hashes_protein(aa, k, moltype).tolist()
    want = oracle.seq_to_hashes_protein(dna, k, moltype, is_protein=False).tolist()
    assert mh.seq_to_hashes(dna) == want and len(want) == 2 * (len(dna) - 3 * k + 1)
    assert mh.seq_to_hashes(dna, force=True, bad_kmers_as_zeroes=True) == [0] + want + [0]   # the iterator's two markers
    pairs = list(mh.kmers_and_hashes(dna))
    assert [h for _, h in pairs] == want
    for kmer, h in pairs[:5] + pairs[-5:]:
        one = mh.copy_and_clear()
        one.add_sequence(kmer)
        # a 3k-base window translates to exactly one residue k-mer in frame 0 forward (plus its reverse complement)
        assert h in one.hashes and len(kmer) == 3 * k
    apairs = list(mh.kmers_and_hashes(aa, is_protein=True))
    assert [h for _, h in apairs] == oracle.seq_to_hashes_protein(aa, k, moltype).tolist()
    assert apairs[0][0] == aa[:k]
    with pytest.raises(ValueError):
        mh.add_kmer(dna[:3 * k + 1])
    km = mh.copy_and_clear()
    km.add_kmer(dna[:3 * k])
    assert len(km) >= 1


def test_golden_translated_genome_and_gene_benchmarks(sm):
    from sourmash_amd.sketch import sketch_file
    fa = golden("num", "genome-s10.fa.gz")
    want = {s["ksize"]: s for s in oracle.read_sig_json(fa + ".sig") if s["molecule"] == "protein"}
    sig, = sketch_file(fa, "k=7,k=10,num=500", moltype="protein")              # `sourmash sketch translate`
    got = {mh.ksize: mh for mh in sig.minhashes()}
    assert sorted(got) == [7, 10]
    for k, mh in got.items():
        assert mh.md5sum() == want[k * 3]["md5sum"] and mh.moltype == "protein" and mh.num == 500
    # tests/test_sourmash_compute.py:811-930
    aa = {s.name.split()[0]: s for s in sketch_file(golden("genes", "ecoli.faa"), "k=7,num=500", moltype="protein",
                                                     input_is_protein=True, singleton=True)}
    tr = {s.name.split()[0]: s for s in sketch_file(golden("genes", "ecoli.genes.fna"), "k=7,num=500", moltype="protein",
                                                     singleton=True)}
    good_aa = sm.load_one_signature_from_json(golden("genes", "benchmark.input_prot.sig"))
    good_tr = sm.load_one_signature_from_json(golden("genes", "benchmark.prot.sig"))
    assert aa["NP_414543.1"].similarity(good_aa) == 1.0
    assert tr["gi|556503834:337-2799"].similarity(good_tr) == 1.0
    assert round(aa["NP_414543.1"].similarity(tr["gi|556503834:2801-3733"]), 3) == 0.0
    assert round(aa["NP_414544.1"].similarity(tr["gi|556503834:2801-3733"]), 3) == 0.166
    assert round(aa["NP_414543.1"].similarity(tr["gi|556503834:337-2799"]), 3) == 0.174
    assert round(aa["NP_414544.1"].similarity(tr["gi|556503834:337-2799"]), 3) == 0.0


# (windows of up to 79 residues take the register-window kernel, residue_core.hpp; longer ones the byte-wise kernel: both here)
@pytest.mark.parametrize("moltype,k,scaled", [("protein", 10, 20), ("dayhoff", 16, 20), ("hp", 42, 20), ("protein", 7, 5), ("dayhoff", 33, 20),
                                              ("hp", 79, 20), ("hp", 85, 20)])
def test_scaled_sketches_vs_oracle(sm, moltype, k, scaled, track_abundance):
    rng = np.random.default_rng(5)
    dna = "".join(rng.choice(list("ACGTacgtN"), size=30_000))
    aa = "".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWYX*"), size=20_000))
    hf = oracle.HF_BY_MOLTYPE[moltype]
    for seq, is_protein in ((dna, False), (aa, True)):
        mh = _mh(sm, moltype, k, scaled=scaled, track_abundance=track_abundance)
        want = oracle.OracleMinHash(0, k * 3, scaled=scaled, hash_function=hf, track_abundance=track_abundance)
        for piece in (seq, seq[:5000], seq[100:7000]):                   # repeats -> abundances > 1
            (mh.add_protein if is_protein else mh.add_sequence)(piece)
            (want.add_protein if is_protein else want.add_sequence)(piece)
        assert np.array_equal(mh._mins_array(), want.mins) and len(want) > 50
        if track_abundance:
            assert list(mh.hashes.values()) == want.abunds.tolist()
        assert mh.md5sum() == want.md5sum()


def test_signature_level_and_json_round_trip(sm):
    from sourmash_amd.sketch import ComputeParameters
    params = ComputeParameters.from_param_str("k=7,k=10,scaled=10,abund", default_moltype="protein")
    assert params.ksizes == [21, 30] and params.protein and not params.dna and params.to_param_str().startswith("protein,")
    sig = sm.SourmashSignature.from_params(params)
    sig.add_protein("MKRISTTITTTITITTGNGAGMKRISTTITTTITITTGNGAG")
    sig.name = "p"
    back = list(sm.load_signatures_from_json(sm.save_signatures_to_json([sig])))
    assert sorted(s.minhash.ksize for s in back) == [7, 10] and all(s.minhash.moltype == "protein" for s in back)
    assert {s.minhash.ksize: s.minhash for s in back}[7] == {mh.ksize: mh for mh in sig.minhashes()}[7]
    for default, k, s in (("dayhoff", 16, 200), ("hp", 42, 200), ("protein", 10, 200)):      # command_sketch.py:25-30
        p = ComputeParameters.from_param_str(default, default_moltype=default)
        assert p.ksizes == [k * 3] and p.scaled == s and p.moltype == default
