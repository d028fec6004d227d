"""CPU-only checks of the C-ABI library and the host layer: the library loads,
exports every symbol include/sourmash_amd.h declares, host container logic
matches the oracle / golden fixtures, and the GPU-only operations fail loudly
(no CPU fallback) when no device is present."""
import ctypes as C
import glob
import pickle

import numpy as np
import pytest

import oracle
from conftest import golden


@pytest.fixture(scope="module")
def sm():
    import sourmash_amd
    return sourmash_amd


def test_every_declared_symbol_is_exported(sm):
    from sourmash_amd._lowlevel import LIBPATH, parse_header
    funcs, consts = parse_header()
    assert len(funcs) > 100 and consts["SOURMASH_ERROR_CODE_INVALID_DNA"] == 1101
    cdll = C.CDLL(LIBPATH)
    missing = [name for name in funcs if not hasattr(cdll, name)]
    assert not missing, missing
    # the hot-path minimum export list of SURVEY.md section 8(b)
    for name in ("kmerminhash_add_sequence", "kmerminhash_seq_to_hashes", "kmerminhash_count_common",
                 "kmerminhash_intersection", "kmerminhash_similarity", "signature_add_sequence",
                 "signatures_load_buffer", "signatures_save_buffer", "computeparams_set_ksizes", "hash_murmur",
                 "sourmash_err_get_last_code", "smgpu_sketch_dna_raw", "smgpu_compare_raw", "smgpu_overlap_raw"):
        assert name in funcs


def test_error_codes_match_reference_header(sm):
    # include/sourmash.h:19-53 values are ABI; exceptions map like src/sourmash/exceptions.py:136-153
    from sourmash_amd.exceptions import exceptions_by_code
    from sourmash_amd._lowlevel import lib
    assert lib.SOURMASH_ERROR_CODE_MISMATCH_K_SIZES == 101 and lib.SOURMASH_ERROR_CODE_SERDE_ERROR == 100004
    assert exceptions_by_code[1101] is ValueError and exceptions_by_code[103] is ValueError
    assert exceptions_by_code[1104] is ValueError
    assert exceptions_by_code[2].__name__ == "Internal" and issubclass(exceptions_by_code[100004], sm.exceptions.SourmashError)


def test_hash_murmur_host(sm):
    assert sm.hash_murmur("ACG") == 1731421407650554201 == oracle.hash_murmur("ACG")
    assert sm.hash_murmur("ACG", 43) != 1731421407650554201
    rng = np.random.default_rng(0)
    for n in (0, 1, 7, 8, 9, 15, 16, 17, 31, 32, 33, 47, 48, 64, 100):
        s = bytes(rng.integers(1, 255, size=n, dtype=np.uint8))
        for seed in (42, 0, 2**32 - 1):
            assert sm.hash_murmur(s, seed) == oracle.hash_murmur(s, seed), (n, seed)


def test_container_semantics_vs_oracle(sm):
    rng = np.random.default_rng(1)
    vals = rng.integers(1, 2**63, size=3000, dtype=np.uint64) * np.uint64(2)
    for kw in (dict(n=0, scaled=4), dict(n=100, scaled=0), dict(n=0, scaled=1)):
        mh = sm.MinHash(kw["n"], 21, scaled=kw["scaled"])
        om = oracle.OracleMinHash(kw["n"], 21, scaled=kw["scaled"])
        mh.add_many(vals); om.add_many(vals)
        assert np.array_equal(mh._mins_array(), om.mins) and mh.md5sum() == om.md5sum()
        mh.remove_many(vals[:500]); om.remove_many(vals[:500])
        assert np.array_equal(mh._mins_array(), om.mins)
    # keep rule inclusive (tests/test_minhash.py:475-490)
    mh = sm.MinHash(0, 4, max_hash=35)
    mx = mh._max_hash
    for h in (10, 20, 30, mx, mx + 1):
        mh.add_hash(h)
    assert list(mh.hashes) == sorted({10, 20, 30, mx})
    # abundance bookkeeping
    mh = sm.MinHash(0, 21, scaled=1, track_abundance=True)
    mh.set_abundances({5: 2, 9: 1, 7: 3})
    mh.add_hash(9); mh.add_hash_with_abundance(11, 4)
    assert dict(mh.hashes) == {5: 2, 7: 3, 9: 2, 11: 4}
    mh.set_abundances({5: 0}, clear=False)                 # abundance 0 removes (minhash.rs:329-332)
    assert 5 not in mh.hashes
    with pytest.raises(RuntimeError):
        sm.MinHash(0, 21, scaled=1).add_hash_with_abundance(3, 1)
    mh2 = sm.MinHash(0, 21, scaled=1)
    mh2.add_hash(3)
    with pytest.raises(RuntimeError):
        mh2.track_abundance = True


def test_merge_downsample_copy_pickle(sm):
    a = sm.MinHash(0, 31, scaled=2, track_abundance=True)
    b = sm.MinHash(0, 31, scaled=2, track_abundance=True)
    a.set_abundances({10: 1, 30: 2, 2**62: 5}); b.set_abundances({30: 3, 40: 1})
    c = a + b
    assert dict(c.hashes) == {10: 1, 30: 5, 40: 1, 2**62: 5} and dict(a.hashes) == {10: 1, 30: 2, 2**62: 5}
    d = c.downsample(scaled=8)
    assert d.scaled == 8 and dict(d.hashes) == {10: 1, 30: 5, 40: 1}
    with pytest.raises(ValueError):
        d.downsample(scaled=2)
    with pytest.raises(ValueError):
        d.downsample(num=5)
    assert c.flatten().track_abundance is False and list(c.flatten().hashes) == [10, 30, 40, 2**62]
    assert pickle.loads(pickle.dumps(c)) == c
    f = c.to_frozen()
    with pytest.raises(TypeError):
        f.add_hash(1)
    assert f.to_mutable() == c and isinstance(f.to_mutable(), sm.MinHash)
    for other in (sm.MinHash(0, 21, scaled=2), sm.MinHash(0, 31, scaled=4), sm.MinHash(0, 31, scaled=2, seed=1)):
        with pytest.raises(ValueError):
            sm.MinHash(0, 31, scaled=2).merge(other)
    assert not sm.MinHash(0, 31, scaled=2).is_compatible(sm.MinHash(0, 21, scaled=2))
    with pytest.raises(ValueError):
        sm.MinHash(0, 31)
    with pytest.raises(ValueError):
        sm.MinHash(10, 31, scaled=5)


def test_signature_json_roundtrip_against_golden(sm):
    path = golden("ecoli", "GCF_000005845.2_ASM584v2_genomic.fna.gz.sig")
    sigs = list(sm.load_signatures_from_json(path))
    want = oracle.read_sig_json(path)
    assert [(s.minhash.ksize, len(s.minhash), s.md5sum()) for s in sigs] == \
           [(w["ksize"], len(w["mins"]), w["md5sum"]) for w in want]
    assert sigs[0].name == "GCF_000005845" and sigs[0].filename.endswith("genomic.fna.gz") and sigs[0].license == "CC0"
    k31 = list(sm.load_signatures_from_json(path, ksize=31))
    assert len(k31) == 1 and k31[0].minhash.ksize == 31
    assert list(sm.load_signatures_from_json(path, select_moltype="protein")) == []
    js = sm.save_signatures_to_json(sigs)
    back = list(sm.load_signatures_from_json(js))
    assert [b.md5sum() for b in back] == [s.md5sum() for s in sigs] and back[1] == sigs[1]
    gz = sm.save_signatures_to_json(sigs, compression=5)
    assert gz[:2] == b"\x1f\x8b" and [b.md5sum() for b in sm.load_signatures_from_json(gz)] == [s.md5sum() for s in sigs]
    # old-format files: num=2^32-1 with max_hash set loads as a scaled sketch (minhash.rs:150); abundances load
    g = list(sm.load_signatures_from_json(golden("gather", "combined.sig"), ksize=21))[0].minhash
    assert g.num == 0 and g.scaled == 10000 and len(g) == 1466
    ab = sm.load_one_signature_from_json(golden("pairs", "track_abund_47.fa.sig")).minhash
    assert ab.track_abundance and sum(ab.hashes.values()) > len(ab)
    with pytest.raises(ValueError):
        sm.load_one_signature_from_json(path)
    with pytest.raises(sm.exceptions.SourmashError):      # SerdeError (code 100004), as in the reference
        list(sm.load_signatures_from_json("[{\"class\": \"sourmash_signature\", \"signatures\": 5}]", do_raise=True))
    assert list(sm.load_signatures_from_json("not json at all")) == []


def test_params_and_templates(sm):
    from sourmash_amd.sketch import ComputeParameters, parse_params_str
    assert parse_params_str("k=21,k=31,scaled=1000,abund")[1] == dict(ksize=[21, 31], scaled=1000, num=0, track_abundance=True)
    with pytest.raises(ValueError):
        parse_params_str("k=21,num=500,scaled=10")
    with pytest.raises(ValueError):
        parse_params_str("foo")
    p = ComputeParameters.from_param_str("k=21,k=51,scaled=100,abund,seed=7")
    assert (p.ksizes, p.scaled, p.num_hashes, p.seed, p.track_abundance, p.dna) == ([21, 51], 100, 0, 7, True, True)
    sig = sm.SourmashSignature.from_params(p)
    assert [(m.ksize, m.scaled, m.seed, m.track_abundance) for m in sig.minhashes()] == [(21, 100, 7, True), (51, 100, 7, True)]
    assert ComputeParameters.from_param_str("dna").to_param_str() == "dna,k=31,scaled=1000"       # defaults are left out (command_sketch.py:926-964)
    assert ComputeParameters.from_param_str("protein,k=7,abund,seed=3").to_param_str() == "protein,k=7,scaled=200,abund,seed=3"


def test_gpu_only_operations_fail_loudly_without_a_device(sm):
    if sm.gpu_available():
        pytest.skip("a GPU is present")
    from sourmash_amd.exceptions import SourmashError
    a, b = sm.MinHash(0, 31, scaled=1000), sm.MinHash(0, 31, scaled=1000)
    a.add_many([1, 2, 3]); b.add_many([2, 3, 4])
    for call in (lambda: a.add_sequence("ACGT" * 20, True), lambda: a.count_common(b), lambda: a.jaccard(b),
                 lambda: a & b, lambda: a.seq_to_hashes("ACGT" * 20)):
        with pytest.raises(SourmashError) as e:
            call()
        assert "no HIP device" in str(e.value)
    with pytest.raises(RuntimeError):
        from sourmash_amd import device as smd
        smd.synth_dna(10)


def test_handle_arrays_of_object_lists(sm):
    "utils.objptr_array: what SketchSet / compare hand to the library for a list of objects -- the handles in order, a closed object refused"
    from sourmash_amd.utils import objptr_array
    mhs = [sm.MinHash(0, 21 + i, scaled=1000) for i in range(257)]
    arr, keep = objptr_array(mhs)
    assert len(arr) == 257 and all(int(arr[i]) == mhs[i]._objptr for i in range(257))
    assert keep.dtype == np.uintp and keep.ctypes.data == C.addressof(arr)             # the C array is a view of the numpy block
    empty, _ = objptr_array([])
    assert len(empty) == 1 and not empty[0]                                            # (callers pass n = 0 beside it)

    class Closed:
        _objptr = 0
    with pytest.raises(RuntimeError, match="closed"):
        objptr_array(mhs[:3] + [Closed()])
