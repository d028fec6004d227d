"""The per-record API with deferred hashing (capi.cpp: add_sequence queues, accessors settle): the observable behaviour
must stay that of the reference's streaming add_sequence (src/core/src/signature.rs:38-58) -- same sketch whatever
the interleaving with other calls, InvalidDNA raised by the offending call after the k-mers in front of the bad one were
added (tests/test_minhash.py:711-719,2594-2602 of the reference).  Run with -m gpu."""
import ctypes as C

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sm():
    import torch  # noqa: F401
    import sourmash_amd
    assert sourmash_amd.gpu_available()
    return sourmash_amd


def _pending(mh):
    from sourmash_amd._lowlevel import lib
    return int(lib.smgpu_minhash_pending_bytes(mh._objptr))


def _records(n, length, seed=3):
    rng = np.random.default_rng(seed)
    return [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), length)).decode() for _ in range(n)]


def test_queue_settles_on_access_and_equals_streaming(sm):
    recs = _records(300, 180) + ["ACGT" * 3, "A" * 30, "", "acgtacgtacgtacgtacgtacgtacgtacgtacgtacgt"]
    mh = sm.MinHash(0, 31, scaled=20)
    want = oracle.OracleMinHash(0, 31, scaled=20)
    for r in recs:
        mh.add_sequence(r)
        want.add_sequence(r.encode())
    assert _pending(mh) > 0                                       # nothing has been hashed yet
    assert len(mh) == len(want)                                   # the size getter settles
    assert _pending(mh) == 0
    assert np.array_equal(mh._mins_array(), want.mins)
    assert mh.md5sum() == want.md5sum()


def test_invalid_dna_raises_from_the_offending_call_with_the_prefix_added(sm):
    good1, good2 = _records(2, 200, seed=9)
    bad = good1[:70] + "R" + good2[:50]                           # k-mers 0..39 are fine, k-mer 40 covers the R
    mh = sm.MinHash(0, 31, scaled=1)
    want = oracle.OracleMinHash(0, 31, scaled=1)
    mh.add_sequence(good1)
    want.add_sequence(good1.encode())
    with pytest.raises(ValueError) as err:
        mh.add_sequence(bad)                                      # raised here, not when the queue is settled
    assert "invalid DNA character in input k-mer: " + bad[40:71].upper() in str(err.value)
    with pytest.raises(oracle.InvalidDNA):
        want.add_sequence(bad.encode())
    mh.add_sequence(good2)
    want.add_sequence(good2.encode())
    assert np.array_equal(mh._mins_array(), want.mins)            # includes the 40 k-mers in front of the bad one
    forced = sm.MinHash(0, 31, scaled=1)
    forced.add_sequence(bad, force=True)
    ow = oracle.OracleMinHash(0, 31, scaled=1)
    ow.add_sequence(bad.encode(), force=True)
    assert np.array_equal(forced._mins_array(), ow.mins) and len(forced) == len(bad) - 30 - 31
    short = sm.MinHash(0, 31, scaled=1)
    short.add_sequence("ACGTNNNN")                                # shorter than k: silently nothing (signature.rs:206-210)
    assert len(short) == 0


def test_operations_that_do_not_commute_settle_first(sm):
    a, b, c = _records(3, 400, seed=21)
    def fresh(**kw):
        return sm.MinHash(0, 21, scaled=1, **kw)
    ha = oracle.OracleMinHash(0, 21, scaled=1)
    ha.add_sequence(a.encode())
    hb = oracle.OracleMinHash(0, 21, scaled=1)
    hb.add_sequence(b.encode())
    # remove after add
    mh = fresh()
    mh.add_sequence(a)
    mh.remove_many(ha.mins[:50].tolist())
    assert np.array_equal(mh._mins_array(), ha.mins[50:])
    # clear drops what is queued
    mh = fresh()
    mh.add_sequence(a)
    mh.clear()
    mh.add_sequence(b)
    assert np.array_equal(mh._mins_array(), hb.mins)
    # add_hash in between commutes
    mh = fresh()
    mh.add_sequence(a)
    mh.add_hash(7)
    mh.add_sequence(b)
    assert np.array_equal(mh._mins_array(), np.union1d(np.union1d(ha.mins, hb.mins), np.array([7], dtype=np.uint64)))
    # merge / copy / count_common with queues on both sides
    x, y = fresh(), fresh()
    x.add_sequence(a)
    y.add_sequence(b)
    y.add_sequence(a[:100])
    z = x.copy()
    assert np.array_equal(z._mins_array(), ha.mins)
    assert x.count_common(y) == len(np.intersect1d(ha.mins, y._mins_array()))
    x.merge(y)
    assert np.array_equal(x._mins_array(), np.union1d(ha.mins, y._mins_array()))
    # abundances add up across queued records and across settles
    w = fresh(track_abundance=True)
    w.add_sequence(a)
    w.add_sequence(a)
    assert set(w.hashes.values()) >= {2}
    w.add_sequence(a)
    ow = oracle.OracleMinHash(0, 21, scaled=1, track_abundance=True)
    for _ in range(3):
        ow.add_sequence(a.encode())
    assert w.hashes == dict(zip(ow.mins.tolist(), ow.abunds.tolist()))
    # bottom-k
    n = sm.MinHash(50, 21)
    on = oracle.OracleMinHash(50, 21)
    for r in (a, b, c):
        n.add_sequence(r)
        on.add_sequence(r.encode())
    assert np.array_equal(n._mins_array(), on.mins) and len(n) == 50


def test_queue_flushes_itself_when_it_is_large(sm):
    recs = _records(90, 500_000, seed=4)                           # 45 MB of records: crosses the 32 MiB threshold once
    mh = sm.MinHash(0, 31, scaled=1000)
    seen_small = False
    for r in recs:
        before = _pending(mh)
        mh.add_sequence(r, True)
        if _pending(mh) < before:
            seen_small = True
    assert seen_small
    whole = sm.MinHash(0, 31, scaled=1000)
    whole.add_sequence_buffer("\n".join(recs).encode())
    assert mh == whole and len(mh) > 40_000


def test_signature_fan_out_and_c_string_semantics(sm):
    from sourmash_amd._lowlevel import lib
    recs = _records(40, 300, seed=13)
    sig_mhs = [sm.MinHash(0, k, scaled=10) for k in (21, 31, 51)]
    sigs = [sm.SourmashSignature(mh, name="x") for mh in sig_mhs]
    for r in recs:
        for s in sigs:
            s.add_sequence(r)
    for s, k in zip(sigs, (21, 31, 51)):
        ow = oracle.OracleMinHash(0, k, scaled=10)
        for r in recs:
            ow.add_sequence(r.encode())
        assert np.array_equal(s.minhash._mins_array(), ow.mins)
    # a NUL ends the record (ffi/minhash.rs:53-59 takes a C string)
    mh = sm.MinHash(0, 21, scaled=1)
    raw = (recs[0][:100] + "\0" + recs[1]).encode()
    assert lib.smgpu_minhash_add_sequence_rc(mh._objptr, raw, len(raw), False) == 0
    ow = oracle.OracleMinHash(0, 21, scaled=1)
    ow.add_sequence(recs[0][:100].encode())
    assert np.array_equal(mh._mins_array(), ow.mins)


def test_c_method_and_ctypes_binding_agree(sm):
    """MinHash.add_sequence is the C method of csrc/fastcall.c; its ctypes twin (minhash.py: _AddSequencePython) must
    see the same sketch and raise the same errors for every argument form the reference accepts (minhash.py:70-85)."""
    from sourmash_amd import minhash as mhmod
    assert mhmod._fastcall is not None, "sourmash_amd/_fastcall*.so is not built (make -C sourmash_amd/csrc)"
    assert type(sm.MinHash.add_sequence).__name__ == "method_descriptor"
    recs = _records(50, 170, seed=21)
    forms = [lambda r: r, lambda r: r.encode(), lambda r: bytearray(r.encode()), lambda r: memoryview(r.encode()),
             lambda r: r.lower(), lambda r: r[:60] + "\0" + r[60:]]          # the last one: a NUL ends the record
    a, b = sm.MinHash(0, 21, scaled=5), sm.MinHash(0, 21, scaled=5)
    want = oracle.OracleMinHash(0, 21, scaled=5)
    for i, r in enumerate(recs):
        arg = forms[i % len(forms)](r)
        a.add_sequence(arg)
        mhmod._AddSequencePython.add_sequence(b, arg)
        raw = bytes(arg) if not isinstance(arg, str) else arg.encode()
        want.add_sequence(raw.split(b"\0")[0])
    a.add_sequence(sequence=recs[0], force=True)                  # keywords
    mhmod._AddSequencePython.add_sequence(b, sequence=recs[0], force=True)
    want.add_sequence(recs[0].encode(), force=True)
    assert np.array_equal(a._mins_array(), want.mins) and np.array_equal(b._mins_array(), want.mins)
    bad = recs[1][:40] + "N" + recs[2][:40]
    for call in (a.add_sequence, lambda s, f=False: mhmod._AddSequencePython.add_sequence(b, s, f)):
        with pytest.raises(ValueError) as err:
            call(bad)
        assert "invalid DNA character in input k-mer: " + bad[20:41].upper() in str(err.value)
        call(bad, True)                                           # force: the k-mers over the N are skipped
    want2 = oracle.OracleMinHash(0, 21, scaled=5)
    with pytest.raises(oracle.InvalidDNA):
        want2.add_sequence(bad.encode())
    want2.add_sequence(bad.encode(), force=True)
    want.merge(want2)
    assert np.array_equal(a._mins_array(), want.mins) and np.array_equal(b._mins_array(), want.mins)
    for wrong in (3.5, None, ["ACGT"]):
        with pytest.raises(TypeError):
            a.add_sequence(wrong)
    frozen = a.to_frozen()
    with pytest.raises(TypeError):
        frozen.add_sequence("ACGT" * 10)                          # FrozenMinHash stays read-only


def test_pending_sketches_straight_into_sketchset_and_counter(sm):
    """A sketch whose records are still queued handed to SketchSet / the gather counter: both entry points take the
    device context themselves, and settling the queue needs it too (round-2 advisor finding: self-deadlock on a
    non-recursive mutex).  Runs under a watchdog so that a regression fails instead of hanging the suite."""
    import threading
    from sourmash_amd.index import SketchSet, _DeviceCounter
    recs = _records(40, 400, seed=21)
    result = {}

    def body():
        rows, wants = [], []
        for i in range(4):
            mh = sm.MinHash(0, 21, scaled=5)
            want = oracle.OracleMinHash(0, 21, scaled=5)
            for r in recs[i * 8:(i + 1) * 8 + 4]:
                mh.add_sequence(r)
                want.add_sequence(r.encode())
            assert _pending(mh) > 0
            rows.append(mh)
            wants.append(want)
        sset = SketchSet(rows)                                    # settles every row on the way in
        assert [int(x) for x in sset.sizes] == [len(w) for w in wants]
        query = sm.MinHash(0, 21, scaled=5)
        qwant = oracle.OracleMinHash(0, 21, scaled=5)
        for r in recs[:30]:
            query.add_sequence(r)
            qwant.add_sequence(r.encode())
        assert _pending(query) > 0
        counter = _DeviceCounter(sset, query)                     # and the query
        got = [int(x) for x in counter.values()]
        result["got"] = got
        result["want"] = [int(oracle.intersection_size(qwant.mins, w.mins)[0]) for w in wants]

    t = threading.Thread(target=body, daemon=True)
    t.start()
    t.join(120)
    assert not t.is_alive(), "deadlock: SketchSet / counter construction from a sketch with queued records"
    assert result["got"] == result["want"] and max(result["want"]) > 0
