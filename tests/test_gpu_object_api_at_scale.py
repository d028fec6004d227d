"""What only a benchmark exercised before: the pinned transfer ring with transfers of several chunks, and
the object-list entry points at sizes where they take it -- compare_all_pairs over thousands of signature objects and a
SketchSet of 20,000 sketch objects through gather, each against the oracle.  Run with -m gpu."""
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


def roundtrip(pieces, out):
    from sourmash_amd._lowlevel import lib
    from sourmash_amd.utils import rustcall
    n = len(pieces)
    ptrs = (C.c_void_p * max(n, 1))(*[p.ctypes.data if p.size else None for p in pieces])
    lens = (C.c_uint64 * max(n, 1))(*[p.nbytes for p in pieces])
    rustcall(lib.smgpu_xfer_roundtrip, ptrs, lens, n, C.c_void_p(out.ctypes.data), out.nbytes)


def test_the_pinned_ring_moves_several_chunks_both_ways(sm):
    "csrc/hostxfer.hpp: 32 MiB chunks, two slots -- a slot is reused from the third chunk on; odd sizes, empty pieces, pieces across chunk borders"
    from sourmash_amd._lowlevel import lib
    rng = np.random.default_rng(7)
    CH = 32 << 20
    sizes = [0, 1, 5, CH - 3, 7, 0, 0, CH + 11, 3 * CH + 1, 12345, 0, 2 * CH, 999_983, CH // 2 + 1, 1]      # 8 chunks and a bit: 268 MB
    pieces = [rng.integers(0, 256, size=s, dtype=np.uint8) for s in sizes]
    want = np.concatenate(pieces)
    assert want.nbytes > 200_000_000
    out = np.zeros(want.nbytes, dtype=np.uint8)                       # a pageable destination: through the ring
    roundtrip(pieces, out)
    assert np.array_equal(out, want)
    # an already pinned destination is filled by one direct copy
    ptr = lib.smgpu_host_alloc(want.nbytes)
    assert ptr
    try:
        pinned = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(want.nbytes,))
        pinned[:] = 0
        roundtrip(pieces, pinned)
        assert np.array_equal(pinned, want)
    finally:
        lib.smgpu_host_free(ptr)
    # many small pieces straddling chunk borders (what a list of sketch objects looks like), with a tail of one byte
    small = [rng.integers(0, 256, size=int(rng.integers(0, 90_000)), dtype=np.uint8) for _ in range(2500)] + [np.array([7], dtype=np.uint8)]
    want = np.concatenate(small)
    assert want.nbytes > 3 * CH
    out = np.empty(want.nbytes, dtype=np.uint8)
    roundtrip(small, out)
    assert np.array_equal(out, want)
    for tiny in ([], [np.zeros(0, dtype=np.uint8)], [np.array([1, 2, 3], dtype=np.uint8)]):
        want = np.concatenate(tiny) if tiny else np.zeros(0, dtype=np.uint8)
        out = np.zeros(want.nbytes, dtype=np.uint8)
        roundtrip(tiny, out)
        assert np.array_equal(out, want)


def test_compare_all_pairs_over_3000_signature_objects(sm):
    "compare.py:326-358 of the reference at a size where the object list goes through the pinned ring (120 MB of hashes, a 72 MB matrix)"
    from sourmash_amd.compare import compare_all_pairs
    from sourmash_amd.synth import synth_sketches
    sk = synth_sketches(3000, seed=99)
    sigs = []
    for i, h in enumerate(sk):
        mh = sm.MinHash(0, 31, scaled=1000)
        mh.add_many(h)
        sigs.append(sm.SourmashSignature(mh, name=f"s{i}"))
    got = compare_all_pairs(sigs, ignore_abundance=True)
    wc, wj = oracle.compare_all_pairs(*oracle.make_csr(sk), nthreads=8)
    want = wj.copy()
    np.fill_diagonal(want, 1.0)
    assert got.shape == (3000, 3000)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


def test_gather_over_a_sketchset_of_20000_objects(sm):
    "SketchSet(list of sketch objects): 20,000 objects packed through the ring, then the whole min-set-cover on the device, against oracle.gather"
    from sourmash_amd.index import SketchSet
    from sourmash_amd.synth import synth_gather
    qh, dbh = synth_gather(n_query=300_000, n_db=20_000, db_size=1500)
    mhs = []
    for h in dbh:
        mh = sm.MinHash(0, 31, scaled=1000)
        mh.add_many(h)
        mhs.append(mh)
    q = sm.MinHash(0, 31, scaled=1000)
    q.add_many(qh)
    db = SketchSet(mhs)
    got = db.gather(q, threshold_bp=50_000)
    gh, go = oracle.make_csr(dbh)
    want = oracle.gather(qh, gh, go, threshold_bp=50_000, scaled=1000, nthreads=8)
    assert len(got) > 100 and got == [(int(i), int(c)) for i, c in want]
