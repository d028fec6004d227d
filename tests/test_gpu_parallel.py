"""GPU parity of the multi-GPU drivers' device backend (one GPU: world_size 1 end to end, and the
sharded tile kernel driven rank by rank the way N ranks would).  Run with -m gpu."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    import torch  # noqa: F401
    from sourmash_amd import parallel
    return parallel.DeviceBackend()


def test_compare_world1_and_emulated_shards(be):
    import torch
    from sourmash_amd import device as smd, parallel
    from sourmash_amd.synth import synth_sketches
    sk = synth_sketches(333, pool_size=9000)
    n = len(sk)
    h, off = smd.pack_csr(sk)
    wc, wj = oracle.compare_all_pairs(*oracle.make_csr(sk), nthreads=8)
    common, jac = parallel.compare_all_pairs_distributed(h, off, n, be)
    torch.cuda.synchronize()
    assert np.array_equal(common.cpu().numpy().view(np.uint32), wc)
    assert np.array_equal(jac.cpu().numpy().view(np.uint64), wj.view(np.uint64))
    for world in (2, 3, 8):
        n_tiles = (n + 15) // 16
        max_count = (n_tiles + world - 1) // world
        pieces = []
        for r in range(world):
            first, stride, count = parallel.tiles_for_rank(n, world, r)
            local = be.compare_tiles(h, off, n, first, stride, count)
            pad = be.zeros((max_count * 16, n), local.dtype)
            pad[:local.shape[0]] = local
            pieces.append(pad)
        full = parallel.assemble_tiles(pieces, n, world, be)
        be.symmetrize(full, n)
        torch.cuda.synchronize()
        assert np.array_equal(full.cpu().numpy().view(np.uint32), wc), world


def test_gather_world1_device_loop(be):
    import torch
    from sourmash_amd import device as smd, parallel
    from sourmash_amd.synth import synth_gather
    qh, dbh = synth_gather(n_query=80_000, n_db=2500, db_size=500)
    dbh[11] = dbh[5].copy()
    h, off = smd.pack_csr(dbh)
    q = torch.from_numpy(qh.view(np.int64).copy()).cuda()
    fh, foff = oracle.make_csr(dbh)
    for thr in (0, 100_000):
        got = parallel.gather_distributed(q, len(qh), h, off, len(dbh), 0, thr, 1000, be)
        assert got == oracle.gather(qh, fh, foff, threshold_bp=thr, scaled=1000), thr
    # sharded database driven shard by shard: local winners combine through the packed MAX key
    lo, hi = 1200, 2500
    h2, off2 = smd.pack_csr(dbh[lo:hi])
    cnt = be.zeros((hi - lo,), torch.int64)
    be.overlaps(q, len(qh), h2, off2, hi - lo, cnt, 0)
    key = int(be.argmax(cnt, hi - lo, lo).item())
    c, g = parallel.unpack_key(key)
    want = [oracle.intersection_size(qh, d)[0] for d in dbh[lo:hi]]
    assert c == max(want) and g == lo + int(np.argmax(want))
