"""world_size-2 test of the multi-GPU drivers over gloo on CPU.

sourmash_amd.parallel's distributed control flow (tile dealing, the single all-gather of the
compare path, the per-round MAX all-reduce + broadcast of the gather path) is backend-agnostic.  The
product backend launches HIP kernels; here the test injects a CPU backend built on the ORACLE (test
infrastructure) so the collectives and the partition logic run for real with two processes."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from sourmash_amd import parallel  # noqa: E402
from sourmash_amd.synth import synth_gather, synth_sketches  # noqa: E402


class OracleBackend:
    "CPU stand-in for DeviceBackend: same interface, numpy/oracle arithmetic (tests only)."
    torch = torch

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype)

    def empty(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype)

    @staticmethod
    def _u64(t, n=None):
        a = t.numpy().view(np.uint64)
        return a if n is None else a[:n]

    def compare_tiles(self, hashes, offsets, n, first, stride, count):
        h, off = self._u64(hashes), self._u64(offsets)
        out = np.zeros((count * parallel.TILE, n), dtype=np.uint32)
        for t in range(count):
            rb = first + t * stride
            for r in range(parallel.TILE):
                i = rb * parallel.TILE + r
                if i >= n:
                    break
                for j in range(rb * parallel.TILE, n):        # tiles on/above the diagonal
                    if j < i:
                        continue
                    c, _ = oracle.intersection_size(h[off[i]:off[i + 1]], h[off[j]:off[j + 1]])
                    out[t * parallel.TILE + r, j] = c
        return torch.from_numpy(out.view(np.int32))

    def symmetrize(self, common, n):
        a = common.numpy()
        iu = np.triu_indices(n, 1)
        a.T[iu] = a[iu]

    def jaccard(self, common, offsets, n):
        c = common.numpy().view(np.uint32).astype(np.float64)
        sizes = np.diff(self._u64(offsets)).astype(np.float64)
        uni = sizes[:, None] + sizes[None, :] - c
        j = c / np.maximum(uni, 1.0)
        np.fill_diagonal(j, 1.0)
        return torch.from_numpy(j)

    def overlaps(self, query, nq, hashes, offsets, ndb, counters, op):
        q, h, off = self._u64(query, nq), self._u64(hashes), self._u64(offsets)
        cnt = counters.numpy()
        for d in range(ndb):
            if op == 1 and cnt[d] == 0:
                continue
            c, _ = oracle.intersection_size(q, h[off[d]:off[d + 1]])
            cnt[d] = c if op == 0 else max(0, cnt[d] - c)

    def argmax(self, counters, ndb, index_base):
        cnt = counters.numpy()[:ndb]
        best = 0
        for d in range(ndb):
            if cnt[d]:
                best = max(best, parallel.pack_key(cnt[d], index_base + d))
        return torch.tensor([best], dtype=torch.int64)

    def select(self, a, na, b, nb, invert):
        x, y = self._u64(a, na), self._u64(b, nb)
        keep = ~np.isin(x, y) if invert else np.isin(x, y)
        out = x[keep]
        t = torch.zeros(max(na, 1), dtype=torch.int64)
        t[:len(out)] = torch.from_numpy(out.view(np.int64).copy())
        return t, len(out)


def _csr(sketches):
    h, off = oracle.make_csr(sketches)
    if h.size == 0:
        h = np.zeros(2, dtype=np.uint64)
    return torch.from_numpy(h.view(np.int64).copy()), torch.from_numpy(off.view(np.int64).copy())


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        be = OracleBackend()
        # ---- compare: 45 sketches -> 3 tiles dealt over 2 ranks, one all-gather ----
        sk = synth_sketches(45, pool_size=1500)
        h, off = _csr(sk)
        common, jac = parallel.compare_all_pairs_distributed(h, off, len(sk), be)
        wc, wj = oracle.compare_all_pairs(*oracle.make_csr(sk))
        ok_cmp = np.array_equal(common.numpy().view(np.uint32), wc) and \
            np.array_equal(jac.numpy().view(np.uint64), wj.view(np.uint64))
        # ---- gather: database sharded by dataset, replicated query ----
        qh, dbh = synth_gather(n_query=4000, n_db=61, db_size=120)
        dbh[7] = dbh[3].copy()                                  # a tie across... the same shard
        dbh[40] = dbh[3].copy()                                 # ...and across shards: lowest index must win
        lo, hi = (0, 30) if rank == 0 else (30, 61)
        sh, soff = _csr(dbh[lo:hi])
        q = torch.from_numpy(qh.view(np.int64).copy())
        res = {}
        for thr in (0, 20_000):
            res[thr] = parallel.gather_distributed(q, len(qh), sh, soff, hi - lo, lo, thr, 1000, be)
        fh, foff = oracle.make_csr(dbh)
        ok_g = all(res[thr] == oracle.gather(qh, fh, foff, threshold_bp=thr, scaled=1000) for thr in res)
        ret[rank] = (bool(ok_cmp), bool(ok_g), len(res[0]))
    finally:
        dist.destroy_process_group()


def test_two_rank_compare_and_gather_over_gloo():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert set(ret.keys()) == {0, 1}
    for r in range(world):
        ok_cmp, ok_g, rounds = ret[r]
        assert ok_cmp, f"rank {r}: distributed compare differs from the oracle"
        assert ok_g, f"rank {r}: distributed gather differs from the oracle"
        assert rounds > 5


def test_tile_dealing_covers_every_tile_once():
    for n in (1, 15, 16, 17, 100, 1000, 10_000):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                first, stride, count = parallel.tiles_for_rank(n, world, r)
                seen += [first + i * stride for i in range(count)]
            assert sorted(seen) == list(range((n + 15) // 16)), (n, world)


def test_packed_key_orders_by_count_then_lowest_index():
    k = parallel.pack_key
    assert k(5, 10) > k(4, 0) and k(5, 3) > k(5, 4) and k(1, 0) > 0
    assert parallel.unpack_key(k(123456, 99999)) == (123456, 99999)
    assert k(2**31 - 1, 0) < 2**63
