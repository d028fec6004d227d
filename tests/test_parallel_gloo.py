"""world_size-2 test of the multi-GPU drivers over gloo on CPU.

sourmash_amd.parallel's distributed control flow (tile dealing, the single all-gather of the
compare path, the candidate exchange of the gather path: one all-gather per batch of rounds) is backend-agnostic.  The
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

    def sort_unique(self, keys):
        return torch.from_numpy(np.unique(keys.numpy().view(np.uint64)).view(np.int64))

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
        assert op == 0
        for d in range(ndb):
            counters[d] = int(oracle.intersection_size(q, h[off[d]:off[d + 1]])[0])

    def gather_state(self, query, nq, hashes, offsets, ndb, index_base):
        return OracleGatherState(self._u64(query, nq), self._u64(hashes), self._u64(offsets), ndb, index_base)


class FlakySharedBackend(OracleBackend):
    """OracleBackend whose indexes claim they can run the resident loop and whose shared exchange goes wrong in a chosen way on a
    chosen rank: gather_distributed's default branch must end with every rank on the record protocol -- agreed, no rank left in
    a collective of its own (parallel.py:435)."""

    def __init__(self, rank, fail_open_on=None, fail_launch_on=None, fail_results_on=None):
        self.rank, self.fail_open_on, self.fail_launch_on, self.fail_results_on = rank, fail_open_on, fail_launch_on, fail_results_on
        self.opened = 0

    def open_exchange(self, world, rank, rowcap, group=None):
        import types

        def make(name, create):
            if self.fail_open_on == rank:
                raise OSError("no shared memory on this rank")
            return types.SimpleNamespace(world=world, rowcap=rowcap, name=name)
        self.opened += 1
        return parallel.open_shared_exchange(self, make, world, rank, rowcap, group)

    def gather_state(self, query, nq, hashes, offsets, ndb, index_base):
        st = OracleGatherState(self._u64(query, nq), self._u64(hashes), self._u64(offsets), ndb, index_base)
        be = self
        st.loop_eligible = lambda n_wg=0: True

        def launch_shared(xchg, rank, run_id, n_wg=0):
            if be.fail_launch_on == rank:
                raise RuntimeError("a row of this shard is longer than the exchange's slots")
            return True
        st.launch_shared = launch_shared
        inner = st.results

        def results():
            if st.done or st.out:                         # (the record protocol's own read-back)
                return inner()
            raise RuntimeError("gather loop: a workgroup or rank waited too long for its peers")   # the shared loop never completes here
        st.results = results
        return st


class OracleGatherState:
    """CPU stand-in for the native per-rank gather state (same steps, plain numpy sets): counters by direct
    intersection with the uncovered query, i.e. the textbook CounterGather rather than the postings walk."""

    def __init__(self, q, h, off, ndb, index_base):
        self.rows = [h[off[d]:off[d + 1]].copy() for d in range(ndb)]
        self.base, self.uncovered = index_base, set(int(x) for x in q)
        self.cnt = np.array([len(self.uncovered.intersection(int(x) for x in r)) for r in self.rows], dtype=np.int64)
        self.done, self.pending, self.out, self.key, self.acc = False, False, [], 0, 0
        self.cands, self.bound, self.needx = [], 0, True

    def begin(self, thr, max_rounds):
        self.thr, self.maxr, self.done, self.pending, self.out = thr, max(max_rounds, 1), False, False, []

    def longest_row(self):
        return max([len(r) for r in self.rows] + [0])

    def _record(self):
        if self.pending:
            self.out.append((parallel.unpack_key(self.key)[1], self.acc))
            self.pending = False
            if len(self.out) >= self.maxr:
                self.done = True

    def export_topk(self, records, k):
        records.zero_()
        if self.done:
            return
        keys = sorted((parallel.pack_key(int(c), self.base + d) for d, c in enumerate(self.cnt) if c), reverse=True)
        bound = keys[k] if len(keys) > k else 0
        for slot, key in enumerate(keys[:k]):
            row = self.rows[parallel.unpack_key(key)[1] - self.base]
            records[slot, 0], records[slot, 1], records[slot, 2] = key, bound, len(row)
            records[slot, 3:3 + len(row)] = torch.from_numpy(row.view(np.int64).copy())

    def load_candidates(self, records, n):
        self.cands, self.bound, self.needx = [], 0, False
        for c in range(n):
            key = int(records[c, 0])
            if not key:
                self.cands.append(None)
                continue
            row = set(int(x) for x in records[c, 3:3 + int(records[c, 2])].numpy().view(np.uint64))
            self.cands.append([key, key >> 32, row])
            self.bound = max(self.bound, int(records[c, 1]))

    def replay(self, rounds):
        for _ in range(rounds):
            if self.done:
                return
            self._record()
            if self.done or self.needx:
                continue
            live = [((c[1] << 32) | (c[0] & 0xFFFFFFFF), c) for c in self.cands if c and c[1]]
            best, winner = max(live, key=lambda t: t[0]) if live else (0, None)
            if best < self.bound:
                self.needx = True
                continue
            self.key = best
            if best == 0 or not self.uncovered or len(self.uncovered) < self.thr or (best >> 32) < self.thr:
                self.done = True
                return
            self.pending = True
            isect = self.uncovered & winner[2]
            self.acc = len(isect)
            self.uncovered -= isect
            for d, r in enumerate(self.rows):
                if self.cnt[d]:
                    self.cnt[d] -= len(isect.intersection(int(x) for x in r))
            for c in self.cands:
                if c:
                    c[1] -= len(isect & c[2])

    def poll(self):
        return len(self.out), self.done

    def results(self):
        return list(self.out)

    def run(self):
        raise AssertionError("single-rank fused loop is a device feature; the gloo tests run the exchange protocol")


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
        # the same exchange cut into bands that travel while the next band's tiles are computed (asynchronous all-gathers)
        parallel.COMPARE_BAND_MIN_SLOTS = 0
        t_banded = {}
        common_b, jac_b = parallel.compare_all_pairs_distributed(h, off, len(sk), be, timing=t_banded)
        parallel.COMPARE_BAND_MIN_SLOTS = 8
        ok_cmp = ok_cmp and np.array_equal(common_b.numpy().view(np.uint32), wc) and np.array_equal(jac_b.numpy().view(np.uint64), wj.view(np.uint64))
        ok_cmp = ok_cmp and t_banded.get("exchange_pieces") == 2
        # the matrices on rank 0 only (what the reference's caller gets): a gather instead of the all-gather, mirror + Jaccard
        # on the root alone, None elsewhere -- plain and in bands
        for min_slots in (8, 0):
            parallel.COMPARE_BAND_MIN_SLOTS = min_slots
            common_r, jac_r = parallel.compare_all_pairs_distributed(h, off, len(sk), be, result_on="root")
            parallel.COMPARE_BAND_MIN_SLOTS = 8
            if rank == 0:
                ok_cmp = ok_cmp and np.array_equal(common_r.numpy().view(np.uint32), wc) and np.array_equal(jac_r.numpy().view(np.uint64), wj.view(np.uint64))
            else:
                ok_cmp = ok_cmp and common_r is None and jac_r is None
        # counts travel as 16-bit words unless a sketch could share 65,536 hashes or more: two sketches of 70,000 that share
        # 66,000 take the 32-bit form (and a wrapped count would show)
        wide = [np.arange(1, 70_001, dtype=np.uint64) * np.uint64(977), np.arange(4001, 74_001, dtype=np.uint64) * np.uint64(977)]
        wide += [s for s in sk[:19]]
        h2, off2 = _csr(wide)
        common2, _ = parallel.compare_all_pairs_distributed(h2, off2, len(wide), be, want_jaccard=False)
        wc2, _ = oracle.compare_all_pairs(*oracle.make_csr(wide))
        ok_cmp = ok_cmp and int(wc2[0, 1]) == 66_000 and np.array_equal(common2.numpy().view(np.uint32), wc2)
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
        # ---- the default (shared-exchange) branch going wrong on ONE rank, at each of its steps: both ranks end on the record
        # protocol with the right answer, nobody hangs in a collective the other never joins ----
        want0 = oracle.gather(qh, fh, foff, threshold_bp=0, scaled=1000)
        for kw in ({"fail_open_on": 1}, {"fail_open_on": 0}, {"fail_launch_on": 1}, {"fail_launch_on": 0}, {}):
            fb = FlakySharedBackend(rank, **kw)
            stats = {}
            got = parallel.gather_distributed(q, len(qh), sh, soff, hi - lo, lo, 0, 1000, fb, stats=stats)
            ok_g = ok_g and got == want0 and fb.opened == 1 and stats.get("shared_exchange", "").startswith("failed")
        # ---- opening the exchange: made once, reused, re-made when it must grow or its run tags would wrap ----
        import types
        owner, made = types.SimpleNamespace(), []

        def make(name, create):
            made.append((name, create))
            return types.SimpleNamespace(world=world, rowcap=100 if len(made) == 1 else 500, name=name)
        x1, run1 = parallel.open_shared_exchange(owner, make, world, rank, 100)
        x2, run2 = parallel.open_shared_exchange(owner, make, world, rank, 80)
        ok_x = x1 is x2 and (run1, run2) == (1, 2) and made == [(x1.name, rank == 0)] and x1.name.startswith("/smg_gx_")
        x3, run3 = parallel.open_shared_exchange(owner, make, world, rank, 300)          # too small now: a new segment
        ok_x = ok_x and x3 is not x1 and run3 == 1 and x3.name != x1.name and len(made) == 2
        owner._xchg_runs = parallel.EXCHANGE_RUNS_MAX                                   # 12 bits of run number in the tags
        x4, run4 = parallel.open_shared_exchange(owner, make, world, rank, 300)
        ok_x = ok_x and x4 is not x3 and run4 == 1 and len(made) == 3
        names = [None, None]
        dist.all_gather_object(names, x4.name)
        ok_x = ok_x and names[0] == names[1]                                            # the creator's name reached the other rank
        ok_g = ok_g and ok_x
        # ---- search / prefetch: one overlap pass per shard, one all-gather of (count, size) pairs ----
        shared, sizes = parallel.overlaps_distributed(q, len(qh), sh, soff, hi - lo, lo, be)
        want_shared = np.array([oracle.intersection_size(qh, d)[0] for d in dbh], dtype=np.uint64)
        ok_o = np.array_equal(shared, want_shared) and np.array_equal(sizes, np.array([len(d) for d in dbh], dtype=np.uint64))
        pf = parallel.prefetch_distributed(q, len(qh), sh, soff, hi - lo, lo, 30_000, 1000, be)
        ok_o = ok_o and pf == [(i, int(c)) for i, c in enumerate(want_shared) if c >= 30 and c > 0] and len(pf) > 3
        for mode in ({"do_containment": True}, {"do_max_containment": True}, {}):
            hits = parallel.search_distributed(q, len(qh), sh, soff, hi - lo, lo, be, threshold=0.01, **mode)
            ref = []
            for i, d in enumerate(dbh):
                c, u = oracle.intersection_size(qh, d)
                sc = (c / len(qh) if "do_containment" in mode else c / min(len(d), len(qh)) if "do_max_containment" in mode
                      else c / u) if c else 0.0
                if sc >= 0.01 and sc > 0:
                    ref.append((sc, i))
            ref.sort(key=lambda t: (-t[0], t[1]))
            ok_o = ok_o and hits == ref and len(hits) > 3
        best = parallel.search_distributed(q, len(qh), sh, soff, hi - lo, lo, be, best_only=True)
        ok_o = ok_o and best == ref[:1]                             # equal scores (rows 3, 7 and 40 are identical): the lowest index
        # ---- sketch: records dealt to the ranks, one all-gather of the kept hashes, union = sketch of everything ----
        seq = oracle.synth_dna(0, 600_000, seed=42, record_len=50_000)           # 11 records + separators
        bounds = [0, 6 * 50_001, len(seq)]                                       # whole records per rank
        mine = oracle.sketch_dna_bulk(seq[bounds[rank]:bounds[rank + 1]], 31, scaled=100)
        union = parallel.allgather_union(torch.from_numpy(mine.view(np.int64).copy()), backend=be)
        whole = oracle.sketch_dna_bulk(seq, 31, scaled=100)
        ok_s = np.array_equal(union.numpy().view(np.uint64), whole)
        big = torch.tensor([5, -3, -1, 7], dtype=torch.int64) if rank == 0 else torch.tensor([-2, 5], dtype=torch.int64)
        ordered = parallel.allgather_union(big, backend=be).numpy().view(np.uint64)          # u64 order with the top bit set (scaled = 1)
        ok_s = ok_s and list(ordered) == sorted({5, 7, 2**64 - 3, 2**64 - 1, 2**64 - 2})
        ret[rank] = (bool(ok_cmp), bool(ok_g), len(res[0]), bool(ok_s), bool(ok_o))
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
        ok_cmp, ok_g, rounds, ok_s, ok_o = ret[r]
        assert ok_o, f"rank {r}: distributed overlaps / prefetch / search differ from the oracle"
        assert ok_s, f"rank {r}: all-gathered sketch union differs from the oracle's sketch of the whole input"
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
