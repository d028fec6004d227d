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


def test_gather_exchange_protocol_and_emulated_shards(be):
    """The sharded protocol (export top-K -> all-gather -> load -> replay rounds) with the real kernels: once on one
    state (stepwise=True), once on three states sharing this GPU whose 'all-gather' is a torch.cat."""
    import torch
    from sourmash_amd import device as smd, parallel
    from sourmash_amd.synth import synth_gather
    qh, dbh = synth_gather(n_query=50_000, n_db=900, db_size=700)
    dbh[400] = dbh[2].copy()                                     # tie across shards: lowest global index wins
    dbh[17] = np.zeros(0, dtype=np.uint64)                       # an empty sketch
    dbh[18] = np.array([5], dtype=np.uint64)                     # nothing in common with the query
    dbh[700] = np.unique(np.concatenate(dbh[600:640]))           # a long row (sizes the records)
    q = torch.from_numpy(qh.view(np.int64).copy()).cuda()
    fh, foff = oracle.make_csr(dbh)
    h, off = smd.pack_csr(dbh)
    for thr_bp in (0, 30_000):
        want = oracle.gather(qh, fh, foff, threshold_bp=thr_bp, scaled=1000)
        stats = {}
        assert parallel.gather_distributed(q, len(qh), h, off, len(dbh), 0, thr_bp, 1000, be, stepwise=True, stats=stats) == want
        assert stats["exchanges"] * stats["rounds_per_exchange"] >= len(want)
        assert parallel.gather_distributed(q, len(qh), h, off, len(dbh), 0, thr_bp, 1000, be, max_rounds=7) == want[:7]
        assert parallel.gather_distributed(q, len(qh), h, off, len(dbh), 0, thr_bp, 1000, be, max_rounds=7, stepwise=True) == want[:7]
    # three shards, one GPU
    bounds = [(0, 250), (250, 610), (610, 900)]
    shards = [smd.pack_csr(dbh[lo:hi]) for lo, hi in bounds]
    states = [be.gather_state(q, len(qh), sh, so, hi - lo, lo) for (sh, so), (lo, hi) in zip(shards, bounds)]
    want_counts = np.array([oracle.intersection_size(qh, d)[0] for d in dbh], dtype=np.uint64)
    assert np.array_equal(np.concatenate([s.counters() for s in states]), want_counts)
    assert max(s.longest_row() for s in states) == max(len(d) for d in dbh)
    stride = parallel.CAND_HEAD + max(len(d) for d in dbh)
    for thr_bp, k, rounds in ((30_000, 5, 9), (0, 16, 16), (30_000, 1, 3)):
        want = oracle.gather(qh, fh, foff, threshold_bp=thr_bp, scaled=1000)
        states = [be.gather_state(q, len(qh), sh, so, hi - lo, lo) for (sh, so), (lo, hi) in zip(shards, bounds)]
        for s in states:
            s.begin(int(np.ceil(thr_bp / 1000)), len(dbh))
        recs = [be.zeros((k, stride), torch.int64) for _ in states]
        done, exchanges = False, 0
        while not done:
            for _ in range(4):
                for s, r in zip(states, recs):
                    s.export_topk(r, k)
                everyone = torch.cat(recs)                               # the all-gather
                for s in states:
                    s.load_candidates(everyone, len(states) * k)
                    s.replay(rounds)
                exchanges += 1
            polls = [s.poll() for s in states]
            assert len({p for p in polls}) == 1                          # replicated, deterministic state
            done = polls[0][1]
        for s in states:
            assert s.results() == want, (thr_bp, k)
        if k > 1:
            assert exchanges < len(want)                                 # fewer exchanges than rounds
        # the counters every shard ends with are the true remaining overlaps
        covered = set()
        for gidx, _ in want:
            covered.update(int(x) for x in dbh[gidx])
        left = np.array([x for x in qh if int(x) not in covered], dtype=np.uint64)
        want_left = np.array([oracle.intersection_size(left, d)[0] for d in dbh], dtype=np.uint64)
        assert np.array_equal(np.concatenate([s.counters() for s in states]), want_left)


def test_gather_native_loops_agree(be, monkeypatch):
    "smgpu_gather_run: the scan loop (arg-max over all counters every round) and the replay loop give the oracle's list"
    import subprocess, sys, json, os
    from conftest import ROOT
    code = (
        "import sys, json, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from sourmash_amd import device as smd, parallel\n"
        "from sourmash_amd.synth import synth_gather\n"
        "qh, dbh = synth_gather(n_query=60_000, n_db=1500, db_size=600)\n"
        "dbh[11] = dbh[5].copy()\n"
        "be = parallel.DeviceBackend(); h, off = smd.pack_csr(dbh)\n"
        "q = torch.from_numpy(qh.view(np.int64).copy()).cuda()\n"
        "print(json.dumps(parallel.gather_distributed(q, len(qh), h, off, len(dbh), 0, 20_000, 1000, be)))\n" % ROOT)
    outs = {}
    for mode in ("scan", "replay"):
        env = dict(os.environ, SMG_GATHER_LOOP=mode)
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs[mode] = [tuple(x) for x in json.loads(p.stdout.strip().splitlines()[-1])]
    from sourmash_amd.synth import synth_gather
    qh, dbh = synth_gather(n_query=60_000, n_db=1500, db_size=600)
    dbh[11] = dbh[5].copy()
    want = oracle.gather(qh, *oracle.make_csr(dbh), threshold_bp=20_000, scaled=1000, nthreads=8)
    assert outs["scan"] == want and outs["replay"] == want


def test_search_and_prefetch_over_emulated_shards(be):
    """the database sharded by dataset, one overlap pass per shard (the streaming kernel for this query
    size), the (count, size) pairs assembled in global order -- the all-gather replaced by the list of per-shard blocks --
    then the single-GPU scoring.  World 1 end to end through search_distributed / prefetch_distributed as well."""
    import torch
    from sourmash_amd import device as smd, parallel
    from sourmash_amd.index import rank_search_hits, prefetch_rows
    from sourmash_amd.synth import synth_gather
    qh, dbh = synth_gather(n_query=140_000, n_db=4500, db_size=300)
    dbh[9] = np.zeros(0, dtype=np.uint64)
    dbh[2000] = dbh[4].copy()                                    # identical rows in different shards: the lowest index ranks first
    dbh[4400] = qh[::5].copy()                                   # fully contained in the query
    q = torch.from_numpy(qh.view(np.int64).copy()).cuda()
    want_shared = np.array([oracle.intersection_size(qh, d)[0] for d in dbh], dtype=np.uint64)
    want_sizes = np.array([len(d) for d in dbh], dtype=np.uint64)
    h, off = smd.pack_csr(dbh)
    shared, sizes = parallel.overlaps_distributed(q, len(qh), h, off, len(dbh), 0, be)
    assert np.array_equal(shared, want_shared) and np.array_equal(sizes, want_sizes)
    for world in (2, 3, 8):
        cuts = [len(dbh) * r // world for r in range(world + 1)]
        pieces, rows = [], []
        for lo, hi in zip(cuts, cuts[1:]):
            sh, so = smd.pack_csr(dbh[lo:hi])
            pieces.append(parallel.local_overlaps(q, len(qh), sh, so, hi - lo, be))
            rows.append(hi - lo)
        full = parallel.assemble_overlaps(pieces, rows).cpu().numpy().view(np.uint64)
        assert np.array_equal(full[:, 0], want_shared) and np.array_equal(full[:, 1], want_sizes), world
    # scoring on the assembled vectors = the reference's per-dataset scores (search.py:88-160), ties by lowest index
    for mode, thr in (({}, 0.001), ({"do_containment": True}, 0.00105), ({"do_max_containment": True}, 0.5)):
        hits = parallel.search_distributed(q, len(qh), h, off, len(dbh), 0, be, threshold=thr, **mode)
        ref = []
        for i, d in enumerate(dbh):
            c, u = oracle.intersection_size(qh, d)
            if not c:
                continue
            sc = c / len(qh) if "do_containment" in mode else c / min(len(d), len(qh)) if "do_max_containment" in mode else c / u
            if sc >= thr:
                ref.append((sc, i))
        ref.sort(key=lambda t: (-t[0], t[1]))
        assert hits == ref and 1 < len(ref) < len(dbh), (mode, len(ref))
    assert parallel.search_distributed(q, len(qh), h, off, len(dbh), 0, be, do_max_containment=True, best_only=True)[0][1] == 4400
    pf = parallel.prefetch_distributed(q, len(qh), h, off, len(dbh), 0, 100_000, 1000, be)
    assert pf == [(i, int(c)) for i, c in enumerate(want_shared) if c >= 100] and len(pf) > 2
    assert pf == prefetch_rows(want_shared, 100_000, 1000)
    assert rank_search_hits(want_shared, want_sizes, len(qh), best_only=True) == \
        parallel.search_distributed(q, len(qh), h, off, len(dbh), 0, be, best_only=True)


def test_gather_shared_exchange_emulated_ranks():
    """The multi-rank form of the resident loop (csrc/gather.hip: gather_launch_loop with a GatherShared): every rank's loop kernel
    runs all rounds, the local winners meet in host-visible memory each round, the winner's query positions travel through
    the same memory.  One process drives 2 / 3 / 4 ranks here -- a kernel per rank on its own stream with its share of the CUs,
    a private pinned exchange -- so the protocol itself (tags, double buffering, staging of a remote winner's row, replicated
    stop rules, ties across ranks) runs for real; across processes only the memory is different (POSIX shared memory).
    Own interpreter: the loops of all ranks must run AT THE SAME TIME, so every stream needs its own hardware queue
    (GPU_MAX_HW_QUEUES, read when HIP starts; with the default of 4, two of the streams can share a queue and the second
    loop would wait for the first to end -- a limit of driving several ranks from one process, not of the protocol)."""
    import os, subprocess, sys
    from conftest import ROOT
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_parallel as t\nt._shared_exchange_emulated()\nprint('ok')\n" % (ROOT, os.path.join(ROOT, "tests")))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, GPU_MAX_HW_QUEUES="8", SMG_GATHER_BUILD="ranges"))
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), (p.stdout[-1500:], p.stderr[-1500:])


def _shared_exchange_emulated():
    import torch
    from sourmash_amd import device as smd, parallel
    from sourmash_amd.synth import synth_gather
    be = parallel.DeviceBackend()
    qh, dbh = synth_gather(n_query=60_000, n_db=2400, db_size=700)
    dbh[1700] = dbh[3].copy()                                    # a tie across ranks: the lowest global index wins
    dbh[17] = np.zeros(0, dtype=np.uint64)
    dbh[2000] = np.unique(np.concatenate(dbh[1000:1030]))        # a long row (sizes the exchange slots) on a later rank
    q = torch.from_numpy(qh.view(np.int64).copy()).cuda()
    fh, foff = oracle.make_csr(dbh)
    for world in (2, 3, 4):
        cuts = [len(dbh) * r // world for r in range(world + 1)]
        shards = []
        for lo, hi in zip(cuts, cuts[1:]):
            h, off = smd.pack_csr(dbh[lo:hi])
            shards.append((h, off, hi - lo, lo))
        for thr_bp, cap in ((0, None), (30_000, None), (0, 9)):
            want = oracle.gather(qh, fh, foff, threshold_bp=thr_bp, scaled=1000, nthreads=8)
            if cap:
                want = want[:cap]
            got = parallel.gather_emulated_ranks(q, len(qh), shards, thr_bp, 1000, be, max_rounds=cap)
            assert got is not None, "the resident loop should apply to these shards"
            for r, picks in enumerate(got):
                assert picks == want, (world, thr_bp, cap, r, picks[:3], want[:3], len(picks), len(want))


def test_gather_distributed_falls_back_when_the_shared_exchange_is_unavailable(be, monkeypatch):
    """gather_distributed with the collectives' code path taken on one rank (force_collectives): the shared-memory exchange
    by default; when opening it fails -- on any rank: the ranks agree before anybody acts -- a fresh index and the record
    protocol take over.  Both give the oracle's picks."""
    import torch
    from sourmash_amd import device as smd, parallel
    from sourmash_amd.synth import synth_gather
    import os
    import torch.distributed as dist
    monkeypatch.setenv("SMG_GATHER_BUILD", "ranges")
    qh, dbh = synth_gather(n_query=50_000, n_db=1300, db_size=600)
    q = torch.from_numpy(qh.view(np.int64).copy()).cuda()
    h, off = smd.pack_csr(dbh)
    want = oracle.gather(qh, *oracle.make_csr(dbh), threshold_bp=20_000, scaled=1000, nthreads=8)
    dist.init_process_group("nccl", rank=0, world_size=1, init_method="tcp://127.0.0.1:%d" % (29600 + os.getpid() % 300),
                            device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        for kind, word in (("device", "hipIpc"), ("shared", "shared host memory")):
            monkeypatch.setenv("SMG_GATHER_EXCHANGE", kind)
            stats = {}
            assert parallel.gather_distributed(q, len(qh), h, off, len(dbh), 0, 20_000, 1000, be, force_collectives=True, stats=stats) == want
            assert word in stats.get("protocol", ""), (kind, stats)
        monkeypatch.delenv("SMG_GATHER_EXCHANGE")
        def broken(*a, **k):
            raise OSError("no shared memory here")
        monkeypatch.setattr(be, "open_exchange", broken)
        stats = {}
        assert parallel.gather_distributed(q, len(qh), h, off, len(dbh), 0, 20_000, 1000, be, force_collectives=True, stats=stats) == want
        assert stats.get("shared_exchange", "").startswith("failed") and stats["exchanges"] > 0, stats
    finally:
        dist.destroy_process_group()
