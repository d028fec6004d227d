"""One rank of tests/test_gpu_two_processes.py: a real process with its own HIP context on the (one) GPU, a gloo process group
with its peer, DeviceBackend -- and the multi-GPU drivers of sourmash_amd.parallel exactly as `bench.py --gpus N` calls them.
Prints one line `RESULT <json>`; the oracle (test infrastructure) is the checker.

usage: python tests/two_process_worker.py RANK WORLD PORT"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SMG_GATHER_BUILD"] = "ranges"            # the staged range builder also for a small database: the resident loop's index
    import numpy as np
    import torch
    import torch.distributed as dist
    import oracle
    import sourmash_amd as sm
    from sourmash_amd import device as smd, parallel
    from sourmash_amd.synth import synth_gather, synth_sketches
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {"rank": rank}
    try:
        be = parallel.DeviceBackend()
        # ---- gather: database sharded by dataset, query replicated (index/__init__.py:856-909, search.py:755-779) ----
        qh, dbh = synth_gather(n_query=60_000, n_db=2400, db_size=700)
        dbh[1700] = dbh[3].copy()                            # equal counters on different ranks: the lowest global index wins
        dbh[9] = dbh[3].copy()                               # ... and on the same rank
        cuts = [len(dbh) * r // world for r in range(world + 1)]
        lo, hi = cuts[rank], cuts[rank + 1]
        h, off = smd.pack_csr(dbh[lo:hi])
        q = torch.from_numpy(qh.view(np.int64).copy()).cuda()
        fh, foff = oracle.make_csr(dbh)
        for thr in (0, 30_000):
            want = oracle.gather(qh, fh, foff, threshold_bp=thr, scaled=1000, nthreads=4)
            for mode in ("device", "shared", "records"):
                os.environ["SMG_GATHER_EXCHANGE"] = mode
                tries, proto, ok, fallbacks = 0, "", True, 0
                while tries < (4 if mode != "records" else 1):
                    tries += 1
                    stats = {}
                    got = parallel.gather_distributed(q, len(qh), h, off, hi - lo, lo, thr, 1000, be, stats=stats)
                    ok = ok and got == want
                    proto = stats.get("protocol") or stats.get("shared_exchange") or "candidate records"
                    fallbacks += 1 if "shared_exchange" in stats else 0
                    if mode == "records" or "resident loop kernels" in proto:
                        break
                out["gather_%s_thr%d" % (mode, thr)] = {"ok": bool(ok), "rounds": len(want), "protocol": proto, "tries": tries,
                                                        "fell_back": fallbacks}
        os.environ.pop("SMG_GATHER_EXCHANGE", None)
        # ---- search / prefetch: one overlap pass per shard + one all-gather of (count, size) pairs (linear.rs:52-113) ----
        shared, sizes = parallel.overlaps_distributed(q, len(qh), h, off, hi - lo, lo, be)
        want_shared = np.array([oracle.intersection_size(qh, d)[0] for d in dbh], dtype=np.uint64)
        ok_o = np.array_equal(shared, want_shared) and np.array_equal(sizes, np.array([len(d) for d in dbh], dtype=np.uint64))
        pf = parallel.prefetch_distributed(q, len(qh), h, off, hi - lo, lo, 300_000, 1000, be)
        ok_o = ok_o and pf == [(i, int(c)) for i, c in enumerate(want_shared) if c >= 300] and len(pf) > 3
        best = parallel.search_distributed(q, len(qh), h, off, hi - lo, lo, be, best_only=True, do_containment=True)
        top = max(int(c) for c in want_shared)
        ok_o = ok_o and best == [(top / len(qh), int(np.argmax(want_shared == top)))]
        out["overlaps"] = bool(ok_o)
        # ---- compare: CSR replicated, 16-row tiles dealt round-robin, ONE all-gather (compare.py:14-64) ----
        sk = synth_sketches(333, pool_size=9000)
        ch, coff = smd.pack_csr(sk)
        common, jac = parallel.compare_all_pairs_distributed(ch, coff, len(sk), be)
        torch.cuda.synchronize()
        wc, wj = oracle.compare_all_pairs(*oracle.make_csr(sk), nthreads=4)
        out["compare"] = bool(np.array_equal(common.cpu().numpy().view(np.uint32), wc) and
                              np.array_equal(jac.cpu().numpy().view(np.uint64), wj.view(np.uint64)))
        # ---- sketch: records dealt to the ranks, one all-gather of the kept hashes (signature.rs:38-58, minhash.rs:432-516) ----
        seq = oracle.synth_dna(0, 900_000, seed=42, record_len=50_000).tobytes()
        n_rec = len(seq) // 50_001
        bounds = [r * (n_rec // world) * 50_001 for r in range(world)] + [len(seq)]
        mh = sm.MinHash(0, 31, scaled=100)
        mh.add_sequence_buffer(seq[bounds[rank]:bounds[rank + 1]])
        mine = torch.from_numpy(mh._mins_array().view(np.int64).copy()).cuda()
        union = parallel.allgather_union(mine)
        out["sketch_union"] = bool(np.array_equal(union.cpu().numpy().view(np.uint64), oracle.sketch_dna_bulk(seq, 31, scaled=100)))
    except Exception as e:                                       # noqa: BLE001 -- the test prints it
        import traceback
        out["error"] = traceback.format_exc()[-1500:]
    finally:
        print("RESULT " + json.dumps(out), flush=True)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
