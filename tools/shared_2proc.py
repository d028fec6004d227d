"""Two PROCESSES on one GPU, one database shard each, running the shared-exchange gather through POSIX shared memory registered
with HIP in both (what ranks on different GPUs of a node do, minus xGMI): python tools/shared_2proc.py"""
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, barrier, name, out):
    import numpy as np
    import torch
    from sourmash_amd import device as smd, parallel
    from sourmash_amd.synth import synth_gather
    os.environ["SMG_GATHER_BUILD"] = "ranges"
    be = parallel.DeviceBackend()
    qh, dbh = synth_gather(n_query=60_000, n_db=2400, db_size=700)
    dbh[1700] = dbh[3].copy()
    cuts = [len(dbh) * r // world for r in range(world + 1)]
    lo, hi = cuts[rank], cuts[rank + 1]
    h, off = smd.pack_csr(dbh[lo:hi])
    q = torch.from_numpy(qh.view(np.int64).copy()).cuda()
    st = be.gather_state(q, len(qh), h, off, hi - lo, lo)
    rowcap = 4096
    if rank == 0:
        xchg = parallel.GatherExchange(be.lib, be.rustcall, world, rowcap, name, create=True)
    barrier.wait()
    if rank != 0:
        xchg = parallel.GatherExchange(be.lib, be.rustcall, world, rowcap, name, create=False)
    st.begin(0, len(dbh))
    be.rustcall(be.lib.smgpu_gather_loop_reserve, st._ptr, 64, rowcap, be._s())
    torch.cuda.synchronize()
    barrier.wait()
    try:
        ok = st.launch_shared(xchg, rank, 1, 64)
        res = st.results()
        out.put((rank, ok, res))
    except Exception as e:
        out.put((rank, False, str(e)[-90:]))
    barrier.wait()


def main():
    import numpy as np
    import oracle
    from sourmash_amd.synth import synth_gather
    world = 2
    ctx = mp.get_context("spawn")
    barrier, out = ctx.Barrier(world), ctx.Queue()
    name = "/smg_gx_test_%d" % os.getpid()
    procs = [ctx.Process(target=worker, args=(r, world, barrier, name, out)) for r in range(world)]
    for p in procs:
        p.start()
    got = [out.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    qh, dbh = synth_gather(n_query=60_000, n_db=2400, db_size=700)
    dbh[1700] = dbh[3].copy()
    want = oracle.gather(qh, *oracle.make_csr(dbh), threshold_bp=0, scaled=1000, nthreads=8)
    for rank, ok, res in sorted(got):
        print(rank, ok, (res == want, len(res)) if isinstance(res, list) else res)


if __name__ == "__main__":
    main()
