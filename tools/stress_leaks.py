"""Device-memory hygiene: 120 rounds of one-shot compare, persistent compare index, gather index + resident loop, the wide overlap
pass and per-pair calls on changing sketches, everything dropped after each round; free device memory must not drift, the
arena (csrc/arena.hpp) must stop going to the driver after the first rounds and its live bytes must come back to the same value.
python tools/stress_leaks.py   ->  'drift MB 0.0', 'driver allocations after warm-up 0' on one MI355X"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sourmash_amd import device as smd, parallel
from sourmash_amd.synth import synth_sketches, synth_gather
sk = synth_sketches(2000, seed=5, pool_size=50_000, keep_one_in=10)
h, off = smd.pack_csr(sk)
qh, dbh = synth_gather(n_query=200_000, n_db=5000, db_size=500)
gh, goff = smd.pack_csr(dbh)
q = torch.from_numpy(qh.view('int64').copy()).cuda()
be = parallel.DeviceBackend()
import sourmash_amd as sm
import numpy as np
rng = np.random.default_rng(3)
mhs = [sm.MinHash(0, 31, scaled=1) for _ in range(8)]
for m in mhs:
    m.add_many(rng.integers(1, 2**40, size=3000).tolist())
cnt = be.zeros((len(dbh),), torch.int64)
free0 = None
a10 = None
for it in range(120):
    c, j = smd.compare_rows(h, off, method="auto")
    idx = smd.BitIndex.build(h, off, threshold=20)
    c2, _ = smd.compare_rows(h, off, index=idx)
    st = be.gather_state(q, len(qh), gh, goff, len(dbh), 0)
    st.begin(5, len(dbh)); r = st.run()
    be.overlaps(q, len(qh), gh, goff, len(dbh), cnt, 0)
    mhs[it % 8].add_hash(int(rng.integers(1, 2**40)))          # a mirror goes stale every round
    pairs = sum(mhs[i].count_common(mhs[(i + 1) % 8]) for i in range(8))
    del c, j, idx, c2, st
    torch.cuda.synchronize()
    if it in (10, 60, 119):
        free, total = torch.cuda.mem_get_info()
        print(it, "free GB %.3f" % (free / 1e9), "rounds", len(r))
        print("   arena", smd.arena_stats())
        if it == 10: free0, a10 = free, smd.arena_stats()
a = smd.arena_stats()
print("drift MB", (free0 - free) / 1e6)
print("driver allocations after warm-up", a["driver_allocs"] - a10["driver_allocs"], "live bytes drift", a["live_bytes"] - a10["live_bytes"])
