"""Device-memory hygiene: 120 rounds of one-shot compare, persistent compare index, gather index + loop, everything dropped
after each round; free device memory must not drift (the library's buffers come from the stream-ordered pool).
python tools/stress_leaks.py   ->  'drift MB 0.0' on one MI355X (round 2)"""
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
free0 = None
for it in range(120):
    c, j = smd.compare_rows(h, off, method="auto")
    idx = smd.BitIndex.build(h, off, threshold=20)
    c2, _ = smd.compare_rows(h, off, index=idx)
    st = be.gather_state(q, len(qh), gh, goff, len(dbh), 0)
    st.begin(5, len(dbh)); r = st.run()
    del c, j, idx, c2, st
    torch.cuda.synchronize()
    if it in (10, 60, 119):
        free, total = torch.cuda.mem_get_info()
        print(it, "free GB %.3f" % (free / 1e9), "rounds", len(r))
        if it == 10: free0 = free
print("drift MB", (free0 - free) / 1e6)
