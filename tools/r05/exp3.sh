#!/bin/bash
# round 5, experiment 3: gather loop prefetch forms (non-blocking), compare API after the views refactor, full GPU suite
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05_exp3; mkdir -p $O
P=$GRAFT_REPO_ROOT/sourmash_amd
for pf in 0 1 2 3; do
  echo "== prefetch $pf" >> $O/gather_ab.txt
  SMG_GATHER_PREFETCH=$pf timeout 300 python tools/bench_gather.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d[k] for k in ('index_build_ms','rounds','loop_ms','us_per_round','overlap_pass_ms')}, all(d['checks'].values()))" >> $O/gather_ab.txt 2>&1
done
for pf in 1 3; do SMG_GATHER_PREFETCH=$pf SMG_GATHER_TRACE=1 timeout 300 python tools/bench_gather.py 2>&1 | grep "persistent loop, work\|of which" | tail -2 >> $O/gather_ab.txt; done
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $O/pytest_gpu.txt
timeout 600 python - > $O/api.txt 2>&1 <<'PY'
import json, time, sys
sys.path.insert(0, '.')
import torch, numpy as np
import bench
from sourmash_amd import device as smd
from sourmash_amd.synth import synth_sketches, synth_gather_device
extra = {}
dev = torch.device('cuda', 0)
# resident references for the ratios
for n, key in ((1000, 'compare_1000x1000_auto'), (10000, 'compare_10000x10000')):
    sk = synth_sketches(n, seed=1234); h, off = smd.pack_csr(sk, device=dev)
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); smd.compare_rows(h, off, method='auto'); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    extra[key] = {'ms': ms, 'auto_ms': ms}
bench.api_extras(extra, torch, np, dev, smd, synth_sketches, synth_gather_device)
print(json.dumps(extra, indent=1))
PY
