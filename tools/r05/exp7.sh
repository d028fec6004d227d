#!/bin/bash
# round 5, experiment 7: the lean ring (element granule) against the plain cursor loads; API lines with page-locked results
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05_exp7; mkdir -p $O
P=$GRAFT_REPO_ROOT/sourmash_amd
for v in "" ring; do
  lib=$P/libsourmash_amd${v:+_$v}.so
  echo "== lib ${v:-plain}" >> $O/ring.txt
  SMG_LIBRARY=$lib timeout 300 python tools/bench_gather.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d[k] for k in ('index_build_ms','rounds','loop_ms','us_per_round','overlap_pass_ms')}, all(d['checks'].values()))" >> $O/ring.txt 2>&1
  for qpr in 9600 11000; do
    SMG_LIBRARY=$lib SMG_OVERLAP_QPR=$qpr timeout 120 python tools/bench_overlap.py --reps 9 2>/dev/null | tail -1 >> $O/ring.txt
  done
  for rows in 196 261; do
    SMG_LIBRARY=$lib SMG_OVERLAP_ROWS=$rows timeout 120 python tools/bench_overlap.py --reps 9 2>/dev/null | tail -1 >> $O/ring.txt
  done
done
( SMG_LIBRARY=$P/libsourmash_amd_ring.so timeout 900 python -m pytest tests/test_gpu_gather.py -m gpu -q -x 2>&1 | tail -5 ) > $O/pytest_ring.txt
for v in "" ring; do
  lib=$P/libsourmash_amd${v:+_$v}.so
  ( cd /tmp && SMG_LIBRARY=$lib rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf7_${v:-plain} -o p -- python $GRAFT_REPO_ROOT/tools/bench_overlap.py --reps 3 > /dev/null 2> /tmp/pf7.log ) || tail -3 /tmp/pf7.log >> $O/fetch.txt
  echo "== lib ${v:-plain}" >> $O/fetch.txt
  python profiles/summarize.py $(find /tmp/pf7_${v:-plain} -name "*.db" | head -1) | grep -i "overlap_lean" >> $O/fetch.txt
done
timeout 600 python - > $O/api.txt 2>&1 <<'PY'
import json, time, sys
sys.path.insert(0, '.')
import torch, numpy as np
import bench
from sourmash_amd import device as smd
from sourmash_amd.synth import synth_sketches, synth_gather_device
extra = {}
dev = torch.device('cuda', 0)
for n, key in ((1000, 'compare_1000x1000_auto'), (10000, 'compare_10000x10000')):
    sk = synth_sketches(n, seed=1234); h, off = smd.pack_csr(sk, device=dev)
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); smd.compare_rows(h, off, method='auto'); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    extra[key] = {'ms': ms, 'auto_ms': ms}
bench.synth_gather_device = None
import types
def no_gather(*a, **k): raise RuntimeError("skipped")
try:
    bench.api_extras(extra, torch, np, dev, smd, synth_sketches, no_gather)
except RuntimeError:
    pass
print(json.dumps({k: v for k, v in extra.items() if k.startswith('compare_api')}, indent=1))
PY
