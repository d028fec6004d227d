#!/bin/bash
# round 5, experiment 4: own-row prefetch forms, dense sketch for every k, register-window protein kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05_exp4; mkdir -p $O
for pf in 0 1 2; do
  echo "== prefetch $pf" >> $O/gather_ab.txt
  SMG_GATHER_PREFETCH=$pf timeout 300 python tools/bench_gather.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d[k] for k in ('index_build_ms','rounds','loop_ms','us_per_round','overlap_pass_ms')}, all(d['checks'].values()))" >> $O/gather_ab.txt 2>&1
done
for pf in 1 2; do SMG_GATHER_PREFETCH=$pf SMG_GATHER_TRACE=1 timeout 300 python tools/bench_gather.py 2>&1 | grep "persistent loop, work\|of which" | tail -2 >> $O/gather_ab.txt; done
( timeout 1500 python -m pytest tests/test_gpu_sketch.py tests/test_gpu_protein.py tests/test_gpu_gather.py tests/test_gpu_minhash_api.py tests/test_gpu_compare.py -m gpu -q 2>&1 | tail -15 ) > $O/pytest_gpu.txt
timeout 600 python - > $O/protein.txt 2>&1 <<'PY'
import json, sys
sys.path.insert(0, '.')
import torch, numpy as np
import bench
from sourmash_amd import device as smd
extra = {}
bench.protein_extras(extra, torch, np, torch.device('cuda', 0), smd, None)
print(json.dumps(extra))
PY
