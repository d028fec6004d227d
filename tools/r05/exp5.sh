#!/bin/bash
# round 5, experiment 5: pruned gather / overlap split, abundance join, table-driven residues / translate
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05_exp5; mkdir -p $O
( timeout 1800 python -m pytest tests/test_gpu_compare.py tests/test_gpu_gather.py tests/test_gpu_protein.py tests/test_gpu_full_configs.py tests/test_gpu_parallel.py tests/test_gpu_index_protocol.py tests/test_gpu_counter_protocol.py -m gpu -q -x 2>&1 | tail -25 ) > $O/pytest_gpu.txt
timeout 300 python tools/bench_gather.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d[k] for k in ('index_build_ms','rounds','loop_ms','us_per_round','overlap_pass_ms')}, all(d['checks'].values()))" > $O/gather.txt 2>&1
timeout 600 python - > $O/extras.txt 2>&1 <<'PY'
import json, sys
sys.path.insert(0, '.')
import torch, numpy as np
import bench
from sourmash_amd import device as smd, parallel
from sourmash_amd.synth import synth_sketches
dev = torch.device('cuda', 0)
be = parallel.DeviceBackend(dev)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
extra = {}
bench.compare_ext_extras(extra, torch, np, dev, be, smd, synth_sketches, timed)
bench.protein_extras(extra, torch, np, dev, smd, None)
print(json.dumps(extra))
PY
for z in 1 2 4 8 16; do echo "slices $z" >> $O/abund_sweep.txt; SMG_ABUND_SLICES=$z timeout 300 python tools/bench_compare_ext.py 2>&1 | tail -2 >> $O/abund_sweep.txt; done
SMG_COMPARE_ABUND=walk timeout 300 python tools/bench_compare_ext.py 2>&1 | tail -2 >> $O/abund_sweep.txt
