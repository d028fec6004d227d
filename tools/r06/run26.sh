#!/bin/bash
# round 6: the object entry points after the handle gathering got cheaper: object-API tests, the API lines of the bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_object_api_at_scale.py tests/test_gpu_index_protocol.py tests/test_gpu_compare.py -x -q -m gpu > $O/pytest_api.txt 2>&1; tail -2 $O/pytest_api.txt
timeout 900 python bench.py --no-cpu-baseline > $O/bench_api.json 2> $O/bench_api.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r06/bench_api.json').read().strip().splitlines()[-1])
s=d['summary']; print({k:s[k] for k in ('compare_api_10000_objects_ms','compare_api_1000_objects_ms','gather_api_c5_objects_ms','c5_gather_total_ms')})
e=json.load(open('gpurun_out/bench_extra.json'))['extra']['gather_api_c5']; print({k:e[k] for k in ('total_ms','pack_and_upload_ms','gather_ms','pcie_bound_ms','wall_over_kernel_plus_transfer')})
P
