#!/bin/bash
# round 6: the toucher beside the resident gather loop: A/B by rows touched per round, then the gather tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
for k in 0 1 2 3 4; do echo "== toucher $k"; SMG_GATHER_TOUCHER=$k SMG_GATHER_TRACE=1 timeout 300 python tools/bench_gather.py 2> $O/gather_trace_$k.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('index_build_ms','rounds','loop_ms','us_per_round','total_ms')}, all(d['checks'].values()))"; grep "persistent loop\|waiting for" $O/gather_trace_$k.err | tail -2 | cut -c1-220; done
timeout 1500 python -m pytest tests/test_gpu_gather.py tests/test_gpu_counter_protocol.py -x -q -m gpu > $O/pytest_gather.txt 2>&1; tail -3 $O/pytest_gather.txt
