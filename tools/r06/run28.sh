#!/bin/bash
# round 6: signature loading with two groups under way at a time
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
for rep in 1 2 3; do timeout 1500 python -m pytest tests/test_gpu_sigload.py tests/test_gpu_collection.py -x -q -m gpu 2>&1 | tail -1; done
SMG_SIGLOAD_TRACE=1 timeout 1500 python tools/bench_sigload.py 10000 100000 > $O/sigload.json 2> $O/sigload.err; python -c "
import json; d=json.loads(open('gpurun_out/r06/sigload.json').read().strip().splitlines()[-1]); print({k:(v['device']['seconds'], v['host']['seconds'], v['same_rows']) for k,v in d.items()})"
grep "sigload\]" $O/sigload.err | tail -6 | cut -c1-220
