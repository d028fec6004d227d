#!/bin/bash
# round 6: pass 1 of the device inflate without its scratch reloads (v_mbcnt for the in-batch prefix count)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_gunzip.py tests/test_gpu_sigload.py tests/test_gpu_ingest.py -x -q -m gpu > $O/pytest_gz.txt 2>&1; tail -2 $O/pytest_gz.txt
timeout 600 python tools/bench_gunzip.py > $O/bench_gunzip.json 2> $O/bench_gunzip.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r06/bench_gunzip.json').read().strip().splitlines()[-1])
for k,v in d.items():
    if isinstance(v,dict) and 'stages' in v: print(k, {a:v['stages'][a] for a in ('scan_ms','pass1_ms','pass2_ms','finish_ms','device_total_ms')}, v.get('sketch_Gbase_per_s'))
P
timeout 900 python tools/bench_sigload.py 10000 > $O/sigload.json 2> $O/sigload.err; python -c "
import json; d=json.loads(open('gpurun_out/r06/sigload.json').read().strip().splitlines()[-1]); print({k:(v['device']['seconds'], v['same_rows']) for k,v in d.items()})"
timeout 600 python tools/bench_sketch_files.py 256 16 | cut -c200-400
