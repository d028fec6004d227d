#!/bin/bash
# round 6, final records: the GPU suite, kernel stats + PMC passes (sketch, gather, ext), then the default bench (which reads the PMC file)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
bash tools/prof_r06.sh all 2>&1 | tail -12
cp gpurun_out/r06_prof/r06_pmc.txt profiles/r06_pmc.txt; cp gpurun_out/r06_prof/r06_gather_pmc.txt profiles/r06_gather_pmc.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'P'
import json
l=open('gpurun_out/r06/bench.json').read().strip().splitlines()
print('stdout lines', len(l), 'bytes', len(l[-1]))
d=json.loads(l[-1]); print(d['value'], d['ms_per_step']); print(d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value']); print(d['summary'])
P
