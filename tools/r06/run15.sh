# the whole GPU suite after the inflater / loader work, then the default bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'P'
import json
l=open('gpurun_out/r06/bench.json').read().strip().splitlines()
print('stdout lines', len(l), 'bytes', len(l[-1]))
d=json.loads(l[-1]); print(d['value'], d['ms_per_step']); print(d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value']); print(d['summary'])
P
