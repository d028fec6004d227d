# round 6, GPU call 1: the GPU suite, the default bench (the line must parse and stay under 8 KB), the sketch PMC passes
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'P'
import json
l=open('gpurun_out/r06/bench.json').read().strip().splitlines()
print('stdout lines', len(l), 'bytes', len(l[-1]))
d=json.loads(l[-1]); print(d['value'], d['ms_per_step'], d['config']['workload']); print(d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value']); print(d['summary'])
P
bash tools/prof_r06.sh sketch 2>&1 | tail -5
