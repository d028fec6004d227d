cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_compare.py -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | head -20
echo "== sweep"
timeout 300 python tools/bench_compare.py 2>&1 | grep -v "^    bits" | cut -c1-200
echo "== c4"
timeout 300 python tools/bench_compare.py c4 2>&1 | tail -1 > gpurun_out/cmp_c4_new.json
python - <<'P'
import json
d=json.loads(open('gpurun_out/cmp_c4_new.json').readline()); print('c4 merge:', d['merge'], 'checks', d['checks'])
P
