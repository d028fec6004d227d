cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_compare.py -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | head -20
for v in 0 2; do
  echo "== variant $v"
  SMG_COMPARE_VARIANT=$v timeout 300 python tools/bench_compare.py c4 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('c4 merge:', d['merge'], 'checks', all(d['checks'].values()))"
done
echo "== sweep"
timeout 300 python tools/bench_compare.py 2>&1 | grep -v "^    bits" | cut -c1-200
