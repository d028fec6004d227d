# merge-path compare kernels side by side (GPU box): the hash-table tile kernel (default) and the walk kernel
cd $GRAFT_REPO_ROOT
for k in hash walk; do
  echo "== SMG_COMPARE_KERNEL=$k"
  SMG_COMPARE_KERNEL=$k timeout 300 python tools/bench_compare.py 2>&1 | grep -v "^    bits" | cut -c1-200
  SMG_COMPARE_KERNEL=$k timeout 300 python tools/bench_compare.py c4 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('c4 merge:', d['merge'], 'checks', all(d['checks'].values()))"
done
