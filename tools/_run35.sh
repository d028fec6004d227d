cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python tools/bench_compare_sparse.py 2>&1 | tail -2
SMG_COMPARE_INDEX=sort python tools/bench_compare_sparse.py 2>&1 | tail -1
