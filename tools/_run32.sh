cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python tools/bench_index_build.py 2>/dev/null
SMG_COMPARE_INDEX=sort python tools/bench_index_build.py 2>/dev/null
