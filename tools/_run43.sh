cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_compare.py tests/test_gpu_parallel.py -x -q 2>&1 | tail -3
python tools/bench_index_build.py 2>/dev/null
