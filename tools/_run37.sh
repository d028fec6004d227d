cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_compare.py -x -q -k "builders" 2>&1 | tail -30
