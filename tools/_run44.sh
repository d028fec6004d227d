cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_compare.py -x -q -k "random_collections" 2>&1 | grep -n "test_gpu_compare.py:\|^E \|passed\|failed" | head -20
