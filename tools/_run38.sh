cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gather.py tests/test_gpu_counter_protocol.py tests/test_gpu_index_protocol.py -x -q 2>&1 | tail -3
for i in 1 2; do python tools/bench_gather.py 2>/dev/null | cut -c150-330; done
