cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_compare.py -x -q -k "builders or index" 2>&1 | tail -3
python tools/bench_compare_sparse.py 2>/dev/null | tail -1 > gpurun_out/r02_compare_sparse.json; cat gpurun_out/r02_compare_sparse.json
SMG_COMPARE_INDEX=sort python tools/bench_compare_sparse.py 2>/dev/null | tail -1
python tools/bench_index_build.py 2>/dev/null > gpurun_out/r02_index_build.json; cat gpurun_out/r02_index_build.json
SMG_COMPARE_INDEX=sort python tools/bench_index_build.py 2>/dev/null > gpurun_out/r02_index_build_sort.json; cat gpurun_out/r02_index_build_sort.json
