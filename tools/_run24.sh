cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_compare.py tests/test_gpu_parallel.py tests/test_gpu_collection.py -x -q 2>&1 | tail -5 > gpurun_out/run24_tests.txt
python tools/bench_compare.py c4 > gpurun_out/run24_c4.json 2> gpurun_out/run24_err.txt
cat gpurun_out/run24_tests.txt; cut -c1-900 gpurun_out/run24_c4.json; tail -3 gpurun_out/run24_err.txt
