cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_deferred.py tests/test_gpu_minhash_api.py tests/test_gpu_signature_api.py -x -q 2>&1 | tail -5 > gpurun_out/run20_tests.txt
python tools/bench_small_calls.py > gpurun_out/run20_small.json 2> gpurun_out/run20_err.txt
cat gpurun_out/run20_tests.txt; cat gpurun_out/run20_small.json; tail -3 gpurun_out/run20_err.txt
