cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_compare.py tests/test_gpu_parallel.py tests/test_gpu_collection.py -x -q 2>&1 | tail -25 > gpurun_out/run31_tests.txt
cat gpurun_out/run31_tests.txt
bash tools/_run30.sh | grep 'dx_\|bitmatrix'
