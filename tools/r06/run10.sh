cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_sigload.py tests/test_gpu_collection.py -x -q -m gpu > $O/pytest_sigload.txt 2>&1; tail -40 $O/pytest_sigload.txt
