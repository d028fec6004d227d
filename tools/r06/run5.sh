cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_gunzip.py tests/test_gpu_ingest.py -x -q -m gpu > $O/pytest_gunzip.txt 2>&1; tail -5 $O/pytest_gunzip.txt
timeout 900 python tools/bench_gunzip.py 400 256 > $O/bench_gunzip.json 2> $O/bench_gunzip.err; echo rc=$?; tail -2 $O/bench_gunzip.err; cat $O/bench_gunzip.json
