# round 6, GPU call 2: the device inflater -- its tests, then the stage timings
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gunzip.py -x -q -m gpu > $O/pytest_gunzip.txt 2>&1; tail -15 $O/pytest_gunzip.txt
timeout 900 python tools/bench_gunzip.py 400 256 > $O/bench_gunzip.json 2> $O/bench_gunzip.err; echo "rc=$?"; tail -3 $O/bench_gunzip.err; cat $O/bench_gunzip.json
