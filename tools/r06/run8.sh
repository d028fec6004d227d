cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_gunzip.py tests/test_gpu_ingest.py -x -q -m gpu > $O/pytest_gunzip.txt 2>&1; tail -15 $O/pytest_gunzip.txt
SMG_INGEST_TRACE=1 timeout 900 python tools/bench_sketch_files.py 256 16 > $O/many.json 2> $O/many.err; cat $O/many.json; grep "batch of" $O/many.err | tail -14
