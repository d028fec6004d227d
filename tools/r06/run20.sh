#!/bin/bash
# round 6: at most four ingest pipelines side by side: 256 files with threads = 1 / 4 / 16; the ingest tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 600 python tools/bench_sketch_files.py 256 16 > $O/sketch_files.json 2> $O/sketch_files.err; cat $O/sketch_files.json
timeout 600 python tools/bench_sketch_files.py 256 16 >> $O/sketch_files.json 2>> $O/sketch_files.err; tail -1 $O/sketch_files.json
timeout 1500 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_gunzip.py -x -q -m gpu > $O/pytest_ingest.txt 2>&1; tail -4 $O/pytest_ingest.txt
