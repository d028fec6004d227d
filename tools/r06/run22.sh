#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_sketch.py tests/test_gpu_protein.py -x -q -m gpu > $O/pytest_sketch.txt 2>&1; tail -3 $O/pytest_sketch.txt
timeout 1200 python -m pytest tests/test_gpu_compare.py -x -q -m gpu -k "angular or abund or ragged" > $O/pytest_abund.txt 2>&1; tail -2 $O/pytest_abund.txt
timeout 600 python tools/bench_compare_ext.py > $O/compare_ext.json 2> $O/compare_ext.err; cat $O/compare_ext.json
