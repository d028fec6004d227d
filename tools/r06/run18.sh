#!/bin/bash
# round 6: the general all-pairs kernel after the walk kernel and the table variants were removed; where `auto` takes it
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_compare.py -x -q -m gpu > $O/pytest_compare.txt 2>&1; tail -4 $O/pytest_compare.txt
timeout 900 python tools/bench_compare_small.py > $O/compare_small.jsonl 2> $O/compare_small.err; cat $O/compare_small.jsonl | cut -c1-330; tail -3 $O/compare_small.err
