#!/bin/bash
# round 5, experiment 12: run-time-k kernel for k > 128 (sketch_words.hip): parity tests, rate by k, byte loop beside it
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05_exp12; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_sketch.py tests/test_gpu_minhash_api.py -m gpu -q -x 2>&1 | tail -8 ) > $O/pytest_gpu.txt
timeout 600 python tools/bench_sketch_k.py long > $O/long_k.json 2> $O/long_k.err
timeout 300 python tools/bench_compare_ext.py > $O/ext.json 2>/dev/null
