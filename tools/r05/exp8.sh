#!/bin/bash
# round 5, experiment 8: abundance join with prefetch, translate by output words
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05_exp8; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_protein.py tests/test_gpu_compare.py tests/test_gpu_signature_api.py tests/test_gpu_sketch.py -m gpu -q -x 2>&1 | tail -8 ) > $O/pytest_gpu.txt
timeout 300 python tools/bench_compare_ext.py > $O/ext.json 2>/dev/null
for z in 2 4 16; do SMG_ABUND_SLICES=$z timeout 300 python tools/bench_compare_ext.py 2>/dev/null | tail -1 >> $O/ext_sweep.txt; done
timeout 300 python tools/bench_protein.py > $O/protein.json 2>/dev/null
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p8/x -o p -- python $GRAFT_REPO_ROOT/tools/bench_compare_ext.py > /dev/null 2> /tmp/p8.log ) || tail -3 /tmp/p8.log
python profiles/summarize.py $(find /tmp/p8/x -name "*.db" | head -1) > $O/ext_kernels.txt
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p8/p -o p -- python $GRAFT_REPO_ROOT/tools/bench_protein.py > /dev/null 2> /tmp/p8p.log ) || tail -3 /tmp/p8p.log
python profiles/summarize.py $(find /tmp/p8/p -name "*.db" | head -1) > $O/protein_kernels.txt
