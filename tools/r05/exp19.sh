#!/bin/bash
# round 5, experiment 19: abundance join on distinct hashes, a wave per match
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05_exp19; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_compare.py -m gpu -q -x -k "abund or angular" 2>&1 | tail -5 ) > $O/pytest_gpu.txt
timeout 300 python tools/bench_compare_ext.py > $O/ext.json 2>/dev/null
for z in 1 2 4 16; do SMG_ABUND_SLICES=$z timeout 300 python tools/bench_compare_ext.py 2>/dev/null | tail -1 >> $O/ext_sweep.txt; done
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p19/x -o p -- python $GRAFT_REPO_ROOT/tools/bench_compare_ext.py > /dev/null 2> /tmp/p19.log ) || tail -3 /tmp/p19.log
python profiles/summarize.py $(find /tmp/p19/x -name "*.db" | head -1) > $O/ext_kernels.txt
