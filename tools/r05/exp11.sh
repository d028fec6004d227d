#!/bin/bash
# round 5, experiment 11: abundance join: which runs go through the worklist (inline threshold 16 / 64 = never / 4), on C3's shape and
# on a collection with a core of shared hashes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05_exp11; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_compare.py -m gpu -q -x -k "abund or angular" 2>&1 | tail -5 ) > $O/pytest_gpu.txt
timeout 300 python tools/bench_compare_ext.py > $O/ext.json 2>/dev/null
for v in i64 i4; do echo $v >> $O/ext_variants.txt; SMG_LIBRARY=$GRAFT_REPO_ROOT/sourmash_amd/libsourmash_amd_$v.so timeout 300 python tools/bench_compare_ext.py 2>/dev/null | tail -1 >> $O/ext_variants.txt; done
echo walk >> $O/ext_variants.txt; SMG_COMPARE_ABUND=walk timeout 300 python tools/bench_compare_ext.py 2>/dev/null | tail -1 >> $O/ext_variants.txt
for z in 4 16; do SMG_ABUND_SLICES=$z timeout 300 python tools/bench_compare_ext.py 2>/dev/null | tail -1 >> $O/ext_sweep.txt; done
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p11/x -o p -- python $GRAFT_REPO_ROOT/tools/bench_compare_ext.py > /dev/null 2> /tmp/p11.log ) || tail -3 /tmp/p11.log
python profiles/summarize.py $(find /tmp/p11/x -name "*.db" | head -1) > $O/ext_kernels.txt
