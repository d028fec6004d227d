#!/bin/bash
# round 6: abundance all-pairs with the lists merged slice by slice (no radix sorts): parity tests, rates, kernel table
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_compare.py -x -q -m gpu -k "angular or abund or ragged" > $O/pytest_abund.txt 2>&1; tail -4 $O/pytest_abund.txt
timeout 600 python tools/bench_compare_ext.py > $O/compare_ext.json 2> $O/compare_ext.err; cat $O/compare_ext.json
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p6/ce -o p -- python $GRAFT_REPO_ROOT/tools/bench_compare_ext.py > /dev/null 2> /tmp/p6_ce.log )
python profiles/summarize.py $(find /tmp/p6/ce -name "*.db" | head -1) > $O/r06_compare_ext_kernels.txt; grep "ap_\|kernel  " $O/r06_compare_ext_kernels.txt | cut -c1-170
