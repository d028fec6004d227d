#!/bin/bash
# round 6: where the time of `sketch` over 256 .fna.gz files goes: per-batch trace, kernel calls per file
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
SMG_INGEST_TRACE=1 timeout 600 python tools/bench_sketch_files.py 256 16 > $O/sketch_files.json 2> $O/sketch_files.err; cat $O/sketch_files.json; grep "batch of" $O/sketch_files.err | tail -12
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p6/sf -o p -- python $GRAFT_REPO_ROOT/tools/bench_sketch_files.py 256 16 > /dev/null 2> /tmp/p6_sf.log )
python profiles/summarize.py $(find /tmp/p6/sf -name "*.db" | head -1) > $O/r06_sketch_files_kernels.txt; head -40 $O/r06_sketch_files_kernels.txt | cut -c1-150
