#!/bin/bash
# round 6: the signature's ksizes in one pass (sketch_multi.hip): ingest tests, 256 files, kernel table
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_gunzip.py -x -q -m gpu > $O/pytest_ingest.txt 2>&1; tail -4 $O/pytest_ingest.txt
timeout 600 python tools/bench_sketch_files.py 256 16 > $O/sketch_files.json 2> $O/sketch_files.err; cat $O/sketch_files.json
timeout 600 python tools/bench_sketch_files.py 256 16 >> $O/sketch_files.json 2>> $O/sketch_files.err; tail -1 $O/sketch_files.json
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p6/sf2 -o p -- python $GRAFT_REPO_ROOT/tools/bench_sketch_files.py 256 16 > /dev/null 2> /tmp/p6_sf.log )
python profiles/summarize.py $(find /tmp/p6/sf2 -name "*.db" | head -1) > $O/r06_sketch_files_kernels.txt; head -14 $O/r06_sketch_files_kernels.txt | cut -c1-150
