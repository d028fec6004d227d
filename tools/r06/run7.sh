cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_gunzip.py tests/test_gpu_ingest.py -x -q -m gpu > $O/pytest_gunzip.txt 2>&1; tail -5 $O/pytest_gunzip.txt
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p6/gz -o p -- python $GRAFT_REPO_ROOT/tools/bench_gunzip.py 400 256 > $GRAFT_REPO_ROOT/$O/bench_gunzip.json 2> /tmp/p6_gz.log ) || tail -3 /tmp/p6_gz.log
cat $O/bench_gunzip.json
python profiles/summarize.py $(find /tmp/p6/gz -name "*.db" | head -1) > $O/r06_gunzip_kernels.txt; grep "gz_\|kernel  " $O/r06_gunzip_kernels.txt | cut -c1-170
