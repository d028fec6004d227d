cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O; export TMPDIR=/tmp
( cd /tmp && SMG_SIGLOAD_TRACE=1 rocprofv3 --kernel-trace --stats -d /tmp/p6/sl -o p -- python $GRAFT_REPO_ROOT/tools/bench_sigload.py 10000 > $GRAFT_REPO_ROOT/$O/sigload10k.json 2> /tmp/p6_sl.log ); cat $O/sigload10k.json; grep "sigload\]" /tmp/p6_sl.log | tail -3
python profiles/summarize.py $(find /tmp/p6/sl -name "*.db" | head -1) > $O/r06_sigload_kernels.txt; grep "gz_\|sj_\|kernel  " $O/r06_sigload_kernels.txt | cut -c1-170
