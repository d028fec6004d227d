cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
SMG_GATHER_TRACE=1 python tools/bench_gather.py > gpurun_out/run23_a.json 2> gpurun_out/run23_trace.txt
cut -c1-330 gpurun_out/run23_a.json; echo; tail -5 gpurun_out/run23_trace.txt | cut -c1-200
